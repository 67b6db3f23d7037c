"""Piped-ring driver: one process per GPU, layer windows per rank, activations handed neighbour to
neighbour with torch.distributed send/recv (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests) instead of the reference's ZeroMQ PUSH/PULL + host bounce (llama_send_tensors /
llama_recv_tensors, src/llama.cpp:18031-18077, ring loop :18503-18564).

The reference keeps ONE batch in flight (rank 0 blocks in recv until the token returns), so a layer
split gives no throughput. Here `world` independent sequences are in flight, staggered by one rank:
at micro-step m rank r works on sequence (m - r) mod world; after `world` micro-steps every sequence
has advanced one token. Rank 0 additionally runs the head (result_norm + lm_head + greedy argmax) on the
activation returned by the last rank, exactly like prima.cpp's rank 0 owns [inp_embd, its window, lm_head].

The schedule is transport- and compute-agnostic (`RankCompute` interface) so it is covered by
world_size-2 gloo tests on CPU with a fake window; `EngineCompute` binds it to the HIP engine.
"""
import torch
import torch.distributed as dist


def partition_layers(layer_bytes, head_bytes, world):
    """Contiguous windows minimising the maximum bytes a rank streams per token (rank 0 also owns the head).
    Returns a list of (lo, hi). The reference does this with -lw / the HiGHS scheduler (common/common.cpp:1717)."""
    n = len(layer_bytes)
    assert 1 <= world <= n
    pre = [0]
    for b in layer_bytes:
        pre.append(pre[-1] + b)
    INF = float("inf")
    # best[r][i] = minimal max-load for ranks r..world-1 covering layers i..n-1
    best = [[INF] * (n + 1) for _ in range(world + 1)]
    cut = [[-1] * (n + 1) for _ in range(world + 1)]
    best[world][n] = 0
    for r in range(world - 1, -1, -1):
        for i in range(n - (world - r) + 1):
            for j in range(i + 1, n - (world - r - 1) + 1):
                load = pre[j] - pre[i] + (head_bytes if r == 0 else 0)
                v = max(load, best[r + 1][j])
                if v < best[r][i]:
                    best[r][i], cut[r][i] = v, j
    out, i = [], 0
    for r in range(world):
        j = cut[r][i]
        out.append((i, j))
        i = j
    return out


class RankCompute:
    """What a rank does inside one micro-step. Buffers are torch tensors on the rank's device."""

    n_embd = 0
    device = "cpu"

    def first_rank_step(self, seq, x_last, forced_token):
        """rank 0: (head on x_last unless it is None) -> next token of `seq` (or forced_token) -> embed -> window.
        Returns the activation tensor to send on."""
        raise NotImplementedError

    def rank_step(self, seq, x_in):
        """rank > 0: window on x_in for sequence `seq`; returns activation to send on."""
        raise NotImplementedError


class RingDriver:
    """Runs the staggered schedule for one rank. Every micro-step ends with ONE grouped exchange
    (dist.batch_isend_irecv = ncclGroupStart/End): send this micro-step's activation to the next rank and
    post the receive for the next micro-step from the previous rank. Grouping makes the 2-rank ring (where
    next == previous) and the NCCL per-communicator ordering deadlock-free; a rank sends at micro-step m iff
    it is active (m >= rank) and its successor receives for m+1 under exactly the same condition."""

    def __init__(self, compute, rank, world, group=None, c_ring=None):
        """c_ring: a prima_cpp_amd.ring.CRing (RCCL transport in C, pm355_ring_* of include/prima_mi355.h): the exchange is enqueued
        on the library's communication stream with event hand-off and the host never waits; without it the exchange goes through
        torch.distributed (gloo in the CPU tests) and the host waits for the requests at the start of the next micro-step."""
        self.c, self.rank, self.world, self.group = compute, rank, world, group
        self.c_ring = c_ring if world > 1 else None
        self.nxt, self.prv = (rank + 1) % world, (rank - 1) % world
        # receive buffers alternate so the receive for m+1 never targets the buffer micro-step m reads
        self.x_in = [torch.empty((1, compute.n_embd), dtype=torch.float32, device=compute.device) for _ in range(2)]
        self.m = 0
        self.reqs = []
        self.last_out = None
        # gloo has no device-memory send/recv: when the transport is gloo and the buffers live on a GPU (single-GPU
        # tests only), stage through pinned host tensors. With nccl (= RCCL) the device buffers go on the wire directly.
        dev_is_gpu = torch.device(compute.device).type == "cuda"
        self.host_stage = world > 1 and dev_is_gpu and self.c_ring is None and dist.get_backend(group) == "gloo"
        if self.host_stage:
            self.h_in = [torch.empty((1, compute.n_embd), dtype=torch.float32).pin_memory() for _ in range(2)]
            self.h_out = [torch.empty((1, compute.n_embd), dtype=torch.float32).pin_memory() for _ in range(2)]
            self.recv_posted = [False, False]

    def _need_recv(self, m):
        if self.world == 1:
            return False
        return m >= self.world if self.rank == 0 else m >= self.rank

    def _wait(self):
        if self.c_ring is not None:
            self.c_ring.wait()                        # device-side: the compute stream waits for the exchange's event
            return
        for q in self.reqs:
            q.wait()
        self.reqs = []

    def micro_step(self, forced_token=None):
        """One micro-step of this rank. Returns the sequence id processed, or None while the pipeline fills."""
        m, r, W = self.m, self.rank, self.world
        self.m += 1
        active = m >= r
        seq = (m - r) % W if active else None
        self._wait()                                  # previous exchange: our last send left, this step's input arrived
        out = None
        if active:
            if W == 1:
                x_in = self.last_out
            else:
                x_in = self.x_in[m & 1] if self._need_recv(m) else None
                if x_in is not None and self.host_stage:
                    x_in.copy_(self.h_in[m & 1], non_blocking=False)
            out = self.c.first_rank_step(seq, x_in, forced_token) if r == 0 else self.c.rank_step(seq, x_in)
            self.last_out = out
        if W > 1 and self.c_ring is not None:
            rcv = self.x_in[(m + 1) & 1] if self._need_recv(m + 1) else None
            self.c_ring.exchange(out if active else None, rcv)       # ncclGroupStart; ncclSend; ncclRecv; ncclGroupEnd on the comm stream
        elif W > 1:
            ops = []
            if active:
                snd = out
                if self.host_stage:
                    self.h_out[m & 1].copy_(out)
                    torch.cuda.current_stream().synchronize()
                    snd = self.h_out[m & 1]
                ops.append(dist.P2POp(dist.isend, snd, self.nxt, self.group))
            if self._need_recv(m + 1):
                rcv = self.h_in[(m + 1) & 1] if self.host_stage else self.x_in[(m + 1) & 1]
                ops.append(dist.P2POp(dist.irecv, rcv, self.prv, self.group))
            if ops:
                self.reqs = dist.batch_isend_irecv(ops)
        return seq

    def flush(self):
        """After the last micro-step: the receive posted for the never-executed next micro-step absorbs the
        predecessor's final send, so every send has been matched; wait for both."""
        self._wait()


class CRing:
    """RCCL transport in C (prima_cpp_amd/csrc/ring.hip). The 128-byte unique id is created on rank 0 and handed to the other
    ranks by the launcher - here through the already initialised torch.distributed group (process launch is all Python keeps)."""

    def __init__(self, rank, world, group=None):
        import ctypes as C
        from . import lib as L
        self.lib = L.load()
        self.lib.pm355_ring_init.restype = C.c_void_p
        self.lib.pm355_ring_init.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.lib.pm355_ring_free.argtypes = [C.c_void_p]
        self.lib.pm355_ring_error.restype = C.c_char_p
        self.lib.pm355_ring_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        self.lib.pm355_ring_wait.argtypes = [C.c_void_p, C.c_void_p]
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            rc = self.lib.pm355_ring_unique_id(buf)
            if rc:
                raise L.PM355Error(f"pm355_ring_unique_id rc={rc}: {self.lib.pm355_ring_error().decode()}")
            ident[0] = bytes(buf.raw)
        if world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        self.h = self.lib.pm355_ring_init(ident[0], rank, world)
        if not self.h:
            raise L.PM355Error(f"pm355_ring_init failed: {self.lib.pm355_ring_error().decode()}")
        self.rank, self.world = rank, world

    def _chk(self, rc, what):
        if rc:
            from . import lib as L
            raise L.PM355Error(f"{what} rc={rc}: {self.lib.pm355_ring_error().decode()}")

    def exchange(self, send, recv):
        n = (send if send is not None else recv).numel() if (send is not None or recv is not None) else 0
        st = torch.cuda.current_stream().cuda_stream
        self._chk(self.lib.pm355_ring_exchange(self.h, send.data_ptr() if send is not None else None,
                                               recv.data_ptr() if recv is not None else None, n, st), "pm355_ring_exchange")

    def wait(self):
        self._chk(self.lib.pm355_ring_wait(self.h, torch.cuda.current_stream().cuda_stream), "pm355_ring_wait")

    def close(self):
        if self.h:
            self.lib.pm355_ring_free(self.h)
            self.h = None


class EngineCompute(RankCompute):
    """Binds the schedule to the HIP engine (prima_cpp_amd.engine.Window). The window must have been finalized with
    n_seq == world; the engine's current-sequence counter rotates by one per step, which is exactly the order in
    which a rank meets the in-flight sequences, so no per-step host -> device traffic is needed."""

    def __init__(self, window, world, use_graph=True):
        self.w, self.world, self.use_graph = window, world, use_graph
        self.n_embd = window.hp.n_embd
        self.device = window.dev
        self.x_out = [torch.empty((1, self.n_embd), dtype=torch.float32, device=self.device) for _ in range(2)]
        self.cur = torch.zeros(1, dtype=torch.int32, device=self.device)          # token being fed (rank 0)
        self.k = 0
        self.expect_seq = 0

    def _out(self):
        self.k ^= 1
        return self.x_out[self.k]

    def _check_seq(self, seq):
        assert seq == self.expect_seq, (seq, self.expect_seq)
        self.expect_seq = (self.expect_seq + 1) % self.world

    def first_rank_step(self, seq, x_last, forced_token):
        self._check_seq(seq)
        out = self._out()
        if forced_token is not None or x_last is None:
            # prompt phase / pipeline fill: the token is dictated, the head result (if any) is not needed
            if forced_token is not None:
                self.cur.fill_(int(forced_token))
            self.w.step_ex(token=self.cur, x_out=out, advance=1, rotate=1, use_graph=self.use_graph)
        else:
            # steady state, ONE graph: head(x_last) -> argmax -> embed -> window
            self.w.step_ex(token=self.cur, x_in=x_last, x_out=out, argmax=self.cur, advance=1, rotate=1, head_first=True,
                           use_graph=self.use_graph)
        return out

    def rank_step(self, seq, x_in):
        self._check_seq(seq)
        out = self._out()
        self.w.step_ex(x_in=x_in, x_out=out, advance=1, rotate=1, use_graph=self.use_graph)
        return out
