"""Piped-ring driver: one process per GPU, layer windows per rank, activations handed neighbour to neighbour - in place of the reference's
ZeroMQ PUSH/PULL + host bounce (llama_send_tensors / llama_recv_tensors, src/llama.cpp:18031-18077, ring loop :18503-18564).

Where the work is: the SCHEDULES and the TRANSPORT live in C (csrc/ring.hip, `pm355_ring_*` of include/prima_mi355.h): RCCL send / recv on a
communication stream with events against the compute stream, the pipelined prompt pass (`pm355_ring_prefill`), the one-sequence loop
(`pm355_ring_single_token`) and the staggered multi-sequence decode loop (`pm355_ring_decode_staggered`). This module is process plumbing
around them: the byte-balancing layer partition, `CRing` (ctypes binding; transport "rccl", or "torch" = the same C schedule over two
callbacks that run torch.distributed batch_isend_irecv - what lets the world-size-2 tests drive the C code over gloo), and `RingDriver`, a
Python reference implementation of the staggered schedule (transport- and compute-agnostic through the `RankCompute` interface) that
the world-2/3 gloo tests compare the C schedule and a serial run with.

The reference keeps ONE batch in flight (rank 0 blocks in recv until the token returns), so a layer split gives no throughput. Here `world`
independent sequences are in flight, staggered by one rank: at micro-step m rank r works on sequence (m - r) mod world; after `world`
micro-steps every sequence has advanced one token. Rank 0 additionally runs the head (result_norm + lm_head + greedy argmax) on the
activation returned by the last rank, exactly like prima.cpp's rank 0 owns [inp_embd, its window, lm_head].
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


def prefill_schedule(rank, world, n_seq, n_prompt, ubatch):
    """The pipelined prompt pass of pm355_ring_prefill (csrc/ring.hip) as data: for every pipeline step s = 0 .. G + world - 2 the tuple
    (chunk or None, tokens, send_floats_per_embd, recv_floats_per_embd) of this rank, in units of n_embd floats. Chunk g = seq * C + c runs
    on rank r at step g + r; a window sends its whole [tokens][n_embd] output to the next rank, the last rank sends only the last row of a
    prompt's final chunk (to rank 0, which receives it at the end of the same step). Host-side mirror used by the CPU tests: every send
    must meet a receive of the same size on the neighbour in the same step."""
    C = (n_prompt + ubatch - 1) // ubatch
    G = n_seq * C

    def clen(g):
        c = g % C
        return n_prompt - c * ubatch if c == C - 1 else ubatch
    out = []
    for s in range(G + world - 1) if world > 1 else range(G):
        g = s - rank
        valid = 0 <= g < G
        T = clen(g) if valid else 0
        snd = 0
        if valid and world > 1:
            if rank < world - 1:
                snd = T
            elif g % C == C - 1:
                snd = 1
        rcv = 0
        if world > 1:
            if rank > 0:
                gn = g + 1
                if 0 <= gn < G:
                    rcv = clen(gn)
            else:
                gl = s - (world - 1)
                if 0 <= gl < G and gl % C == C - 1:
                    rcv = 1
        out.append((g if valid else None, T, snd, rcv))
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

    def __init__(self, compute, rank, world, group=None, c_ring=None, depth=1):
        """c_ring: a prima_cpp_amd.ring.CRing (RCCL transport in C, pm355_ring_* of include/prima_mi355.h): the exchange is enqueued
        on the library's communication stream with event hand-off and the host never waits; without it the exchange goes through
        torch.distributed (gloo in the CPU tests) and the host waits for the requests at the start of the next micro-step.
        depth = 2 (round 6, world > 1): 2 x world sequences in flight - rank r works on sequence (m - 2 r) mod (2 world), the row sent after
        micro-step m is consumed by the successor at m + 2, so the exchange of step m is only waited for before step m + 2 (the schedule
        of pm355_ring_decode_staggered on a window finalized for 2 x world sequences)."""
        self.c, self.rank, self.world, self.group = compute, rank, world, group
        self.depth = depth if world > 1 else 1
        assert self.depth in (1, 2)
        self.n_seq = self.depth * world
        self.c_ring = c_ring if world > 1 else None
        self.nxt, self.prv = (rank + 1) % world, (rank - 1) % world
        # receive buffers alternate so the receive for a later micro-step never targets the buffer micro-step m reads
        self.x_in = [torch.empty((1, compute.n_embd), dtype=torch.float32, device=compute.device) for _ in range(2)]
        self.m = 0
        self.reqs = {}                                # micro-step -> requests of the exchange enqueued at its end
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
        return m >= self.depth * self.world if self.rank == 0 else m >= self.depth * self.rank

    def _wait(self, upto):
        """every exchange enqueued at a micro-step <= upto has completed (our row has left, the input it carried is here)"""
        if self.c_ring is not None:
            self.c_ring.wait()                        # device-side: the compute stream waits for the exchange's event
            return
        for k in sorted(k for k in self.reqs if k <= upto):
            for q in self.reqs.pop(k):
                q.wait()

    def micro_step(self, forced_token=None):
        """One micro-step of this rank. Returns the sequence id processed, or None while the pipeline fills."""
        m, r, W, D = self.m, self.rank, self.world, self.depth
        self.m += 1
        active = m >= D * r
        seq = (m - D * r) % self.n_seq if active else None
        self._wait(m - D)                             # the exchange of D micro-steps ago: that row has left, this step's input arrived
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
            rcv = self.x_in[(m + D) & 1] if self._need_recv(m + D) else None
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
            if self._need_recv(m + D):
                rcv = self.h_in[(m + D) & 1] if self.host_stage else self.x_in[(m + D) & 1]
                ops.append(dist.P2POp(dist.irecv, rcv, self.prv, self.group))
            if ops:
                self.reqs[m] = dist.batch_isend_irecv(ops)
        return seq

    def flush(self):
        """After the last micro-step: the receive posted for the never-executed next micro-step absorbs the
        predecessor's final send, so every send has been matched; wait for both."""
        self._wait(self.m)


class CRing:
    """The ring in C (prima_cpp_amd/csrc/ring.hip, pm355_ring_* of include/prima_mi355.h): exchanges, multi-token micro-steps, the
    pipelined prompt pass and the single-sequence token loop. Two transports under the same C code:
      transport="rccl"  ncclSend / ncclRecv on the library's communication stream (one GPU per rank); the 128-byte unique id is created
                        on rank 0 and handed to the other ranks through the already initialised torch.distributed group
      transport="torch" the exchanges are carried by torch.distributed point-to-point operations of `group` (any backend; with gloo the
                        device buffers are staged through host memory) via the C API's transport callbacks - how two ranks share ONE GPU
                        in the tests and in `PM355_DIST_BACKEND=gloo python -m torch.distributed.run ... bench.py --gpus 2`."""

    def __init__(self, rank, world, group=None, transport="rccl"):
        import ctypes as C
        from . import lib as L
        self.lib = L.load()
        self.L = L
        lib = self.lib
        lib.pm355_ring_init.restype = C.c_void_p
        lib.pm355_ring_init.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.pm355_ring_init_cb.restype = C.c_void_p
        lib.pm355_ring_free.argtypes = [C.c_void_p]
        lib.pm355_ring_error.restype = C.c_char_p
        lib.pm355_ring_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.pm355_ring_exchange2.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        lib.pm355_ring_wait.argtypes = [C.c_void_p, C.c_void_p]
        lib.pm355_ring_prefill.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.pm355_ring_single_token.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pm355_ring_step_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        lib.pm355_ring_decode_staggered.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.pm355_ring_decode_last_output.restype = C.c_void_p
        lib.pm355_ring_decode_last_output.argtypes = [C.c_void_p]
        lib.pm355_ring_init_local.restype = C.c_void_p
        self.rank, self.world, self.group, self.transport = rank, world, group, transport
        self.h = None
        if transport == "local":
            assert world == 1 and rank == 0
            self.h = lib.pm355_ring_init_local()
        elif transport == "torch":
            XF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p)
            WF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
            self._pend = []                      # exchanges enqueued and not yet waited for, oldest first (the two-deep decode loop keeps two)
            self._xf, self._wf = XF(self._cb_exchange), WF(self._cb_wait)      # (kept alive with the object)
            lib.pm355_ring_init_cb.argtypes = [C.c_int, C.c_int, XF, WF, C.c_void_p]
            self.h = lib.pm355_ring_init_cb(rank, world, self._xf, self._wf, None)
        else:
            # every rank must leave the broadcast, also when rank 0 cannot create the id (then all of them fail together)
            ident = [None]
            err = None
            if rank == 0:
                buf = C.create_string_buffer(128)
                rc = lib.pm355_ring_unique_id(buf)
                if rc:
                    err = f"pm355_ring_unique_id rc={rc}: {lib.pm355_ring_error().decode()}"
                else:
                    ident[0] = bytes(buf.raw)
            if world > 1:
                dist.broadcast_object_list(ident, src=0, group=group)
            if ident[0] is None:
                raise L.PM355Error(err or "rank 0 could not create the RCCL unique id")
            self.h = lib.pm355_ring_init(ident[0], rank, world)
        if not self.h:
            raise L.PM355Error(f"pm355_ring_init failed: {lib.pm355_ring_error().decode()}")

    # ---- transport callbacks (transport="torch"): device buffers travel through torch.distributed ------------------------------
    def _cb_exchange(self, user, send, n_send, recv, n_recv, stream):
        try:
            lib = self.lib
            nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
            on_dev = dist.get_backend(self.group) != "gloo"
            ops, hs, hr = [], None, None
            dev = torch.device("cuda", torch.cuda.current_device())
            if send and n_send:
                hs = torch.empty(n_send, dtype=torch.float32, device=dev if on_dev else "cpu")
                cp = lib.pm355_memcpy_d2d if on_dev else lib.pm355_memcpy_d2h
                if cp(hs.data_ptr(), send, n_send * 4, stream) or lib.pm355_sync(stream):
                    return 1
                ops.append(dist.P2POp(dist.isend, hs, nxt, self.group))
            if recv and n_recv:
                hr = torch.empty(n_recv, dtype=torch.float32, device=dev if on_dev else "cpu")
                ops.append(dist.P2POp(dist.irecv, hr, prv, self.group))
            self._pend.append((dist.batch_isend_irecv(ops) if ops else [], hs, hr, recv, n_recv, on_dev))
            return 0
        except Exception as e:                         # never let an exception cross the C frames
            print(f"[ring transport] exchange failed: {e}", flush=True)
            return 1

    def _cb_wait(self, user, stream):
        try:
            while self._pend:                        # "everything enqueued so far", oldest first
                reqs, hs, hr, recv, n_recv, on_dev = self._pend.pop(0)
                for q in reqs:
                    q.wait()
                if hr is not None:
                    if on_dev:
                        torch.cuda.current_stream().synchronize()
                    cp = self.lib.pm355_memcpy_d2d if on_dev else self.lib.pm355_memcpy_h2d
                    if cp(recv, hr.data_ptr(), n_recv * 4, stream) or self.lib.pm355_sync(stream):
                        return 1
            return 0
        except Exception as e:
            print(f"[ring transport] wait failed: {e}", flush=True)
            return 1

    def _chk(self, rc, what):
        if rc:
            raise self.L.PM355Error(f"{what} rc={rc}: {self.lib.pm355_ring_error().decode()}")

    def exchange(self, send, recv):
        st = torch.cuda.current_stream().cuda_stream
        self._chk(self.lib.pm355_ring_exchange2(self.h, send.data_ptr() if send is not None else None, send.numel() if send is not None else 0,
                                                recv.data_ptr() if recv is not None else None, recv.numel() if recv is not None else 0, st),
                  "pm355_ring_exchange")

    def wait(self):
        self._chk(self.lib.pm355_ring_wait(self.h, torch.cuda.current_stream().cuda_stream), "pm355_ring_wait")

    def prefill(self, window, tokens, n_seq, n_prompt, ubatch, final_rows):
        """Pipelined prompt pass (pm355_ring_prefill): tokens int32 [n_seq, n_prompt] on rank 0 (None elsewhere); final_rows f32
        [n_seq, n_embd] on rank 0 receives each prompt's last hidden row from the last rank. The window needs finalize(max_tokens >= ubatch)."""
        st = torch.cuda.current_stream().cuda_stream
        self._chk(self.lib.pm355_ring_prefill(self.h, window.h, n_seq, tokens.data_ptr() if tokens is not None else None, n_prompt, ubatch,
                                              final_rows.data_ptr() if final_rows is not None else None, st), "pm355_ring_prefill")
        self.wait()

    def single_token(self, window, seq, token, logits=None):
        """ONE sequence in flight (the reference's mode): a token of `seq` once round the ring; rank 0 gets the next token in `token`."""
        st = torch.cuda.current_stream().cuda_stream
        self._chk(self.lib.pm355_ring_single_token(self.h, window.h, seq, token.data_ptr() if token is not None else None,
                                                   logits.data_ptr() if logits is not None else None, st), "pm355_ring_single_token")

    def decode_staggered(self, window, n_micro, forced=None, tokens_out=None, reset=False, use_graph=True):
        """n_micro micro-steps of the staggered multi-sequence decode schedule in C (pm355_ring_decode_staggered): no interpreter in the loop.
        forced: per micro-step token ids for rank 0 (None / negative = the head's argmax); tokens_out: int32 device tensor [n_micro] on rank 0."""
        import ctypes as C
        st = torch.cuda.current_stream().cuda_stream
        fa = None
        if forced is not None:
            fa = (C.c_int32 * n_micro)(*[-1 if f is None else int(f) for f in forced])
        self._chk(self.lib.pm355_ring_decode_staggered(self.h, window.h, n_micro, fa, tokens_out.data_ptr() if tokens_out is not None else None,
                                                       1 if reset else 0, 1 if use_graph else 0, st), "pm355_ring_decode_staggered")

    def last_output(self, n_embd):
        """this rank's last output row of the staggered loop, as a torch view (debug / finiteness checks)"""
        import ctypes as C
        p = self.lib.pm355_ring_decode_last_output(self.h)
        if not p:
            return None
        t = torch.empty(n_embd, dtype=torch.float32, device="cuda")
        self._chk(self.lib.pm355_memcpy_d2d(t.data_ptr(), p, n_embd * 4, torch.cuda.current_stream().cuda_stream), "copy of the last output row")
        return t

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
