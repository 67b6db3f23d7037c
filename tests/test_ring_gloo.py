"""world_size-2 / 3 CPU (gloo) tests of the piped-ring schedule with a fake window: the staggered,
multi-sequence ring must reproduce exactly what a single process computing every sequence serially does."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

E = 16
VOCAB = 97


def _layer(x, il, pos):
    """deterministic fake layer: depends on layer id, position and input (exact in fp32 integer range)"""
    return torch.remainder(x * 3.0 + float(il * 7 + pos), 1009.0)


def _embed(tok):
    return (torch.arange(E, dtype=torch.float32) * 5.0 + float(tok)).view(1, E)


def _head(x):
    return int(torch.remainder(x.sum(), VOCAB).item())


class FakeCompute:
    device = "cpu"
    n_embd = E

    def __init__(self, lo, hi, n_seq):
        self.lo, self.hi = lo, hi
        self.pos = [0] * n_seq
        self.tok = [None] * n_seq
        self.generated = [[] for _ in range(n_seq)]

    def _window(self, seq, x):
        for il in range(self.lo, self.hi):
            x = _layer(x, il, self.pos[seq])
        self.pos[seq] += 1
        return x.clone()

    def first_rank_step(self, seq, x_last, forced_token):
        if x_last is not None:
            t = _head(x_last)
            self.generated[seq].append(t)
            self.tok[seq] = t
        if forced_token is not None:
            self.tok[seq] = forced_token
        return self._window(seq, _embed(self.tok[seq]))

    def rank_step(self, seq, x_in):
        return self._window(seq, x_in)


def _serial(n_layer, world, first_tokens, n_rounds):
    out = []
    for s in range(world):
        tok, gen = first_tokens[s], []
        for pos in range(n_rounds):
            x = _embed(tok)
            for il in range(n_layer):
                x = _layer(x, il, pos)
            tok = _head(x)
            gen.append(tok)
        out.append(gen)
    return out


def _worker(rank, world, port, n_layer, n_rounds, q, depth=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prima_cpp_amd.ring import RingDriver, partition_layers
    wins = partition_layers([10 + (i % 3) for i in range(n_layer)], 12, world)
    lo, hi = wins[rank]
    n_seq = depth * world
    comp = FakeCompute(lo, hi, n_seq)
    drv = RingDriver(comp, rank, world, depth=depth)
    first = [11 + 3 * s for s in range(n_seq)]
    # n_rounds full rounds + one extra round on rank 0 so the last tokens are produced by the head
    total = n_seq * (n_rounds + 1)
    for m in range(total):
        forced = first[m] if (rank == 0 and m < n_seq) else None
        drv.micro_step(forced_token=forced)
    drv.flush()
    if rank == 0:
        q.put(comp.generated)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("world", [2, 3])
def test_ring_schedule_matches_serial(world, depth):
    """depth 2 (round 6): 2 x world sequences in flight, every hop consumed two micro-steps after it was sent - the same tokens as the serial run."""
    n_layer, n_rounds = 7, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_layer, n_rounds, q, depth)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n_seq = depth * world
    want = _serial(n_layer, n_seq, [11 + 3 * s for s in range(n_seq)], n_rounds)
    for s in range(n_seq):
        assert got[s][:n_rounds] == want[s], (s, got[s], want[s])


def test_ring_world1_matches_serial():
    from prima_cpp_amd.ring import RingDriver
    comp = FakeCompute(0, 5, 1)
    drv = RingDriver(comp, 0, 1)
    for m in range(7):
        drv.micro_step(forced_token=11 if m == 0 else None)
    assert comp.generated[0][:6] == _serial(5, 1, [11], 6)[0]


def test_partition_layers_balances_bytes():
    from prima_cpp_amd.ring import partition_layers
    lb = [513] * 80
    wins = partition_layers(lb, 862, 8)
    assert wins[0][0] == 0 and wins[-1][1] == 80 and all(a[1] == b[0] for a, b in zip(wins, wins[1:]))
    loads = [sum(lb[lo:hi]) + (862 if i == 0 else 0) for i, (lo, hi) in enumerate(wins)]
    assert max(loads) <= 11 * 513          # 80 layers + head over 8 ranks: nobody gets more than 11 layers' worth
    assert partition_layers([1, 1, 1], 0, 3) == [(0, 1), (1, 2), (2, 3)]


def test_pipelined_prompt_schedule_every_send_meets_its_receive():
    """The prompt pipeline of pm355_ring_prefill (mirrored by ring.prefill_schedule): in every step the number of rows a rank sends equals
    what its successor receives, every chunk visits the ranks in order one step apart, and rank 0 collects exactly one row per prompt."""
    from prima_cpp_amd.ring import prefill_schedule
    for world in (1, 2, 3, 8):
        for n_seq, n_prompt, ubatch in ((1, 5, 512), (2, 11, 4), (3, 2048, 512), (8, 16, 16), (2, 1000, 512)):
            sch = [prefill_schedule(r, world, n_seq, n_prompt, ubatch) for r in range(world)]
            steps = len(sch[0])
            assert all(len(x) == steps for x in sch)
            C = (n_prompt + ubatch - 1) // ubatch
            for s in range(steps):
                for r in range(world):
                    g, T, snd, rcv = sch[r][s]
                    nxt = (r + 1) % world
                    if world > 1:
                        assert snd == sch[nxt][s][3], (world, n_seq, n_prompt, ubatch, s, r)      # same step, same size
                    if g is not None:
                        assert s == g + r and T == (n_prompt - (g % C) * ubatch if g % C == C - 1 else ubatch)
            for r in range(world):
                done = [g for g, *_ in sch[r] if g is not None]
                assert done == list(range(n_seq * C))                                              # every chunk, in order, once
            if world > 1:
                assert sum(x[3] for x in sch[0]) == n_seq                                            # one returned row per prompt
                assert sum(x[1] for x in sch[1]) == n_seq * n_prompt
