"""Decode mat-vec launches of one Llama-3-70B layer exactly as the engine issues them (fused norm / pair / multi-job),
timed with HIP events, rotating over several weight copies so the 256 MB infinity cache cannot serve them.
Usage (GPU box): [PM355_LIB=ab/x.so] python tools/gemv_ab.py [--iters 60]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemv_bench import P, Q4_K, Q5_K, Q6_K, rand_weight  # noqa: E402


def timed(fn, n_copies, iters):
    for i in range(n_copies):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % n_copies)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--E", type=int, default=8192)
    ap.add_argument("--F", type=int, default=28672)
    a = ap.parse_args()
    E, F, KV = a.E, a.F, 1024
    x = torch.randn(1, E, device="cuda")
    xf = torch.randn(1, F, device="cuda")
    nw = torch.ones(E, device="cuda")
    res = torch.randn(E, device="cuda")
    tot = 0.0
    out = []

    def report(name, us, nbytes):
        nonlocal tot
        tot += us
        out.append(f"{name:28s} {nbytes/1e6:7.1f} MB {us:7.2f} us {nbytes/us/1e3:7.1f} GB/s")

    nc = 3
    g = [(rand_weight(Q4_K, E, F), rand_weight(Q4_K, E, F)) for _ in range(nc)]
    us = timed(lambda i: P.mul_mat_vec_fused([g[i][0]], x, norm_w=nw, eps=1e-5, w2s=[g[i][1]]), nc, a.iters)
    report("gate/up pair q4_K +norm", us, 2 * g[0][0].nbytes)
    del g
    for t, nm in ((Q4_K, "q4_K"), (Q6_K, "q6_K")):
        d = [rand_weight(t, F, E) for _ in range(nc)]
        us = timed(lambda i: P.mul_mat_vec_fused([d[i]], xf, resids=[res]), nc, a.iters)
        report(f"down {nm} +resid", us, d[0].nbytes)
        del d
    nq = 6
    for tv, nm in ((Q6_K, "q6_K"), (Q5_K, "q5_K")):
        q = [(rand_weight(Q4_K, E, E), rand_weight(Q4_K, E, KV), rand_weight(tv, E, KV)) for _ in range(nq)]
        us = timed(lambda i: P.mul_mat_vec_fused(list(q[i]), x, norm_w=nw, eps=1e-5), nq, a.iters)
        report(f"qkv q4_K/q4_K/{nm} +norm", us, sum(w.nbytes for w in q[0]))
        del q
    o = [rand_weight(Q4_K, E, E) for _ in range(nq)]
    us = timed(lambda i: P.mul_mat_vec_fused([o[i]], x, resids=[res]), nq, a.iters)
    report("wo q4_K +resid", us, o[0].nbytes)
    print("\n".join(out))
    print(f"layer mat-vec total (q4 down, q6 v): {tot:.1f} us (sum of all rows above)")


if __name__ == "__main__":
    main()
