"""Micro-benchmark of the batch-1 quantized mat-vec at the Llama-3-70B / 8B shapes (achieved GB/s).
Usage (GPU box): python tools/gemv_bench.py [--iters 50]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402
from prima_cpp_amd.lib import Q4_K, Q5_K, Q6_K, Q8_0, row_size  # noqa: E402

NAMES = {Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", Q8_0: "q8_0"}


def rand_weight(t, K, N):
    """random bytes with sane fp16 scales, generated on device"""
    rb = row_size(t, K)
    raw = torch.randint(0, 256, (N, rb), dtype=torch.uint8, device="cuda")
    nper, bs = {Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210), Q8_0: (32, 34)}[t]
    nb = K // nper
    v = raw.view(N, nb, bs)
    h = torch.tensor([0x00, 0x1C], dtype=torch.uint8, device="cuda")   # fp16 0x1C00 = 2^-8
    if t == Q6_K:
        v[:, :, 208:210] = h
    elif t == Q8_0:
        v[:, :, 0:2] = h
    else:
        v[:, :, 0:2] = h
        v[:, :, 2:4] = h
    w = P.QWeight(t, K, N, raw.view(-1))
    if t in (Q4_K, Q6_K, Q8_0):
        dst = torch.empty(N * P.L.load().pm355_row_stride(t, K), dtype=torch.uint8, device="cuda")
        P.check(P.L.load().pm355_repack_rows(t, raw.data_ptr(), dst.data_ptr(), K, N, 1, P.stream_ptr()), "repack")
        w = P.QWeight(t, K, N, dst)
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    shapes = [("70B wq", 8192, 8192), ("70B wk", 8192, 1024), ("70B gate", 8192, 28672), ("70B down", 28672, 8192),
              ("70B lm_head", 8192, 128256), ("8B wq", 4096, 4096), ("8B gate", 4096, 14336), ("8B down", 14336, 4096)]
    for name, K, N in shapes:
        for t in (Q4_K, Q6_K, Q5_K, Q8_0):
            if t in (Q5_K, Q8_0) and "wq" not in name and "down" not in name:
                continue
            w = rand_weight(t, K, N)
            # rotate over several copies so the 256 MB infinity cache cannot hold the weights
            copies = [w] + [P.QWeight(t, K, N, w.data.clone()) for _ in range(max(0, min(7, int(600e6 // w.nbytes))))]
            x = torch.randn(1, K, device="cuda")
            xq = P.quantize_act(x, P.vec_dot_act_type(t))
            for c in copies:
                P.mul_mat_vec(c, xq=xq)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.iters):
                P.mul_mat_vec(copies[i % len(copies)], xq=xq)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            print(f"{name:12s} {NAMES[t]:5s} K={K:6d} N={N:6d} {w.nbytes/1e6:8.1f} MB  {us:8.1f} us  {w.nbytes/us/1e3:7.1f} GB/s"
                  f"  ({len(copies)} copies)", flush=True)
            del copies, w


if __name__ == "__main__":
    main()
