"""ncols = 1 / 2 / 4 / 8 mat-vec on the Llama-3-70B ffn_gate shape: time per call and effective weight GB/s per column."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemv_bench import P, Q4_K, Q6_K, rand_weight  # noqa: E402

K, N = 8192, 28672
for t, nm in ((Q4_K, "q4_K"), (Q6_K, "q6_K")):
    ws = [rand_weight(t, K, N) for _ in range(3)]
    for C in (1, 2, 4, 8):
        x = torch.randn(C, K, device="cuda")
        xq = P.quantize_act(x, P.Q8_K)
        for w in ws:
            P.mul_mat_vec(w, xq=xq, ncols=C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        it = 30
        for i in range(it):
            P.mul_mat_vec(ws[i % 3], xq=xq, ncols=C)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        print(f"{nm} ncols={C}: {us:8.1f} us per call, {us / C:7.1f} us per column, weights streamed at {ws[0].nbytes * (1 if C == 1 else 1) / us / 1e3:7.1f} GB/s per call")
    del ws
