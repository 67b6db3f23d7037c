"""Multi-column mat-vec (2..8 tokens per call: short batches, speculative decoding, parallel sequences): time per call and effective
weight-stream rate on the ffn_gate shape, against the single-column launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemv_bench import P, Q4_K, rand_weight  # noqa: E402
from prima_cpp_amd.lib import Q6_K  # noqa: E402

for t, name in ((Q4_K, "Q4_K"), (Q6_K, "Q6_K")):
    K, N = 8192, 28672
    ws = [rand_weight(t, K, N) for _ in range(6)]                 # rotate over > 1 GB of weights: no cache reuse
    for nc in (1, 2, 3, 4, 8):
        x = torch.randn(nc, K, device="cuda")
        xq = P.quantize_act(x, P.vec_dot_act_type(t))
        for w in ws:
            P.mul_mat_vec(w, xq=xq, ncols=nc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            P.mul_mat_vec(ws[i % 6], xq=xq, ncols=nc)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print(f"{name} N={N} K={K} ncols={nc}: {us:7.1f} us per call, weights at {ws[0].nbytes / us / 1e3:6.0f} GB/s, {us / nc:6.1f} us per column")
