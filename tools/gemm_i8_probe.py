"""Prompt GEMM on the integer matrix cores (mmq_big.hip) next to the F16 GEMM (mmq.hip) on the Llama-3-70B / Qwen2.5-72B layer shapes.
Times include the activation pass each path needs per call (Q8_K quantization + tables / f32 -> F16 conversion).
usage: python tools/gemm_i8_probe.py [T=2048] [shape ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemv_bench import P, Q4_K, rand_weight  # noqa: E402
from prima_cpp_amd.lib import Q6_K  # noqa: E402

args = [a for a in sys.argv[1:]]
T = int(args[0]) if args else 2048
SHAPES = {"gate": (8192, 28672, Q4_K), "gate6": (8192, 29568, Q6_K), "down": (28672, 8192, Q6_K), "down4": (28672, 8192, Q4_K), "wo": (8192, 8192, Q4_K),
          "wk": (8192, 1024, Q4_K), "wq6": (8192, 8192, Q6_K)}
names = args[1:] or ["gate", "down4", "down", "wo", "wk", "gate6"]
iters = int(os.environ.get("PMC_ITERS", "10"))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name in names:
    K, N, t = SHAPES[name]
    w = rand_weight(t, K, N)
    x = torch.randn(T, K, device="cuda") * 0.5
    fl = 2.0 * T * N * K
    only = os.environ.get("PROBE_ONLY", "")
    ms8 = timed(lambda: P.mul_mat_i8(w, x)) if only != "f16" else float("nan")
    ms16 = timed(lambda: P.mul_mat_mfma(w, x)) if only != "i8" else float("nan")
    print(f"{name:6s} T={T} N={N} K={K} type {t}: int8 {ms8 * 1e3:8.1f} us = {fl / ms8 / 1e9:7.1f} TOP/s | f16 {ms16 * 1e3:8.1f} us = {fl / ms16 / 1e9:7.1f} TFLOP/s", flush=True)
    del w, x
