"""Prefill GEMM probe: the Llama-3-70B ffn_gate shape (N 28672 x K 8192, Q4_K) and friends at T tokens on the MFMA kernel
(mmq.hip). Prints achieved dense TFLOP/s from HIP events; run under `rocprofv3 --pmc MfmaUtil ...` for the counter passes kept
under profiles/ (PMC_ITERS bounds the launches of a counter pass).
usage: python tools/gemm_probe.py [T=2048] [shape=gate|down|qkv|wo]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemv_bench import P, Q4_K, rand_weight  # noqa: E402
from prima_cpp_amd.lib import Q6_K  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
shape = sys.argv[2] if len(sys.argv) > 2 else "gate"
K, N, t = {"gate": (8192, 28672, Q4_K), "gate6": (8192, 28672, Q6_K), "down": (28672, 8192, Q6_K), "down4": (28672, 8192, Q4_K), "wo": (8192, 8192, Q4_K), "wk": (8192, 1024, Q4_K),
           "qkv": (8192, 10240, Q4_K), "gateq": (8192, 29568, Q6_K)}[shape]   # qkv: wq | wk | wv as one launch of three jobs; gateq: Qwen2.5-72B's ffn_gate | ffn_up as a pair launch
x = torch.randn(T, K, device="cuda") * 0.5
iters = int(os.environ.get("PMC_ITERS", "10"))
if shape == "qkv":
    ws = [rand_weight(Q4_K, K, 8192), rand_weight(Q4_K, K, 1024), rand_weight(Q6_K, K, 1024)]
    run = lambda: P.mul_mat_mfma_multi(ws, x)
elif shape == "gateq":
    wg, wu = rand_weight(t, K, N), rand_weight(t, K, N)
    run = lambda: P.mul_mat_mfma_pair(wg, wu, x)
    N *= 2
else:
    w = rand_weight(t, K, N)
    run = lambda: P.mul_mat_mfma(w, x)
y = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    y = run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"{shape} T={T} N={N} K={K}: {ms * 1e3:.1f} us per call (incl. the f32->f16 conversion of X), {2.0 * T * N * K / ms / 1e9:.1f} TFLOP/s "
      f"= {2.0 * T * N * K / ms / 1e9 / 2500 * 100:.1f} % of the dense F16 MFMA peak")
