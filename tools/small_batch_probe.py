"""Small-batch mat-mul (mmq_i8.hip, 1..32 tokens on the integer matrix cores): time per call on the Llama-3-70B layer shapes against the
mat-vec (1 token), the multi-column mat-vec (<= 8) and the F16 MFMA GEMM (>= 16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemv_bench import P, Q4_K, rand_weight  # noqa: E402
from prima_cpp_amd.lib import Q6_K  # noqa: E402


def timed(fn, n=30):
    for _ in range(3): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = (("ffn_gate", Q4_K, 8192, 28672), ("ffn_down", Q6_K, 28672, 8192), ("attn_q", Q4_K, 8192, 8192), ("attn_k", Q4_K, 8192, 1024))
only = sys.argv[1:] or None
t_list = tuple(int(v) for v in os.environ["PROBE_T"].split(",")) if os.environ.get("PROBE_T") else (1, 2, 4, 8, 16, 32)
small_only = os.environ.get("PROBE_SMALL_ONLY") == "1"
for name, t, K, N in shapes:
    if only and name not in only: continue
    nw = max(2, int(1.2e9 // (K * N * 0.6)))                   # rotate over > 1 GB of weights: no cache reuse
    ws = [rand_weight(t, K, N) for _ in range(min(nw, 8))]
    for T in t_list:
        x = torch.randn(T, K, device="cuda")
        xq = P.quantize_act(x, P.vec_dot_act_type(t))
        us = timed(lambda i: P.mul_mat_small(ws[i % len(ws)], xq=xq, n_tokens=T))
        line = f"{name} {('Q4_K' if t == Q4_K else 'Q6_K')} K={K} N={N} T={T:2d}: small {us:7.1f} us ({ws[0].nbytes / us / 1e3:5.0f} GB/s of weights)"
        if T <= 8 and not small_only:
            line += f" | mat-vec cols {timed(lambda i: P.mul_mat_vec(ws[i % len(ws)], xq=xq, ncols=T)):7.1f} us"
        if T >= 16 and not small_only:
            line += f" | F16 GEMM {timed(lambda i: P.mul_mat_mfma(ws[i % len(ws)], x)):7.1f} us"
        print(line, flush=True)
    del ws
    torch.cuda.empty_cache()
