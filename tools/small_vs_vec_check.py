"""mmq_i8 against the mat-vec on the engine's own shapes (same integers, f32 order differs): max relative error and NMSE."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemv_bench import P, Q4_K, rand_weight  # noqa: E402
from prima_cpp_amd.lib import Q6_K  # noqa: E402

torch.manual_seed(3)
for t, K, N in ((Q4_K, 8192, 8192), (Q4_K, 28672, 8192), (Q6_K, 28672, 8192), (Q4_K, 8192, 28672)):
    w = rand_weight(t, K, N)
    for T in (3, 8):
        x = torch.randn(T, K, device="cuda")
        xq = P.quantize_act(x, P.vec_dot_act_type(t))
        resid = torch.randn(T, N, device="cuda")
        a = P.mul_mat_small(w, xq=xq, n_tokens=T, resid=resid)
        b = P.mul_mat_vec(w, xq=xq, ncols=T, resid=resid)
        d = (a - b).double()
        print(f"type {t} K={K} N={N} T={T}: max |diff| {d.abs().max().item():.3e} (|y| max {b.abs().max().item():.1f}), NMSE {(d.pow(2).sum() / b.double().pow(2).sum()).item():.2e}, "
              f"elements off by > 1e-3 relative: {int(((d.abs() > 1e-3 * b.abs().clamp(min=1.0))).sum().item())}", flush=True)
