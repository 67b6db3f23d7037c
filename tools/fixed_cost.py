"""Fixed (size-independent) cost of a decode mat-vec launch: tiny-N launches of each flavour, to be run under
rocprofv3 --kernel-trace (kernel durations in dispatch order; 20 launches per flavour).
  rocprofv3 --kernel-trace --output-format csv -d out -o fc -- python tools/fixed_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemv_bench import P, Q4_K, Q6_K, rand_weight  # noqa: E402

E, F = 8192, 28672
x = torch.randn(1, E, device="cuda")
xf = torch.randn(1, F, device="cuda")
nw = torch.ones(E, device="cuda")
res = torch.randn(E, device="cuda")
flavours = []
for N in (256, 4096):
    w = rand_weight(Q4_K, E, N)
    w2 = rand_weight(Q4_K, E, N)
    wd = rand_weight(Q4_K, F, N)
    w6 = rand_weight(Q6_K, E, N)
    xq = P.quantize_act(x, P.Q8_K)
    flavours += [
        (f"N={N} K=8192 q4 f32+norm", lambda w=w: P.mul_mat_vec_fused([w], x, norm_w=nw, eps=1e-5)),
        (f"N={N} K=8192 q4 f32", lambda w=w: P.mul_mat_vec_fused([w], x)),
        (f"N={N} K=8192 q4 preq", lambda w=w, xq=xq: P.mul_mat_vec(w, xq=xq)),
        (f"N={N} K=8192 q4 pair+norm", lambda w=w, w2=w2: P.mul_mat_vec_fused([w], x, norm_w=nw, eps=1e-5, w2s=[w2])),
        (f"N={N} K=28672 q4 f32", lambda wd=wd: P.mul_mat_vec_fused([wd], xf)),
        (f"N={N} K=8192 q6 f32+norm", lambda w6=w6: P.mul_mat_vec_fused([w6], x, norm_w=nw, eps=1e-5)),
    ]
torch.cuda.synchronize()
print("FLAVOURS", [f[0] for f in flavours])
for name, fn in flavours:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
