"""Minimal workload for a rocprofv3 --pmc pass: the dominant kernel only (Q4_K gate/up pair mat-vec at the Llama-3-70B
shape), a handful of eager launches, no hipGraph.
Usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv ... -- python tools/pmc_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemv_bench import P, Q4_K, rand_weight, row_size  # noqa: E402

K, N = 8192, 28672
ws = [rand_weight(Q4_K, K, N) for _ in range(2)]
x = torch.randn(1, K, device="cuda")
nw = torch.ones(K, device="cuda")
for _ in range(int(os.environ.get("PMC_ITERS", "6"))):
    P.mul_mat_vec_fused([ws[0]], x, norm_w=nw, eps=1e-5, w2s=[ws[1]])
torch.cuda.synchronize()
print("algorithmic_bytes_per_launch", 2 * N * row_size(Q4_K, K))
