"""Decode attention at long context on the Llama-3-70B head shape: fused one-workgroup-per-head kernel vs the split path.
Run under rocprofv3 --kernel-trace --stats to see the per-kernel durations."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402

H, Hkv, dh, n_ctx = 64, 8, 128, 4096
q = torch.randn(1, H * dh, device="cuda")
K = (torch.randn(n_ctx, Hkv * dh, device="cuda")).half().view(torch.int16)
V = (torch.randn(Hkv * dh, n_ctx, device="cuda")).half().view(torch.int16)
for n_past in (255, 1023, 3799):
    for name, fn in (("per-head", P.attn_decode), ("split", P.attn_decode_split)):
        for _ in range(3):
            fn(q, K, V, n_past, H, Hkv, dh, n_ctx, 1.0 / np.sqrt(dh))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(q, K, V, n_past, H, Hkv, dh, n_ctx, 1.0 / np.sqrt(dh))
        e1.record()
        torch.cuda.synchronize()
        print(f"n_kv={n_past + 1:5d} {name:9s}: {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us per call (eager, incl. launch gaps)")
