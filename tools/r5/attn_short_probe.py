"""us per launch of pm355_attn_cached (one workgroup per head) against the number of cells attended, 70B head shape, from a captured graph of 80 launches
(the way the decode step issues it).   python tools/r5/attn_short_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import prima_cpp_amd.ops as P
H, Hkv, dh, n_ctx = 64, 8, 128, 4096
g = torch.Generator(device="cuda").manual_seed(1)
kcs = [torch.randn(n_ctx * Hkv * dh, device="cuda", generator=g).half().view(torch.int16) for _ in range(8)]
vcs = [torch.randn(n_ctx * Hkv * dh, device="cuda", generator=g).half().view(torch.int16) for _ in range(8)]
q = torch.randn(1, H * dh, device="cuda", generator=g).half().float()
for n_kv in (16, 64, 65, 96, 128, 192, 256, 384, 512, 640):
    pos = torch.tensor([n_kv - 1], dtype=torch.int32, device="cuda")
    for i in range(8):
        P.attn_cached(q, kcs[i], vcs[i], pos, H, Hkv, dh, n_ctx, dh ** -0.5, max_keys=648)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for i in range(80):
                P.attn_cached(q, kcs[i % 8], vcs[i % 8], pos, H, Hkv, dh, n_ctx, dh ** -0.5, max_keys=648)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"n_kv {n_kv:4d}: {e0.elapsed_time(e1) * 1e3 / 400:6.2f} us per launch (incl. the boundary to the next launch)")
