"""Per-launch time of the long-context single-token attention over cached cells (attn_flash_mfma.hip) at a model's head shape, with the
KV bytes NOT resident in the 256 MB infinity cache: `nbuf` caches are attended round-robin. Prints us per launch and the F16 KV bytes
over that time (the kernel's roofline is the 8 TB/s stream of K and V^T).     python tools/attn_long_probe.py [H Hkv dh] [cells,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402

TS = "--ts" in sys.argv            # measurement build (python -m prima_cpp_amd.build ts -DPM_TS; PM355_LIB=ab/ts.so): phases inside the launch
argv = [a for a in sys.argv if a != "--ts"]
H, Hkv, dh = (int(x) for x in argv[1:4]) if len(argv) > 3 else (64, 8, 128)
cells_list = [int(x) for x in (argv[4] if len(argv) > 4 else "2048,8192,32768").split(",")]
n_ctx = max(cells_list) + 256
Nkv = Hkv * dh
nbuf = max(2, int(600e6 / (n_ctx * Nkv * 4)) + 1)
g = torch.Generator(device="cuda").manual_seed(5)
kcs = [torch.randn(n_ctx * Nkv, device="cuda", generator=g).to(torch.float16).view(torch.int16) for _ in range(nbuf)]
vcs = [torch.randn(n_ctx * Nkv, device="cuda", generator=g).to(torch.float16).view(torch.int16) for _ in range(nbuf)]
q = torch.randn(1, H * dh, device="cuda", generator=g).to(torch.float16).float()
scratch = P.attn_split_scratch(H, dh, n_ctx)
scale = dh ** -0.5
for cells in cells_list:
    n_kv = cells - 7
    pd = torch.tensor([n_kv - 1], dtype=torch.int32, device="cuda")
    grid = 1024
    while grid < n_kv:
        grid *= 2
    grid = min(grid, n_ctx)
    for i in range(nbuf):
        out = P.attn_cached(q, kcs[i], vcs[i], pd, H, Hkv, dh, n_ctx, scale, max_keys=grid, scratch=scratch)
    torch.cuda.synchronize()
    reps = 10 * nbuf
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        out = P.attn_cached(q, kcs[i % nbuf], vcs[i % nbuf], pd, H, Hkv, dh, n_ctx, scale, max_keys=grid, scratch=scratch)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    kv = 2 * n_kv * Nkv * 2
    print(f"H {H} Hkv {Hkv} dh {dh} n_kv {n_kv:6d}: {us:7.1f} us / launch   KV {kv / 1e6:7.1f} MB -> {kv / us / 1e6:5.2f} TB/s ({kv / us / 1e6 / 8 * 100:4.1f} % of 8 TB/s)")
    if TS:
        import ctypes as C
        import numpy as np
        from prima_cpp_amd import lib as L
        lib = L.load()
        lib.pm355_ts_read.argtypes = [C.c_void_p, C.c_int]
        assert lib.pm355_ts_enable(4) == 0
        for i in range(3):
            lib.pm355_ts_reset()
            out = P.attn_cached(q, kcs[i % nbuf], vcs[i % nbuf], pd, H, Hkv, dh, n_ctx, scale, max_keys=grid, scratch=scratch)
            torch.cuda.synchronize()
        buf = np.zeros((4, 256, 8), dtype=np.uint64)
        assert lib.pm355_ts_read(buf.ctypes.data, 4) == 0
        t = buf.reshape(1024, 8).astype(np.int64)
        t = t[t[:, 7] > 0]
        t0 = t[:, 0].min()
        ph = lambda a, b: f"{(t[:, b] - t[:, a]).mean() / 100:6.2f}"
        last = t[t[:, 6] == 1]
        print(f"   {len(t)} workgroups recorded of {int(t[0, 7])}; us: start spread {(t[:, 0].max() - t0) / 100:.2f} | entry->first tile {ph(0, 1)} | key loop rest {ph(1, 2)} | "
              f"wave merge + publish {ph(2, 3)} | drain + ticket {ph(3, 4)} | non-last exit at {(t[t[:, 6] == 0][:, 5].mean() - t0) / 100 if (t[:, 6] == 0).any() else 0:.2f} | "
              f"last: ticket at {(last[:, 4].mean() - t0) / 100 if len(last) else 0:.2f}, merge {((last[:, 5] - last[:, 4]).mean()) / 100 if len(last) else 0:.2f}, end {(t[:, 5].max() - t0) / 100:.2f}")
        for name, k in (("entry", 0), ("first tile", 1), ("key loop end", 2), ("published", 3), ("ticket", 4), ("exit", 5)):
            v = (t[:, k] - t0) / 100
            print(f"      {name:13s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
        lib.pm355_ts_enable(0)
