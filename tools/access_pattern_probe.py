"""What HBM delivers for the access patterns of the long-context decode attention (Llama-3-70B head shape, 32k cells: 64 spans x 8 KV heads =
512 workgroups of 4 waves, 67 MB each for K and V^T), against alternatives that read the same bytes in longer contiguous pieces.
Every pattern reads its bytes exactly once; buffers rotate so nothing is served from the 256 MB infinity cache.   python tools/access_pattern_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402

plib = P.L.load_probe()
plib.pm355_probe_chunk_read.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_int64] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
n_ctx, Hkv, dh, span = 33024, 8, 128, 512
nspan = 64
row = Hkv * dh * 2                     # K row of a cell
vrow = n_ctx * 2                       # V^T row of a dimension
size = n_ctx * row
nbuf = 6
bufs = [torch.randint(0, 255, (size + (1 << 20),), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
# name, nx, wgx, wgy, wave, outer, inner, piece, chunk, n_outer   (strides in bytes; a wave reads n_outer x 8 instructions of 1 KB)
K = 1024
pats = [
    ("K  as is: 256 B of a 2 KB cell row per (cell, head); spans fastest in the grid", nspan, span * row, dh * 2, 32 * row, 128 * row, 4 * row, row, 256, 4),
    ("K  as is, heads fastest in the grid", Hkv, dh * 2, span * row, 32 * row, 128 * row, 4 * row, row, 256, 4),
    ("K  head-major layout [head][cell][128]: a wave tile = 8 KB contiguous", nspan, span * dh * 2, n_ctx * dh * 2, 32 * dh * 2, 128 * dh * 2, K, 0, 1024, 4),
    ("V^T as is: 64 B of a 66 KB row per wave instruction piece (16 rows), 4 waves side by side", nspan, span * 2, dh * vrow, 64, 256, 16 * vrow, vrow, 64, 4),
    ("V^T as is, heads fastest", Hkv, dh * vrow, span * 2, 64, 256, 16 * vrow, vrow, 64, 4),
    ("V^T 256 B per row per instruction (4 rows), one wave per 128 keys", nspan, span * 2, dh * vrow, 256, 0, 4 * vrow, vrow, 256, 4),
    ("V^T tiled layout [head][span of 512][128 dims][512 keys]: 1 KB per row, rows adjacent", nspan, span * dh * 2, n_ctx * dh * 2, 32 * K, 0, 4 * K, 0, 1024, 4),
    ("plain stream: every wave 32 KB contiguous", 512, 4 * 32 * K, 0, 32 * K, 8 * K, K, 0, 1024, 4),
]
for name, nx, wgx, wgy, wave, outer, inner, piece, chunk, n_outer in pats:
    if "V^T 256 B per row" in name:
        # 32 instructions per wave: 4 rows each -> 128 rows = all dims, 256 B = 128 keys; the wave's group of 8 instructions = 32 rows
        outer, inner, n_outer = 32 * vrow, 4 * vrow, 4
    if "tiled layout" in name:
        outer, n_outer = 8 * 4 * K, 4
    best = 1e9
    for rep in range(2 * nbuf):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        P.check(plib.pm355_probe_chunk_read(bufs[rep % nbuf].data_ptr(), 512, nx, wgx, wgy, wave, outer, inner, piece, chunk, n_outer, sink.data_ptr(), P.stream_ptr()), "probe")
        e1.record()
        torch.cuda.synchronize()
        if rep >= nbuf:
            best = min(best, e0.elapsed_time(e1) * 1e3)
    nbytes = 512 * 4 * n_outer * 8 * 1024
    print(f"{name:95s}: {nbytes / 1e6:6.1f} MB in {best:6.1f} us = {nbytes / best / 1e6:5.2f} TB/s")
