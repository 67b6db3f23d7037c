"""Does the 256 MB infinity cache (MALL) serve the mat-vec's non-temporal streaming loads? Reads a 36 MiB span
(a) rotating through 4 GiB (always HBM) and (b) the same span again and again (MALL-resident if it allocates)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402

lib = P.L.load()
plib = P.L.load_probe()
plib.pm355_probe_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
total = 4 << 30
src = torch.empty(total, dtype=torch.uint8, device="cuda")
src.random_(0, 255)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(off, span, wg, unroll):
    e0.record()
    P.check(plib.pm355_probe_stream_read(src.data_ptr() + off, span, wg, unroll, sink.data_ptr(), P.stream_ptr()), "probe")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


# (c) prefetch experiments: a span is first read by a PARTIAL grid (192 workgroups, as if the other 64 CUs ran attention) with
# plain or nt loads, then read by the full mat-vec-like grid with nt loads: is the second read faster than a cold one?
for span_mb in (36, 64):
    span = span_mb << 20
    for pre, name in ((-8, "plain"), (8, "nt")):
        cold, warm, pf = [], [], []
        for rep in range(20):
            off = ((2 * rep) * span) % (total - 2 * span)
            cold.append(timed(off, span, 1, 8))
            off += span
            pf.append(timed(off, span, -192, pre))
            warm.append(timed(off, span, 1, 8))
        med = lambda v: sorted(v[4:])[len(v[4:]) // 2]
        print(f"span {span_mb:3d} MiB prefetch({name:5s}, 192 WGs) {med(pf):6.2f} us; nt read cold {med(cold):6.2f} us -> after prefetch {med(warm):6.2f} us")

for span_mb in (36, 128):
    span = span_mb << 20
    for mode in ("rotate", "same"):
        ts = []
        for rep in range(24):
            off = ((rep * span) % (total - span)) if mode == "rotate" else 0
            e0.record()
            P.check(plib.pm355_probe_stream_read(src.data_ptr() + off, span, 1, 8, sink.data_ptr(), P.stream_ptr()), "probe")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[4:])
        print(f"span {span_mb:4d} MiB {mode:7s}: median {ts[len(ts)//2]:7.2f} us  min {ts[0]:7.2f} us  -> {span / ts[len(ts)//2] / 1e3:7.1f} GB/s")
