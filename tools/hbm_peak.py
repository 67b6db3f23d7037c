"""HBM read ceiling of this box for the mat-vec's access pattern + a D2D copy for reference.
Usage (GPU box): python tools/hbm_peak.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402


def main():
    lib = P.L.load()
    plib = P.L.load_probe()
    plib.pm355_probe_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    nbytes = 2 << 30
    src = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda")
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for span in (264 << 20, 1 << 30):            # one gate/up pair's worth, and 1 GiB
        for wg in (1, 2):
            for un in (4, 8):
                best = 1e9
                for rep in range(5):
                    off = (rep % 2) * (1 << 30)      # alternate halves: nothing survives in the 256 MB infinity cache
                    e0.record()
                    P.check(plib.pm355_probe_stream_read(src.data_ptr() + off, span, wg, un, sink.data_ptr(), P.stream_ptr()), "probe")
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3)
                print(f"stream_read span={span >> 20:5d} MiB wg/cu={wg} unroll={un}: {best:8.1f} us  {span / best / 1e3:7.1f} GB/s")
    dst = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        e0.record(); dst.copy_(src[:1 << 30]); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3
    print(f"d2d copy 1 GiB: {t:.1f} us  read+write {2 * (1 << 30) / t / 1e3:.1f} GB/s")


if __name__ == "__main__":
    main()
