"""What does ONE dependent kernel node cost on this box? A chain of 1-thread kernels (pm355_set_i32x2) launched (a) eagerly on a
stream, (b) from a captured hipGraph; run under `rocprofv3 --kernel-trace` and read the kernel durations / gaps, and (c) wall time per
node of the replayed graph with HIP events (no profiler)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402

lib = P.L.load()
lib.pm355_set_i32x2.restype = C.c_int
lib.pm355_set_i32x2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
buf = torch.zeros(64, dtype=torch.int32, device="cuda")
N = 400
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    st = torch.cuda.current_stream().cuda_stream
    for i in range(N):                                  # (a) eager chain
        lib.pm355_set_i32x2(buf.data_ptr(), i, i, st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for i in range(N):
            lib.pm355_set_i32x2(buf.data_ptr(), i, i, torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(side); g.replay(); e1.record(side); torch.cuda.synchronize()
    print(f"hipGraph replay of {N} dependent 1-thread kernels: {e0.elapsed_time(e1) * 1e3 / N:.2f} us per node (HIP events)")
    e0.record(side)
    for i in range(N):
        lib.pm355_set_i32x2(buf.data_ptr(), i, i, st)
    e1.record(side); torch.cuda.synchronize()
    print(f"eager launches of {N} dependent 1-thread kernels: {e0.elapsed_time(e1) * 1e3 / N:.2f} us per launch (HIP events, host-bound if > GPU time)")
