"""Host -> HBM staging rate of upload.hip (pinned ring + copier threads + private stream) against a plain pageable hipMemcpy, from an
mmap'd file in the page cache (what the reference's loader hands to set_tensor) and from anonymous memory."""
import ctypes as C
import mmap
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402

lib = P.L.load()
N = 2 << 30
path = "/tmp/upload_probe.bin"
src = np.random.default_rng(1).integers(0, 256, N, dtype=np.uint8)
src.tofile(path)
dst = torch.zeros(N, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
fd = os.open(path, os.O_RDONLY)


def fresh_map():
    m = mmap.mmap(fd, N, prot=mmap.PROT_READ)        # a new mapping: page faults are paid again, as in a model load
    return m, np.frombuffer(m, dtype=np.uint8)


for name in ("anonymous", "mmap (page cache)"):
    for mode in ("hipMemcpy pageable", "uploader t=0", "uploader t=2", "uploader t=4", "uploader t=8", "uploader t=4 Q4_K repack"):
        if name == "anonymous":
            m, a = None, src
        else:
            m, a = fresh_map()
        t0 = time.perf_counter()
        if mode.startswith("hipMemcpy"):
            P.check(lib.pm355_memcpy_h2d(dst.data_ptr(), a.ctypes.data, N, None), "h2d")
            P.check(lib.pm355_sync(None), "sync")
        else:
            th = int(mode.split("t=")[1].split()[0])
            up = lib.pm355_uploader_new(0, th)
            t0 = time.perf_counter()
            if "Q4_K" in mode:
                K = 8192
                rb = lib.pm355_row_size(12, K)
                n = (N // rb) * rb
                P.check(lib.pm355_upload(up, 12, K, a.ctypes.data, dst.data_ptr(), n, 1), "upload")
            else:
                P.check(lib.pm355_upload(up, -1, 0, a.ctypes.data, dst.data_ptr(), N, 0), "upload")
            P.check(lib.pm355_uploader_sync(up), "sync")
        dt = time.perf_counter() - t0
        if not mode.startswith("hipMemcpy"):
            lib.pm355_uploader_free(up)
        print(f"{name:18s} {mode:28s} {N / dt / 1e9:6.1f} GB/s")
        del a
        if m is not None:
            m.close()
os.close(fd)
os.unlink(path)
print("cpus:", len(os.sched_getaffinity(0)))
