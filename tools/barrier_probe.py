import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P
lib = P.L.load()
lib.pm355_probe_grid_barrier.restype = C.c_int
lib.pm355_probe_grid_barrier.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_void_p]
torch.zeros(1, device="cuda")
for n in (100, 1000):
    us = C.c_float(0)
    rc = lib.pm355_probe_grid_barrier(n, C.byref(us), P.stream_ptr())
    print("phases", n, "rc", rc, "us per barrier %.2f" % us.value)
