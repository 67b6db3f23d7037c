"""Round 5: what a seam of the persistent loader / consumer layer (csrc/engine_probe.hip) costs UNDER LOAD, and which part of it.
Same skeleton and byte counts as tools/engine_probe.py (Llama-3-70B Q4_K_M layer, 125 fills of 16 KiB per CU); the layer's weights are cut
into 5 / 3 / 2 / 1 phases (= seams per layer) and the seam is thinned (tiny activation vector = barrier only; tiny output = no store burst):
the slope over the seam count is the exposed cost per seam, the intercept the streaming floor with consumers."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402

plib = P.L.load_probe()
IP = C.POINTER(C.c_int)
plib.pm355_probe_engine.restype = C.c_int
plib.pm355_probe_engine.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, IP, IP, IP, C.c_int, C.c_float, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), IP, C.c_void_p]
CUS = torch.cuda.get_device_properties(0).multi_processor_count
FILL = 16384
TOT = 125
layer_bytes = TOT * FILL * CUS
N_REG, N_LAYERS, ACT_STRIDE = 8, 16, 28672
w = torch.empty(layer_bytes * N_REG, dtype=torch.uint8, device="cuda")
w.random_(0, 255)
act = torch.rand(N_LAYERS * 8 * ACT_STRIDE + 64, dtype=torch.float32, device="cuda")
ctr = torch.zeros((33 * 128 + 64) // 4, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()


def arr(v):
    return (C.c_int * len(v))(*v)


def run(ch, act_n, out_n, nw=16, ns=8, nt=1, thin=3):
    us, err = C.c_float(0), C.c_int(0)
    best = 1e30
    for _ in range(3):
        rc = plib.pm355_probe_engine(w.data_ptr(), layer_bytes, N_REG, N_LAYERS, len(ch), arr(ch), arr(act_n), arr(out_n), -1, 0.0,
                                    act.data_ptr(), ACT_STRIDE, ctr.data_ptr(), nw, ns, nt, thin, C.byref(us), C.byref(err), P.stream_ptr())
        if rc or err.value:
            return float("nan")
        best = min(best, us.value)
    return best / N_LAYERS


splits = {
    "5 seams (QKV | - | wo | gate/up | down)": [12, 0, 9, 64, 40],
    "4 seams (QKV | wo | gate/up | down)": [12, 9, 64, 40],
    "3 seams (QKV+wo | gate/up | down)": [21, 64, 40],
    "2 seams (attention half | ffn half)": [21, 104],
    "1 seam": [125],
}
print(f"CUs {CUS}; layer {layer_bytes / 1e6:.1f} MB; floor at 6.8 TB/s = {layer_bytes / 6.8e6:.1f} us")
for nw in (16, 8):
    for name, ch in splits.items():
        n = len(ch)
        full = run(ch, [8192 if i < n - 1 or n == 1 else 28672 for i in range(n)], [8192] * n, nw=nw)
        # the real vector sizes of the 70B layer where the split has them
        if n == 5:
            real = run(ch, [8192, 0, 8192, 8192, 28672], [10240, 8192, 8192, 28672, 8192], nw=nw)
        elif n == 4:
            real = run(ch, [8192, 8192, 8192, 28672], [10240, 8192, 28672, 8192], nw=nw)
        else:
            real = float("nan")
        tiny_act = run(ch, [256] * n, [8192] * n, nw=nw)
        tiny_both = run(ch, [256] * n, [256] * n, nw=nw)
        print(f"nw {nw:2d} {name:42s}: 8192-vectors {full:7.2f} | real sizes {real:7.2f} | barrier + stores only {tiny_act:7.2f} | barrier only {tiny_both:7.2f}  us per layer")
