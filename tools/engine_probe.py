"""Ceiling of a persistent decode layer on a run-ahead LDS-DMA weight loader (csrc/engine_probe.hip): Llama-3-70B Q4_K_M byte counts
per phase (QKV, attention, wo, gate/up, down), real seams (write-through outputs, device-wide barrier, every workgroup re-reads and
re-quantizes the activation vector), stand-in consumer arithmetic. Prints microseconds per layer for a sweep of engine geometries,
next to the same bytes streamed by the mat-vec's load pattern in one launch (pm355_probe_stream_read). The 5-launch path measures
107 us per layer (profiles/r02_decode_summary.txt)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.ops as P  # noqa: E402

lib = P.L.load()
plib = P.L.load_probe()
IP = C.POINTER(C.c_int)
plib.pm355_probe_engine.restype = C.c_int
plib.pm355_probe_engine.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, IP, IP, IP, C.c_int, C.c_float, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), IP, C.c_void_p]
plib.pm355_probe_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

CUS = torch.cuda.get_device_properties(0).multi_processor_count
FILL = 16384
# Llama-3-70B Q4_K_M, per CU fills of 16 KiB: wq+wk+wv 47 MB, attention 0, wo 37.7 MB, gate+up 264 MB, down 132 (Q4_K) / 193 MB (Q6_K)
CH = [12, 0, 9, 64, 40]
ACT = [8192, 0, 8192, 8192, 28672]
OUT = [10240, 8192, 8192, 28672, 8192]
layer_bytes = sum(CH) * FILL * CUS
N_REG = 8
N_LAYERS = 16
ACT_STRIDE = 28672

w = torch.empty(layer_bytes * N_REG, dtype=torch.uint8, device="cuda")
w.random_(0, 255)
act = torch.rand(N_LAYERS * len(CH) * ACT_STRIDE + 64, dtype=torch.float32, device="cuda")
ctr = torch.zeros((33 * 128 + 64) // 4, dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()


def arr(v):
    return (C.c_int * len(v))(*v)


def run(ch, act_n, out_n, attn_ph, attn_us, n_layers, nw, ns, nt, thin, region_stride=layer_bytes, n_reg=N_REG):
    us, err = C.c_float(0), C.c_int(0)
    best = 1e30
    for _ in range(3):
        rc = plib.pm355_probe_engine(w.data_ptr(), region_stride, n_reg, n_layers, len(ch), arr(ch), arr(act_n), arr(out_n), attn_ph, attn_us,
                                    act.data_ptr(), ACT_STRIDE, ctr.data_ptr(), nw, ns, nt, thin, C.byref(us), C.byref(err), P.stream_ptr())
        if rc or err.value:
            return None, (rc, err.value)
        best = min(best, us.value)
    return best, None


def stream_ref(nbytes):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(6):
        off = (rep % (N_REG - 1)) * layer_bytes
        e0.record()
        P.check(plib.pm355_probe_stream_read(w.data_ptr() + off, nbytes, 1, 8, sink.data_ptr(), P.stream_ptr()), "probe")
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts[2:])


print(f"CUs {CUS}; layer = {layer_bytes / 1e6:.1f} MB in {sum(CH)} fills per CU")
t = stream_ref(layer_bytes)
print(f"one-launch nt stream of one layer's bytes: {t:.1f} us = {layer_bytes / t / 1e3:.0f} GB/s")

# (1) the loader engine alone: one phase, no seams
for nw, ns, nt, thin in ((8, 8, 1, 2), (8, 8, 1, 3), (8, 8, 0, 3), (4, 8, 1, 3), (16, 8, 1, 3), (8, 4, 1, 3)):
    n = sum(CH) * 4
    us, e = run([n], [8192], [8192], -1, 0.0, 1, nw, ns, nt, thin, region_stride=n * FILL * CUS, n_reg=2)
    tag = f"nw {nw:2d} ns {ns} nt {nt} thin {thin}"
    if us is None:
        print(f"stream-only {tag}: FAILED {e}")
    else:
        print(f"stream-only {tag}: {us:8.1f} us for {n * FILL * CUS / 1e6:.0f} MB = {n * FILL * CUS / us / 1e3:6.0f} GB/s")

# (2) the layer with its seams
for nw, ns, nt, thin, attn_us in ((8, 8, 1, 3, 4.0), (8, 8, 1, 3, 0.0), (8, 7, 1, 3, 4.0), (8, 8, 1, 2, 4.0), (8, 8, 0, 3, 4.0),
                                 (4, 8, 1, 3, 4.0), (16, 8, 1, 3, 4.0), (8, 4, 1, 3, 4.0), (8, 2, 1, 3, 4.0)):
    us, e = run(CH, ACT, OUT, 1, attn_us, N_LAYERS, nw, ns, nt, thin)
    tag = f"nw {nw:2d} ns {ns} nt {nt} thin {thin} attn {attn_us:.0f} us"
    if us is None:
        print(f"layer {tag}: FAILED {e}")
    else:
        print(f"layer {tag}: {us / N_LAYERS:7.2f} us per layer = {layer_bytes * N_LAYERS / us / 1e3:6.0f} GB/s")

# (3) seams only: no weights at all (what five seams + the attention stand-in cost by themselves)
us, e = run([0, 0, 0, 0, 0], ACT, OUT, 1, 4.0, N_LAYERS, 8, 8, 1, 3)
print("seams only (no weights):", "FAILED %s" % (e,) if us is None else "%.2f us per layer" % (us / N_LAYERS))
us, e = run([0, 0, 0, 0, 0], ACT, OUT, 1, 0.0, N_LAYERS, 8, 8, 1, 3)
print("seams only, no attention hold:", "FAILED %s" % (e,) if us is None else "%.2f us per layer" % (us / N_LAYERS))
