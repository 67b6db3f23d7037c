"""Co-resident launches vs kernel boundaries on the Llama-3-70B decode layer's byte counts (tools/csrc/overlap_probe.hip).

mode 0: the product's launch shape - one stream, 1024-thread workgroups, 2 register sets, kernel boundaries between dependent launches
mode 1: the same with 512-thread workgroups (what halving the waves per launch costs on its own)
mode 2: launches alternate between two streams, 512 threads / <= 128 VGPRs so that launch k+1 is resident next to launch k, weights
        pre-issued (nset register sets, + lds KiB per wave by LDS-DMA) before a device counter says launch k's outputs are out
Prints microseconds per layer and, from in-kernel s_memrealtime stamps, where the seams are. Usage: python tools/overlap_probe.py [layers]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "prima_cpp_amd", "libprima_mi355_probe.so"))
LP, IP, FP = C.POINTER(C.c_long), C.POINTER(C.c_int), C.POINTER(C.c_float)
lib.pm355_probe_overlap.restype = C.c_int
lib.pm355_probe_overlap.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, LP, LP, IP, IP, IP, FP, C.c_int, C.c_int,
                                    C.POINTER(C.c_float), C.POINTER(C.c_uint), C.c_void_p]

MB = 1 << 20
NAMES = ["QKV", "attn", "wo", "gate/up", "down"]
BYTES = [int(49.6e6), 0, int(37.7e6), int(264.2e6), int(132.1e6)]
BYTES_ALT = [int(51.7e6), 0, int(37.7e6), int(264.2e6), int(192.7e6)]      # layers with Q6_K attn_v / ffn_down
N_IN = [8192, 10240, 8192, 8192, 28672]
N_OUT = [10240, 8192, 8192, 28672, 8192]
NORM = [1, 0, 0, 1, 0]
HOLD = [0.0, 3.0, 0.0, 0.0, 0.0]
G = 256
ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
N_LAYERS = int(ARGS[0]) if ARGS else 16
QUICK = "--quick" in sys.argv


def arr(t, v):
    return (t * len(v))(*v)


def run(mode, nset=2, lds=0, poll=4, nap=0.0, graph=1, reps=20, stamps=False, hold=HOLD, n_layers=N_LAYERS):
    us, err = C.c_float(0), C.c_uint(0)
    ts = np.zeros((n_layers * 5, G, 8), dtype=np.uint64) if stamps else None
    rc = lib.pm355_probe_overlap(mode, nset, lds, poll, nap, n_layers, 5, arr(C.c_long, BYTES), arr(C.c_long, BYTES_ALT), arr(C.c_int, N_IN), arr(C.c_int, N_OUT),
                                 arr(C.c_int, NORM), arr(C.c_float, hold), graph, reps, C.byref(us), C.byref(err),
                                 ts.ctypes.data_as(C.c_void_p) if stamps else None)
    return rc, us.value, err.value, ts


def anatomy(ts, label):
    t = ts.astype(np.int64) * 0.01          # us
    n_k = t.shape[0]
    print(f"  seams of {label} (us, mean over layers; negative start gap = the next launch was resident before this one ended)")
    print(f"  {'launch':8s} {'span':>7s} {'wait':>7s} {'prologue':>8s} {'rows':>7s} {'store':>6s} {'end_spread':>10s} {'next_start-end':>14s} {'lastrow->next_firstrow':>22s}")
    for ph in range(5):
        ks = [k for k in range(ph, n_k - 1, 5) if k >= 5]
        f = lambda fn: float(np.mean([fn(k) for k in ks]))
        span = f(lambda k: t[k, :, 5].max() - t[k, :, 0].min())
        wait = f(lambda k: (t[k, :, 1] - t[k, :, 0]).mean())
        pro = f(lambda k: (t[k, :, 2] - t[k, :, 1]).mean())
        rows = f(lambda k: (t[k, :, 3] - t[k, :, 2]).mean())
        store = f(lambda k: (t[k, :, 5] - t[k, :, 3]).mean())
        spread = f(lambda k: t[k, :, 5].max() - t[k, :, 5].min())
        gap = f(lambda k: t[k + 1, :, 0].min() - t[k, :, 5].max())
        seam = f(lambda k: t[k + 1, :, 2].min() - t[k, :, 3].max())
        print(f"  {NAMES[ph]:8s} {span:7.2f} {wait:7.2f} {pro:8.2f} {rows:7.2f} {store:6.2f} {spread:10.2f} {gap:14.2f} {seam:22.2f}")
    print(f"  {'':8s} {'dep_seen spread':>16s} {'prologue_end spread':>20s} {'rows_end spread':>16s} {'rows min':>9s} {'rows max':>9s} {'dep_seen - prev_end':>20s}")
    for ph in range(5):
        ks = [k for k in range(ph, n_k - 1, 5) if k >= 5]
        f = lambda fn: float(np.mean([fn(k) for k in ks]))
        print(f"  {NAMES[ph]:8s} {f(lambda k: t[k, :, 1].max() - t[k, :, 1].min()):16.2f} {f(lambda k: t[k, :, 2].max() - t[k, :, 2].min()):20.2f} "
              f"{f(lambda k: t[k, :, 3].max() - t[k, :, 3].min()):16.2f} {f(lambda k: (t[k, :, 3] - t[k, :, 2]).min()):9.2f} {f(lambda k: (t[k, :, 3] - t[k, :, 2]).max()):9.2f} "
              f"{f(lambda k: t[k, :, 1].min() - t[k - 1, :, 5].max()):20.2f}")
    tot = (t[n_k - 1, :, 5].max() - t[5, :, 0].min()) / ((n_k - 5) / 5)
    print(f"  layer (first entry of layer 1 -> last exit): {tot:.2f} us")


layer_mb = (sum(BYTES) + sum(BYTES_ALT)) / 2 / 1e6
print(f"Llama-3-70B Q4_K_M layer: {layer_mb:.1f} MB, {N_LAYERS} layers per graph; 8 TB/s = {layer_mb / 8e6 * 1e6:.1f} us per layer")
rows = []
CONFIGS = [
    ("mode 0  one stream, 1024 thr, 2 sets, boundaries", dict(mode=0)),
    ("mode 1  one stream,  512 thr, 2 sets, boundaries", dict(mode=1, nset=2)),
    ("mode 1  one stream,  512 thr, 4 sets, boundaries", dict(mode=1, nset=4)),
    ("mode 2  two streams, 512 thr, 2 sets", dict(mode=2, nset=2)),
    ("mode 2  two streams, 512 thr, 3 sets", dict(mode=2, nset=3)),
    ("mode 2  two streams, 512 thr, 4 sets", dict(mode=2, nset=4)),
    ("mode 2  two streams, 512 thr, 4 sets + 3 KiB/wave LDS", dict(mode=2, nset=4, lds=3)),
    ("mode 2  two streams, 512 thr, 4 sets + 5 KiB/wave LDS", dict(mode=2, nset=4, lds=5)),
    ("mode 2  two streams, 512 thr, 2 sets + 5 KiB/wave LDS", dict(mode=2, nset=2, lds=5)),
    ("mode 2  two streams, 512 thr, 4 sets, EAGER launches", dict(mode=2, nset=4, graph=0)),
    ("mode 0  EAGER launches", dict(mode=0, graph=0)),
    ("mode 2  4 sets, no attention hold", dict(mode=2, nset=4, hold=[0.0] * 5)),
    ("mode 0  no attention hold", dict(mode=0, hold=[0.0] * 5)),
]
if QUICK:
    CONFIGS = [("mode 0  one stream, 1024 thr, 2 sets, boundaries", dict(mode=0))]
    for ns in (2, 3, 4):
        for ob in (0, 16):
            CONFIGS.append((f"mode 0  1024 thr, {ns} sets pre-issued, order barrier {ob // 16}", dict(mode=0, nset=ns + ob)))
    CONFIGS += [("mode 2  eager, 4 sets, poll 16 nap 0.8", dict(mode=2, nset=4, graph=0, poll=16, nap=0.8))]
for name, kw in CONFIGS:
    best = None
    for _ in range(3):
        rc, us, err, _ts = run(**kw)
        if rc or err:
            best = (None, rc, err)
            break
        best = us if best is None else min(best, us)
    if isinstance(best, tuple):
        print(f"{name:58s}  FAILED rc {best[1]} err {best[2]} (1 = spin timeout, 2 = stale activation)")
    else:
        print(f"{name:58s} {best:8.2f} us per layer = {layer_mb / best:.3f} TB/s", flush=True)
for label, kw in [("mode 0", dict(mode=0)), ("mode 0, 4 sets", dict(mode=0, nset=4)), ("mode 0, 4 sets, order barrier", dict(mode=0, nset=20)),
                  ("mode 0, 3 sets, order barrier", dict(mode=0, nset=19))]:
    rc, us, err, ts = run(stamps=True, reps=3, **kw)
    if rc or err:
        print(label, "FAILED", rc, err)
        continue
    anatomy(ts, f"{label} ({us:.2f} us per layer with stamps)")
