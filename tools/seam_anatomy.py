"""Seam anatomy of a decoded token: where the time between "the previous launch stopped streaming" and "this launch streams" goes.

Needs the measurement build of the library (s_memrealtime stamps inside the mat-vec and the single-token attention kernels,
csrc/ts.hip):   python -m prima_cpp_amd.build ts -DPM_TS   then on the GPU box   PM355_LIB=ab/ts.so python tools/seam_anatomy.py

Every workgroup's thread 0 records the chip-wide 100 MHz counter at: entry | after the rms-norm barrier | activations quantized in LDS |
its wave finished its rows | all 16 waves finished | outputs stored. The tool replays the captured single-token graph a few times and
prints, per launch kind and averaged over the layers, the launch's span, how long its workgroups take to start, the prologue, the
row phase, the spread of the finish times and the gap to the next launch's first workgroup (= the kernel boundary)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.engine as E  # noqa: E402
from prima_cpp_amd import lib as L  # noqa: E402

WGS = 256


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-70b")
    ap.add_argument("--layers", type=int, default=0, help="0 = all")
    ap.add_argument("--pos", type=int, default=24, help="position of the analysed token")
    ap.add_argument("--replays", type=int, default=6)
    a = ap.parse_args()
    lib = L.load()
    if not hasattr(lib, "pm355_ts_enable"):
        sys.exit("this library was not built with -DPM_TS: python -m prima_cpp_amd.build ts -DPM_TS; PM355_LIB=ab/ts.so")
    lib.pm355_ts_read.argtypes = [C.c_void_p, C.c_int]
    hp = dict({"llama3-70b": E.LLAMA3_70B, "llama3-8b": E.LLAMA3_8B, "qwen2.5-72b": E.QWEN25_72B}[a.model])
    mix = E.q6_k_types if a.model.startswith("qwen") else E.q4_k_m_types
    full_layers = hp["n_layer"]
    if a.layers:
        hp["n_layer"] = a.layers
    win = E.Window(hp, n_ctx=4096)
    if a.layers:                      # keep the type mixture of the full model's first layers
        win.hp_dict = dict(hp, n_layer=full_layers)
    win.fill_synthetic(mix, seed=1234)
    win.finalize(max_tokens=1)
    tok = torch.zeros(1, dtype=torch.int32, device="cuda")
    am = torch.zeros(1, dtype=torch.int32, device="cuda")
    win.set_pos(0)
    for _ in range(a.pos):            # fill the cache up to the analysed position without instrumentation (un-captured)
        win.step(token=tok, argmax=am, use_graph=False)
    torch.cuda.synchronize()
    n_slots = 8 * hp["n_layer"] + 16
    assert lib.pm355_ts_enable(n_slots) == 0
    lib.pm355_ts_reset()
    # the first graph step captures (every launch function draws its slot during the capture), later steps replay the same slots
    for _ in range(a.replays):
        win.set_pos(a.pos)
        win.step(token=tok, argmax=am, use_graph=True)
    torch.cuda.synchronize()
    used = lib.pm355_ts_used()
    buf = np.zeros((used, WGS, 8), dtype=np.uint64)
    assert lib.pm355_ts_read(buf.ctypes.data, used) == 0
    t = buf.astype(np.int64)
    rows = []
    for s in range(used):
        n = int(t[s, 0, 7])
        if n <= 0:
            continue
        n = min(n, WGS)
        w = t[s, :n]
        tag = int(w[0, 6])
        kind = "attention" if (tag & 15) == 2 else ("gate/up pair" if tag & 16 else f"matvec K={tag >> 12} x{(tag >> 8) & 3}")
        rows.append(dict(slot=s, kind=kind, n=n, t0=w[:, 0], t1=w[:, 1], t2=w[:, 2], t3=w[:, 3], t4=w[:, 4], t5=w[:, 5]))
    us = lambda ticks: float(ticks) / 100.0
    # name the launches by their place in the layer: QKV, attention, wo, gate/up, down
    names = ["QKV", "attention", "wo", "gate/up", "down"]
    per = {}
    n_l = hp["n_layer"]
    for i, r in enumerate(rows[: 5 * n_l]):
        r["name"] = names[i % 5]
    for i, r in enumerate(rows[5 * n_l:]):
        r["name"] = "lm_head" if i == 0 else f"tail{i}"
    for i, r in enumerate(rows):
        nxt = rows[i + 1] if i + 1 < len(rows) else None
        d = per.setdefault(r["name"] + (" (K=28672 Q6_K)" if False else ""), [])
        d.append(dict(
            span=us(r["t5"].max() - r["t0"].min()),
            start_spread=us(r["t0"].max() - r["t0"].min()),
            prologue=us((r["t2"] - r["t0"]).mean()),
            norm=us((r["t1"] - r["t0"]).mean()) if r["t1"].max() > 0 else 0.0,
            rows_wave0=us((r["t3"] - r["t2"]).mean()),
            rows_all=us((r["t4"] - r["t2"]).mean()),
            store=us((r["t5"] - r["t4"]).mean()),
            end_spread=us(r["t5"].max() - r["t5"].min()),
            first_end=us(r["t5"].min() - r["t0"].min()),
            gap_next=us(nxt["t0"].min() - r["t5"].max()) if nxt else 0.0,
            stream_window=us(r["t4"].max() - r["t2"].min()),
        ))
    cols = ["span", "start_spread", "prologue", "norm", "rows_all", "store", "end_spread", "gap_next", "stream_window"]
    print(f"model {a.model}, {n_l} layers, position {a.pos}; microseconds, mean over the layers (s_memrealtime, 10 ns ticks)")
    print(f"{'launch':12s} {'n':>4s} " + " ".join(f"{c:>13s}" for c in cols))
    tot = 0.0
    for name in names + ["lm_head"]:
        if name not in per:
            continue
        d = per[name]
        m = {c: np.mean([x[c] for x in d]) for c in cols}
        print(f"{name:12s} {len(d):4d} " + " ".join(f"{m[c]:13.2f}" for c in cols))
        if name in names:
            tot += m["span"] + m["gap_next"]
    print(f"layer = sum(span + gap_next) = {tot:.1f} us")
    if len(rows) >= 5 * n_l:
        tk = us(rows[-1]["t5"].max() - rows[0]["t0"].min())
        print(f"token (first launch entry -> last launch exit): {tk:.1f} us")
    # who finishes late? finish time of the row phase (t4) relative to the launch's first t2, per workgroup, averaged over the layers:
    # by XCD (workgroup id % 8: consecutive workgroups go to consecutive XCDs) and the slowest / fastest workgroups
    for name in ("gate/up", "down", "QKV", "wo"):
        sel = [r for r in rows if r.get("name") == name and r["n"] == WGS]
        if not sel:
            continue
        rel = np.mean([(r["t4"] - r["t2"].min()) / 100.0 for r in sel], axis=0)          # [256] us
        ent = np.mean([(r["t0"] - r["t0"].min()) / 100.0 for r in sel], axis=0)
        by_xcd = [rel[x::8].mean() for x in range(8)]
        by_xcd_ent = [ent[x::8].mean() for x in range(8)]
        order = np.argsort(rel)
        print(f"{name:8s} row-phase finish by XCD (us): " + " ".join(f"{v:6.2f}" for v in by_xcd) + f" | entry by XCD: " + " ".join(f"{v:5.2f}" for v in by_xcd_ent))
        print(f"{'':8s} mean {rel.mean():.2f} min {rel.min():.2f} max {rel.max():.2f}; slowest workgroups {order[-8:].tolist()} fastest {order[:8].tolist()}")
        # is it the same workgroups every layer? correlation of per-workgroup finish between two halves of the layers
        h = len(sel) // 2
        a_ = np.mean([(r["t4"] - r["t2"].min()) / 100.0 for r in sel[:h]], axis=0); b_ = np.mean([(r["t4"] - r["t2"].min()) / 100.0 for r in sel[h:]], axis=0)
        print(f"{'':8s} correlation of per-workgroup finish times, first vs second half of the layers: {np.corrcoef(a_, b_)[0, 1]:.2f}")
    # split the two kinds of `down` (Q6_K on more-bits layers)
    if "down" in per:
        sp = np.array([x["span"] for x in per["down"]])
        print(f"down spans: min {sp.min():.1f} median {np.median(sp):.1f} max {sp.max():.1f}")
    win.close()


if __name__ == "__main__":
    main()
