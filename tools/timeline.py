"""Per-launch timeline of ONE decoded token from a rocprofv3 kernel_trace.csv: duration of every launch and the idle gap since the
previous launch ended, averaged over the layers of the token (tools/prof_summary.py gives the per-kernel totals).
usage: python tools/timeline.py <kernel_trace.csv> [token_index_from_end=3]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    m = re.search(r"gemv_q_kernel<(\d+), (\d+), (true|false)", n)
    if m:
        return f"gemv<{m.group(1)},{m.group(2)},{'pair' if m.group(3) == 'true' else 'single'}>"
    m2 = re.search(r"(attn_rope_fused_kernel<\d+>|attn_split_\w+_kernel(<\d+>)?|argmax_kernel|embed_rows_kernel|advance_kernel|set_i32\w*)", n)
    return m2.group(1) if m2 else re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:36]


def main(path, back=3):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", ""))) for r in csv.DictReader(open(path))))
    # token boundaries = the argmax launches
    idx = [i for i, r in enumerate(rows) if r[2] == "argmax_kernel"]
    if len(idx) < back + 2:
        print("not enough tokens in the trace"); return
    lo, hi = idx[-back - 1] + 1, idx[-back] + 1
    tok = rows[lo:hi]
    print(f"token window: {len(tok)} launches, {(tok[-1][1] - tok[0][0]) / 1e3:.1f} us wall, {sum(e - s for s, e, _, _ in tok) / 1e3:.1f} us busy")
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    order = []
    prev_end = None
    for s, e, n, g in tok:
        key = (n, g)
        if key not in agg:
            order.append(key)
        a = agg[key]
        a[0] += 1; a[1] += (e - s) / 1e3
        if prev_end is not None:
            a[2] += (s - prev_end) / 1e3
        prev_end = e
    print(f"{'kernel':40s} {'grid':>8s} {'n':>4s} {'avg us':>8s} {'gap before us':>14s} {'us/token':>10s}")
    for key in order:
        c, d, gsum = agg[key]
        print(f"{key[0]:40s} {key[1]:>8s} {c:4d} {d / c:8.2f} {gsum / c:14.2f} {d + gsum:10.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
