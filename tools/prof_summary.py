"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid, LDS) count / avg / min / total, plus ms per token."""
import csv
import re
import sys
from collections import defaultdict


def main(path, tokens):
    rows = list(csv.DictReader(open(path)))
    d = defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        m = re.search(r"gemv_q_kernel<(\d+), (\d+), (true|false)", n)
        if m:
            name = f"gemv<{m.group(1)},{m.group(2)},{'pair' if m.group(3) == 'true' else 'single'}>"
        else:
            m2 = re.search(r"(gemm_q_f16_kernel<\d+>|attn_prefill_kernel<\d+>|gemv_q_cols_kernel<\d+, \d+>)", n)
            name = m2.group(1) if m2 else re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:40]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("LDS_Block_Size", ""))
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in d.values())
    print(f"{'kernel':42s} {'grid':>8s} {'lds':>7s} {'n':>6s} {'avg us':>9s} {'min us':>9s} {'ms/token':>9s}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < 0.002 * tot:
            continue
        print(f"{k[0]:42s} {k[1]:>8s} {k[2]:>7s} {len(v):6d} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {sum(v)/1e6/tokens:9.3f}")
    print(f"total kernel time per token: {tot/1e6/tokens:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
