#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace) of one small-batch mat-mul call: prologue launch vs main kernel.
# usage (on the GPU box, repo root): bash tools/small_batch_trace.sh [shape] [T]
export TMPDIR=/tmp
R=$PWD; SHAPE=${1:-ffn_gate}; T=${2:-8}
cd /tmp && rm -rf /tmp/st && PROBE_T=$T PROBE_SMALL_ONLY=1 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python $R/tools/small_batch_probe.py $SHAPE > /tmp/st.log 2>&1
python - <<PY
import csv, glob, collections
k = glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(k)): acc[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in acc.items(): print(f"  {n}: {len(v)} launches, avg {sum(v) / len(v):.2f} us, min {min(v):.2f}")
PY
