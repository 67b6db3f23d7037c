#!/bin/bash
# rocprofv3 --pmc passes over the small-batch mat-mul (mmq_i8.hip) on the ffn_gate shape, 8 tokens: where the wave cycles go.
# Run ON THE GPU BOX from the repo root: bash tools/small_batch_pmc.sh [shape] > gpurun_out/small_batch_pmc.txt
export TMPDIR=/tmp
R=$PWD
SHAPE=${1:-ffn_gate}
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i + 1))
  rm -rf /tmp/ps_$i && PROBE_T=8 PROBE_SMALL_ONLY=1 timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/ps_$i -- python $R/tools/small_batch_probe.py $SHAPE > /tmp/ps_$i.log 2>&1
  python - <<PY
import csv, glob, collections
try:
    f = glob.glob("/tmp/ps_$i/**/*counter_collection.csv", recursive=True)[0]; k = glob.glob("/tmp/ps_$i/**/*kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mmq_i8_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(k)) if "mmq_i8_kernel" in r["Kernel_Name"]]
    for c, v in acc.items(): print(f"  {c}: avg {sum(v) / len(v):.6g} over {len(v)} launches; kernel duration {sum(t) / len(t):.1f} us")
    if not acc: print("  pass $i ($set): no rows;", open("/tmp/ps_$i.log").read()[-400:])
except Exception as e:
    print("  pass $i ($set): failed", e, open("/tmp/ps_$i.log").read()[-400:])
PY
done
