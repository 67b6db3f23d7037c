#!/bin/bash
# One rocprofv3 --pmc pass per counter over the integer prompt GEMM (mmq_big.hip, ffn_gate shape, T = 2048).
# Run ON THE GPU BOX from the repo root: bash tools/gemm_i8_pmc.sh [shape] > gpurun_out/gemm_i8_pmc.txt
export TMPDIR=/tmp
R=$PWD
SHAPE=${1:-gate}
cd /tmp
for c in MfmaUtil VALUBusy LdsUtil LdsBankConflict GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum MemUnitStalled; do
  rm -rf /tmp/pg_$c && PROBE_ONLY=i8 PMC_ITERS=3 timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pg_$c -- python $R/tools/gemm_i8_probe.py 2048 $SHAPE > /dev/null 2>&1
  python - <<PY
import csv, glob
try:
    f = glob.glob("/tmp/pg_$c/**/*counter_collection.csv", recursive=True)[0]; k = glob.glob("/tmp/pg_$c/**/*kernel_trace.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "mmq_big" in r["Kernel_Name"]]
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(k)) if "mmq_big" in r["Kernel_Name"]]
    print(f"  $c: avg {sum(v) / len(v):.6g} over {len(v)} launches; kernel duration {sum(t) / len(t):.1f} us")
except Exception as e:
    print("  $c: failed", e)
PY
done
