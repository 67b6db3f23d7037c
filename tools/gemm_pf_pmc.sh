#!/bin/bash
# One rocprofv3 --pmc pass per counter over the prompt-GEMM probe (mmq_pf.hip, gemm_pf_kernel). Run ON THE GPU BOX from the repo root:
#   bash tools/gemm_pf_pmc.sh [shape=gate] [T=2048] > gpurun_out/gemm_pf_pmc.txt
export TMPDIR=/tmp
R=$PWD
S=${1:-gate}; T=${2:-2048}
cd /tmp
for c in MfmaUtil VALUBusy LdsUtil GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum; do
  rm -rf /tmp/pg_$c && PMC_ITERS=3 timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pg_$c -- python $R/tools/gemm_probe.py $T $S > /dev/null 2>&1
  python - <<PY
import csv, glob
try:
    f = glob.glob("/tmp/pg_$c/**/*counter_collection.csv", recursive=True)[0]; k = glob.glob("/tmp/pg_$c/**/*kernel_trace.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_pf" in r["Kernel_Name"]]
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(k)) if "gemm_pf" in r["Kernel_Name"]]
    print(f"  $c: avg {sum(v) / len(v):.6g} over {len(v)} launches; kernel duration {sum(t) / len(t):.1f} us")
except Exception as e:
    print("  $c: failed", e)
PY
done
