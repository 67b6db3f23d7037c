#!/bin/bash
# One rocprofv3 --pmc pass per counter over the prefill GEMM probe (ffn_gate shape, T = 2048): memory-path counters of gemm_q_f16_kernel2.
# Run ON THE GPU BOX from the repo root: bash tools/gemm_pmc.sh > gpurun_out/gemm_pmc.txt
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM MemUnitStalled; do
  rm -rf /tmp/pg_$c && PM355_GEMM_KERNEL=2 PMC_ITERS=3 timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pg_$c -- python $R/tools/gemm_probe.py 2048 gate > /dev/null 2>&1
  python - <<PY
import csv, glob
try:
    f = glob.glob("/tmp/pg_$c/**/*counter_collection.csv", recursive=True)[0]; k = glob.glob("/tmp/pg_$c/**/*kernel_trace.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_q_f16" in r["Kernel_Name"]]
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(k)) if "gemm_q_f16" in r["Kernel_Name"]]
    print(f"  $c: avg {sum(v) / len(v):.6g} over {len(v)} launches; kernel duration {sum(t) / len(t):.1f} us")
except Exception as e:
    print("  $c: failed", e)
PY
done
