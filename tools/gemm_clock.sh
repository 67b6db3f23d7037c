export TMPDIR=/tmp
R=$PWD
cd /tmp
for e in 0 2 1 96; do
  for c in GRBM_GUI_ACTIVE; do
    rm -rf /tmp/pg_$c && PM355_LIB=$R/prima_cpp_amd/libprima_abl.so PM355_GEMM_KERNEL=2 PM355_GEMM_EXP=$e PMC_ITERS=3 timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pg_$c -- python $R/tools/gemm_probe.py 2048 gate > /dev/null 2>&1
    python - <<PY
import csv, glob
f = glob.glob("/tmp/pg_$c/**/*counter_collection.csv", recursive=True)[0]; k = glob.glob("/tmp/pg_$c/**/*kernel_trace.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_q_f16" in r["Kernel_Name"]]
t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(k)) if "gemm_q_f16" in r["Kernel_Name"]]
a, d = sum(v) / len(v), sum(t) / len(t)
print(f"  exp $e: $c avg {a:.6g}; kernel duration {d:.1f} us -> {a / 8 / d / 1e3:.3f} GHz")
PY
  done
done
