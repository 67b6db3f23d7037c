#!/bin/bash
# ablation sweep of gemm_pf_kernel (library built with -DPM_GEMM_ABLATE=1: python -m prima_cpp_amd.build pfab -DPM_GEMM_ABLATE=1); kernel durations from
# rocprofv3 --kernel-trace (the probe's own figure includes the f32 -> f16 conversion launch). Run ON THE GPU BOX from the repo root.
export TMPDIR=/tmp
R=$PWD
S=${1:-wo}; T=${2:-2048}
cd /tmp
PM355_LIB=$R/ab/pfab.so PM355_GEMM_EXP=128 PMC_ITERS=2 python $R/tools/gemm_probe.py $T $S 2>&1 | grep "pf trace" | tail -4
for e in 0 3 4 8 16 24 27; do
  rm -rf /tmp/pa_$e && PM355_LIB=$R/ab/pfab.so PM355_GEMM_EXP=$e PMC_ITERS=6 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pa_$e -- python $R/tools/gemm_probe.py $T $S > /dev/null 2>&1
  python - <<PY
import csv, glob
try:
    k = glob.glob("/tmp/pa_$e/**/*kernel_trace.csv", recursive=True)[0]
    t = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(k)) if "gemm_pf" in r["Kernel_Name"])
    print(f"  $S T=$T exp $e: median {t[len(t) // 2]:.1f} us, min {t[0]:.1f} us over {len(t)} launches")
except Exception as ex:
    print("  exp $e: failed", ex)
PY
done
