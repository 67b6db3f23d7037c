#!/bin/bash
# Collects the rocprofv3 evidence of a round into gpurun_out/$1 (copy what should be judged into profiles/). Run ON THE GPU BOX from the
# repo root:  bash tools/collect_profiles.sh r02
# kernel-trace / stats passes and PMC passes are SEPARATE runs (never --pmc together with trace domains other than --kernel-trace).
set -u
R=${1:-r06}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# (1) decode: kernel trace of the default bench workload (Llama-3-70B Q4_K_M) -> per-kernel table + one-token timeline
rm -rf /tmp/prof_dec && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $OLDPWD/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-extras --prefill 0 > $OUT/${R}_bench_under_profiler.json 2>/dev/null
f=$(find /tmp/prof_dec -name "*kernel_trace.csv" | head -1)
{ echo "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-extras --prefill 0"; echo "(Llama-3-70B Q4_K_M decode: 52 tokens; tools/prof_summary.py <trace> 53, tools/timeline.py <trace>)"; echo;
  python $OLDPWD/tools/prof_summary.py $f 53; echo; python $OLDPWD/tools/timeline.py $f 3; } > $OUT/${R}_decode_summary.txt
cp $(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1) $OUT/${R}_decode_kernel_stats.csv 2>/dev/null
# (2) HBM traffic of the dominant decode kernel: FETCH_SIZE / WRITE_SIZE, one pass each
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $OLDPWD/tools/pmc_probe.py > /tmp/pmc_$c.log 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $OUT/${R}_pmc_$c.csv
done
python - <<PY
import csv, hashlib, json
def src_hash():       # the kernel this measurement belongs to: bench.py refuses the file once these sources change
    h = hashlib.sha256()
    for f in ("mmvq.hip", "mmvq_device.h", "pm355_device.h"):
        h.update(open("$OLDPWD/prima_cpp_amd/csrc/" + f, "rb").read())
    return h.hexdigest()[:16]
def vals(path):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "gemv_q_kernel" in r["Kernel_Name"]]
f, w = vals("$OUT/${R}_pmc_FETCH_SIZE.csv"), vals("$OUT/${R}_pmc_WRITE_SIZE.csv")
alg = 2 * 28672 * 4608
fr = sorted(f)[len(f) // 2] * 1024 * 2          # KB -> bytes, gfx950 correction x2 (MI355X_MICROARCH.md, HBM section)
wr = sorted(w)[len(w) // 2] * 1024
json.dump({"source": "rocprofv3 --pmc <counter> --kernel-trace --output-format csv -- python tools/pmc_probe.py (separate passes)",
           "kernel": "gemv_q_kernel<12,12,true> Q4_K gate/up pair, K=8192, N=28672 (Llama-3-70B ffn), row-SoA layout",
           "algorithmic_bytes_per_launch": alg, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
           "hbm_read_bytes_per_launch": int(fr), "hbm_write_bytes_per_launch_uncalibrated": int(wr),
           "traffic_over_algorithmic": fr / alg, "kernel_source_sha16": src_hash()}, open("$OUT/${R}_pmc_traffic.json", "w"), indent=1)
print("traffic/algorithmic", fr / alg)
PY
# (3) prefill GEMM: achieved TFLOP/s + MFMA / VALU / LDS utilisation and wait counters, one pass per counter (SKIP_PREFILL=1: leave it out)
[ "${SKIP_PREFILL:-0}" = 1 ] || { echo "prefill GEMM (ffn_gate shape, T = 2048), tools/gemm_probe.py; counters: separate rocprofv3 --pmc <counter> --kernel-trace passes, 4 launches each";
  echo "== default kernel selection (mmq_pf.hip); gate6 = the ffn_gate shape in Q6_K (Qwen2.5-72B file type); then the second generation (PM355_GEMM_KERNEL=2) on the same box"
  for s in gate gate6 wo down wk qkv gateq; do python $OLDPWD/tools/gemm_probe.py 2048 $s 2>/dev/null; done   # (qkv: wq | wk | wv as one launch; gateq: Qwen2.5-72B's gate | up pair launch)
  for s in gate gate6 wo down; do python $OLDPWD/tools/gemm_probe.py 512 $s 2>/dev/null; done
  for s in gate gate6 wo down; do echo -n "kernel2: "; PM355_GEMM_KERNEL=2 python $OLDPWD/tools/gemm_probe.py 2048 $s 2>/dev/null; done
  python $OLDPWD/tools/torch_gemm_ref.py 2>/dev/null
  echo "== the launches of the layer as the engine makes them (bench.py prefill.roofline measures the same): see BENCH / r06_bench_builder_run.json"
  for s in gate gate6; do echo "== counters, default kernel (mmq_pf.hip gemm_pf_kernel), $s shape";
    for c in MfmaUtil VALUBusy LdsUtil GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
      rm -rf /tmp/pg_$c && PMC_ITERS=3 timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pg_$c -- python $OLDPWD/tools/gemm_probe.py 2048 $s > /dev/null 2>&1
      python - <<PY
import csv, glob
try:
    f = glob.glob("/tmp/pg_$c/**/*counter_collection.csv", recursive=True)[0]; k = glob.glob("/tmp/pg_$c/**/*kernel_trace.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_pf" in r["Kernel_Name"]]
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(k)) if "gemm_pf" in r["Kernel_Name"]]
    print(f"  $c: avg {sum(v) / len(v):.5g} over {len(v)} launches; kernel duration {sum(t) / len(t):.1f} us")
except Exception as e:
    print("  $c: failed", e)
PY
    done; done; } > $OUT/${R}_prefill_pmc.txt 2>&1
ls -la $OUT
