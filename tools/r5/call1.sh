#!/bin/bash
# Round 5, GPU call 1: correctness of the changed mat-vec / layout code, then interleaved A/B of
#   r4   = round-4 behaviour (separate Q6_K tail streams, nt tail loads)   [PM355_SS=0 -> the round-4 decode form]
#   q6c  = round-4 Q6_K layout, cached tail loads
#   cur  = grouped Q6_K tail + cached loads (the default build)
# each with / without the producer-side sum of squares, on the three model shapes; seam anatomy of the new form; seam-cost probe of the engine skeleton.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/c1
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q \
    -k "repack or gemv or small_batch or prompt_matmul or get_rows or fused_prologue or mfma_prefill_gemm or sum_of_squares or fused_qkv or uploader or qkv_epilogue" \
    > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log ) 
tail -3 $O/tests.log
one() {   # lib ss model steps
  local lib=$1 ss=$2 model=$3
  local env_lib=""
  [ "$lib" != "cur" ] && env_lib="$PWD/ab/$lib.so"
  PM355_LIB=$env_lib PM355_SS=$ss timeout 300 python bench.py --model $model --steps 96 --warmup 8 --no-cpu-baseline --no-extras --prefill 0 2>$O/last_err.log | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib ss=$ss $model', d['value'], d['ms_per_step'], d.get('roofline', {}).get('avg_launch_us'))"
}
{
for i in 1 2 3; do
  for cfg in "r4 0" "r4 1" "q6c 1" "cur 1" "cur 0"; do one $cfg llama3-70b; done
done
for i in 1 2; do
  for cfg in "r4 0" "cur 1"; do one $cfg llama3-8b; done
  for cfg in "r4 0" "r4 1" "q6c 1" "cur 1"; do one $cfg qwen2.5-72b; done
done
} > $O/ab.log 2>&1
cat $O/ab.log
PM355_LIB=$PWD/ab/ts.so timeout 300 python tools/seam_anatomy.py > $O/seam_anatomy.txt 2>&1
tail -25 $O/seam_anatomy.txt
timeout 300 python tools/engine_probe2.py > $O/engine_probe2.txt 2>&1
cat $O/engine_probe2.txt
