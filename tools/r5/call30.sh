#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=$R/gpurun_out/c30; mkdir -p $O
cd $R
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from prima_cpp_amd import gguf as G
G.write_synthetic_model("/tmp/ab_8b.gguf", arch=0, n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256)
PY
cd /tmp
for v in r03 r04; do
  export LD_LIBRARY_PATH=$R/ab/${v}lib
  export REFDRV_PROMPT=128000,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19 REFDRV_NGEN=64 REFDRV_OUT=/tmp/out_$v.bin GGML_MI355_STATS=1
  rm -rf /tmp/prof_$v && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- $R/oracle/_ref/llama-ref-driver-avx2 -m /tmp/ab_8b.gguf -c 4096 -t 16 -ngl 99 --keep-out-in-cuda > $O/drv_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
  python $R/tools/prof_summary.py $f 64 > $O/summary_$v.txt 2>&1; echo "== $v"; head -12 $O/summary_$v.txt; grep -E "steady state|graph_compute phases" $O/drv_$v.log | cut -c1-400
done
