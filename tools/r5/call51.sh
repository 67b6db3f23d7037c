#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c51; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_llama_decode.py -x -q -k "plugin_vs_cpu or flash_attn or graph_reuse or quantized_kv" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -3 $O/tests.log
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export GGML_MI355_NO_SMALL_STAGE=1; else unset GGML_MI355_NO_SMALL_STAGE; fi
  ( timeout 300 python tools/r5/plugin_ab.py cur 1 2>&1 | grep "^round" | sed "s/^/no_small_stage=$v /" )
done
