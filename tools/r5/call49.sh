#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c49; mkdir -p $O
for st in 100 200 400 600; do for sm in 640 320; do
  PM355_ATTN_SPLIT_MIN=$sm timeout 200 python tools/r5/decode_time.py 32 $st 2>&1 | grep -E "DECODE_TIME|Error" | tail -1 | sed "s/^/start=$st split_min=$sm /"
done; done
( timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_shapes.py tests/test_gpu_llama_decode.py -x -q -k "not backend_ops" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -3 $O/tests.log
