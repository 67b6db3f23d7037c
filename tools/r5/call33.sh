#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c33; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_shapes.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -4 $O/tests.log
for i in 1 2; do for v in r03 r04 cur; do
  unset PM355_LIB; [ $v != cur ] && export PM355_LIB=$PWD/ab/${v}lib/libprima_mi355.so
  timeout 200 python tools/r5/decode_time.py 64 2>&1 | grep -E "DECODE_TIME|Error" | tail -1
done; done
