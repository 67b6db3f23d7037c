#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c38; mkdir -p $O
for i in 1 2 3; do for v in 0 1; do
  PM355_HOT_SPEC=$v timeout 200 python tools/r5/decode_time.py 64 2>&1 | grep -E "DECODE_TIME|Error" | tail -1 | sed "s/^/hot=$v /"
done; done
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_shapes.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -4 $O/tests.log
