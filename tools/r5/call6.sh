#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c6; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "engine" > $O/test_ops_engine.log 2>&1; echo "rc=$?" >> $O/test_ops_engine.log )
grep -E "passed|failed|FAILED|AssertionError|Error|rc=" $O/test_ops_engine.log | head -40
bash tools/r5/call2.sh
