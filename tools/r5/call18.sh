#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c18; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "q8_0_token_step" > $O/test_q8.log 2>&1; echo "rc=$?" >> $O/test_q8.log )
grep -E "passed|failed|FAILED|Error|rc=|assert|^E " $O/test_q8.log | head -20
( timeout 900 python -m pytest tests/test_gpu_parity_8d.py -x -q -s -k "long_context_q8_0" > $O/test_q8_model.log 2>&1; echo "rc=$?" >> $O/test_q8_model.log )
grep -E "^\[8d|passed|failed|FAILED|Error|rc=|assert|^E " $O/test_q8_model.log | head -20
