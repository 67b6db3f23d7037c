#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c43; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_llama_decode.py -x -q -k "not backend_ops and not 8b_shape" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -3 $O/tests.log
( timeout 600 python tools/r5/plugin_ab.py r03,r04,cur 2 2>&1 | grep -v amdgpu.ids | tee $O/plugin_ab.log | tail -8 )
