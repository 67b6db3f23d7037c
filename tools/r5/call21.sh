#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c21; mkdir -p $O
( timeout 600 python tools/r5/plugin_ab.py r03,r04,cur 2 2>&1 | grep -v amdgpu.ids | tee $O/plugin_ab.log | tail -8 )
for cfg in "64 0" "32 0" "32 1"; do set -- $cfg
  ( PM355_MMQ_MAX_TOKENS=$1 PM355_PROMPT_I8=$2 timeout 300 python tools/r5/small_cross.py 33,40,48,64 2>&1 | grep "^T " | tee -a $O/small_cross.log )
done
