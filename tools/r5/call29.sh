#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c29; mkdir -p $O
( timeout 600 python tools/r5/plugin_ab.py r03,r04,cur 2 2>&1 | grep -v amdgpu.ids | tee $O/plugin_ab.log | tail -8 )
