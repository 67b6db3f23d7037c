#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for v in r04 cur; do
  unset PM355_LIB; [ $v != cur ] && export PM355_LIB=$PWD/ab/${v}lib/libprima_mi355.so
  timeout 300 python tools/r5/small_cross.py 2,3,4,8,16 2>&1 | grep "^T " | sed "s/^/$v /"
done
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "col or batch" 2>&1 | tail -2
