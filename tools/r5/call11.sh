#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c11; mkdir -p $O
for v in engdbg engdbg255; do
( PM355_LIB=$PWD/ab/$v.so timeout 300 python tools/engine_check.py --layers 3 --tokens 3 --time-steps 0 > $O/dbg_$v.log 2>&1; echo "rc=$?" >> $O/dbg_$v.log )
grep -E "eng phase" $O/dbg_$v.log | tail -18
done
