#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c9; mkdir -p $O
( PM355_LIB=$PWD/ab/engdbg.so timeout 300 python tools/engine_check.py --layers 1 --tokens 3 --time-steps 0 > $O/dbg_1l.log 2>&1; echo "rc=$?" >> $O/dbg_1l.log )
grep -E "eng phase|eng loader|eng consumer|ENGINE_CHECK" $O/dbg_1l.log | tail -9
