#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c5; mkdir -p $O
PM355_LIB=$PWD/ab/engdbg.so timeout 120 python tools/r5/dbg1.py > $O/dbg1.log 2>&1; echo "rc=$?" >> $O/dbg1.log; cat $O/dbg1.log
