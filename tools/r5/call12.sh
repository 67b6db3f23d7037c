#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c12; mkdir -p $O
for v in eng_s0 eng_s2304 eng_s4608; do
( PM355_LIB=$PWD/ab/$v.so timeout 300 python tools/engine_check.py --layers 0 --tokens 4 --time-steps 48 > $O/check_$v.log 2>&1 ); echo $v; tail -3 $O/check_$v.log | head -2
done
