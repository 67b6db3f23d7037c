#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for i in 1 2; do for v in r03 r04 cur; do
  unset PM355_LIB; [ $v != cur ] && export PM355_LIB=$PWD/ab/${v}lib/libprima_mi355.so
  timeout 200 python tools/r5/decode_time.py 64 2>&1 | grep -E "DECODE_TIME|Error" | tail -1
done; done
