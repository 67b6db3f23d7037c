#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for i in 1 2 3; do for v in 0 1; do
  PM355_KC_SPEC=$v timeout 200 python tools/r5/decode_time.py 64 2>&1 | grep -E "DECODE_TIME|Error" | tail -1 | sed "s/^/kc=$v /"
done; done
