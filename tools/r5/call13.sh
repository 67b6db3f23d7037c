#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c13; mkdir -p $O
df -h /tmp | tail -1; nproc; free -g | head -2
FREE=$(df --output=avail -BG /tmp | tail -1 | tr -dc 0-9)
DEPTH=80; [ "$FREE" -lt 70 ] && DEPTH=32
echo "depth $DEPTH (free ${FREE}G)"
( PM355_8D_DEPTH=$DEPTH timeout 1500 python -m pytest tests/test_gpu_parity_8d.py -x -q -s -k "at_depth" > $O/depth.log 2>&1; echo "rc=$?" >> $O/depth.log )
grep -E "^\[8d|passed|failed|rc=|Error|assert" $O/depth.log | head -30
