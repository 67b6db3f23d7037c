#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for i in 1 2 3; do for v in 0 1; do
  PM355_HOT_SPEC=$v timeout 200 python tools/r5/decode_time.py 64 2>&1 | grep -E "DECODE_TIME|Error" | tail -1 | sed "s/^/hot=$v /"
done; done
cd /tmp
export PM355_HOT_SPEC=1
rm -rf /tmp/prof_1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1 -- python $R/tools/r5/decode_time.py 48 > /tmp/dt_1.log 2>&1
f=$(find /tmp/prof_1 -name "*kernel_trace.csv" | head -1)
python $R/tools/prof_summary.py $f 168 2>&1 | head -8
