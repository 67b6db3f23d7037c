#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for v in 0 1; do
  export PM355_HOT_SPEC=$v
  rm -rf /tmp/prof_$v && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/tools/r5/decode_time.py 48 > /tmp/dt_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
  echo "== hot=$v"; python $R/tools/prof_summary.py $f 168 2>&1 | head -8
done
