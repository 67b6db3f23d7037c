#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=$R/gpurun_out/c34; mkdir -p $O
cd /tmp
for v in r03 cur; do
  unset PM355_LIB; [ $v != cur ] && export PM355_LIB=$R/ab/${v}lib/libprima_mi355.so
  rm -rf /tmp/prof_$v && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/tools/r5/decode_time.py 48 > $O/dt_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
  python $R/tools/prof_summary.py $f 168 > $O/summary_$v.txt 2>&1; echo "== $v"; head -9 $O/summary_$v.txt
done
