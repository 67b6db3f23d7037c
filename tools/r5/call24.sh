#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=$R/gpurun_out/c24; mkdir -p $O
cd /tmp
for v in r04 cur; do
  unset PM355_LIB
  [ $v = r04 ] && export PM355_LIB=$R/ab/r04lib/libprima_mi355.so
  rm -rf /tmp/prof_$v && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-extras --prefill 0 > $O/bench_$v.json 2>/dev/null
  f=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
  python $R/tools/prof_summary.py $f 53 > $O/summary_$v.txt 2>&1
  echo "== $v"; head -30 $O/summary_$v.txt
done
