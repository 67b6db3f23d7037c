#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=$R/gpurun_out/c26; mkdir -p $O
cd $R
for i in 1 2; do
for v in r04 cur nosscode; do
  unset PM355_LIB PM355_SS
  [ $v = r04 ] && export PM355_LIB=$R/ab/r04lib/libprima_mi355.so
  [ $v = nosscode ] && export PM355_LIB=$R/ab/v_nosscode.so PM355_SS=0
  ( timeout 300 python bench.py --no-extras --no-cpu-baseline --prefill 0 --steps 64 --warmup 8 > $O/bench_${v}_$i.log 2>&1 ); echo "$v $i $(grep -o '"value": [0-9.]*' $O/bench_${v}_$i.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$i.log | head -1)"
done
done
cd /tmp
v=nosscode; export PM355_LIB=$R/ab/v_nosscode.so PM355_SS=0
rm -rf /tmp/prof_$v && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-extras --prefill 0 > $O/bench_prof_$v.json 2>/dev/null
f=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
python $R/tools/prof_summary.py $f 53 > $O/summary_$v.txt 2>&1; head -7 $O/summary_$v.txt
