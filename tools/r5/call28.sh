#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=$R/gpurun_out/c28; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -k "sum_of_squares or attention_in_the_qkv or attention_tail or engine or qkv or fused" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -4 $O/tests.log
for i in 1 2 3; do
for v in r04 cur cur_ss; do
  unset PM355_LIB PM355_SS
  [ $v = r04 ] && export PM355_LIB=$R/ab/r04lib/libprima_mi355.so
  [ $v = cur_ss ] && export PM355_SS=1
  ( timeout 300 python bench.py --no-extras --no-cpu-baseline --prefill 0 --steps 64 --warmup 8 > $O/bench_${v}_$i.log 2>&1 ); echo "$v $i $(grep -o '"value": [0-9.]*' $O/bench_${v}_$i.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$i.log | head -1)"
done
done
