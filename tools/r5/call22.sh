#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c22; mkdir -p $O
for i in 1 2; do
for v in r03 r04 cur cur_noss; do
  unset PM355_LIB PM355_SS
  case $v in r03|r04) export PM355_LIB=$PWD/ab/${v}lib/libprima_mi355.so;; cur_noss) export PM355_SS=0;; esac
  ( timeout 300 python bench.py --no-extras --no-cpu-baseline --prefill 0 --steps 64 --warmup 8 > $O/bench_${v}_$i.log 2>&1 ); echo "$v $i $(grep -o '"value": [0-9.]*' $O/bench_${v}_$i.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$i.log | head -1) $(tail -1 $O/bench_${v}_$i.log | cut -c1-80)"
done
done
