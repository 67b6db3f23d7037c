#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c23; mkdir -p $O
for i in 1 2; do
for v in r04 cur v_notail v_preload v_both; do
  unset PM355_LIB
  case $v in r04) export PM355_LIB=$PWD/ab/r04lib/libprima_mi355.so;; v_*) export PM355_LIB=$PWD/ab/$v.so;; esac
  ( timeout 300 python bench.py --no-extras --no-cpu-baseline --prefill 0 --steps 64 --warmup 8 > $O/bench_${v}_$i.log 2>&1 ); echo "$v $i $(grep -o '"value": [0-9.]*' $O/bench_${v}_$i.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$i.log | head -1)"
done
done
