#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c16; mkdir -p $O
for i in 1 2 3; do
for v in r4 cur; do
  if [ $v = cur ]; then unset PM355_LIB; else export PM355_LIB=$PWD/ab/$v.so; fi
  ( timeout 300 python bench.py --no-extras --no-cpu-baseline --prefill 0 --steps 64 --warmup 8 > $O/bench_${v}_$i.log 2>&1 ); echo "$v $i $(grep -o '"value": [0-9.]*' $O/bench_${v}_$i.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$i.log | head -1)"
done
done
