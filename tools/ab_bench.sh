#!/bin/bash
# interleaved A/B of prebuilt libraries under ab/*.so (same box, same process conditions): tools/ab_bench.sh [rounds] [bench args]
R=${1:-3}; shift
for i in $(seq $R); do for f in ab/*.so; do cp $f prima_cpp_amd/libprima_mi355.so; echo -n "$f  "; python bench.py --steps 48 --warmup 4 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])"; done; done
