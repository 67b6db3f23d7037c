#!/bin/bash
# interleaved A/B of prebuilt libraries under ab/*.so on the headline decode step only (same box): tools/ab_decode.sh [rounds] [model]
R=${1:-3}; M=${2:-llama3-70b}
for i in $(seq $R); do for f in ab/*.so; do echo -n "$f  "; PM355_LIB=$PWD/$f python bench.py --model $M --steps 96 --warmup 8 --no-cpu-baseline --no-extras --prefill 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline', {}).get('avg_launch_us'))"; done; done
