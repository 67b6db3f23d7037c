#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c14; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "attention_in_the_qkv_launch_tail" > $O/test_tail.log 2>&1; echo "rc=$?" >> $O/test_tail.log )
grep -E "passed|failed|FAILED|Error|rc=|assert" $O/test_tail.log | head -20
( timeout 900 python -m pytest tests/test_gpu_engine.py -x -q > $O/test_engine.log 2>&1; echo "rc=$?" >> $O/test_engine.log ); grep -E "passed|failed|FAILED|rc=" $O/test_engine.log | head
for v in 1 0 1 0; do
( PM355_ATTN_TAIL=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --prefill 0 --steps 96 --warmup 16 > $O/bench_tail$v.log 2>&1 ); echo "tail=$v $(grep -o '"value": [0-9.]*' $O/bench_tail$v.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_tail$v.log | head -1)"
done
