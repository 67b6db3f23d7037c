#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/c35; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -k "qkv or rope or engine or neox or epilogue or attention_tail" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -3 $O/tests.log
for i in 1 2; do for v in r03 cur; do
  unset PM355_LIB; [ $v != cur ] && export PM355_LIB=$PWD/ab/${v}lib/libprima_mi355.so
  timeout 200 python tools/r5/decode_time.py 64 2>&1 | grep -E "DECODE_TIME|Error" | tail -1
done; done
cd /tmp; unset PM355_LIB
rm -rf /tmp/prof_c && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python $R/tools/r5/decode_time.py 48 > $R/$O/dt_cur.log 2>&1
f=$(find /tmp/prof_c -name "*kernel_trace.csv" | head -1)
python $R/tools/prof_summary.py $f 168 2>&1 | head -8
