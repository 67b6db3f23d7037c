#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c10; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "engine" > $O/test_ops_engine.log 2>&1; echo "rc=$?" >> $O/test_ops_engine.log )
grep -E "passed|failed|FAILED|AssertionError|Error|rc=" $O/test_ops_engine.log | head -20
( timeout 300 python -m pytest tests/test_gpu_engine.py -x -q -s -k "persistent_decode_engine" > $O/test_engine.log 2>&1; echo "rc=$?" >> $O/test_engine.log ); grep -E "engine vs|passed|failed|rc=" $O/test_engine.log
( timeout 300 python tools/engine_check.py --layers 1 --tokens 8 --time-steps 16 > $O/check_1l.log 2>&1 ); tail -3 $O/check_1l.log
( timeout 300 python tools/engine_check.py --model llama3-8b --layers 0 --tokens 8 --time-steps 48 > $O/check_8b.log 2>&1 ); tail -3 $O/check_8b.log
( timeout 400 python tools/engine_check.py --layers 0 --tokens 8 --time-steps 48 > $O/check_80l.log 2>&1 ); tail -3 $O/check_80l.log
