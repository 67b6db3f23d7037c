#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c8; mkdir -p $O
for spec in "llama3-70b 2" "llama3-8b 2" "llama3-8b 3" "llama3-70b 3"; do
  set -- $spec
  ( timeout 300 python tools/engine_check.py --model $1 --layers $2 --tokens 6 --time-steps 0 > $O/check_$1_$2.log 2>&1 ); grep -E "per-token|engine vs" $O/check_$1_$2.log
done
( PM355_SS=1 timeout 300 python -m pytest tests/test_gpu_engine.py -x -q -s -k "persistent_decode_engine" > $O/test_engine.log 2>&1; echo "rc=$?" >> $O/test_engine.log ); grep -E "engine vs|passed|failed|rc=" $O/test_engine.log
