#!/bin/bash
# Round 5, GPU call 2: bring-up of the persistent decode engine. Every step under its own timeout; the engine's waits are bounded.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c2; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_engine.py -x -q -s -k "persistent_decode_engine" > $O/test_engine.log 2>&1; echo "rc=$?" >> $O/test_engine.log )
tail -15 $O/test_engine.log
( PM355_ENGINE_VERBOSE=1 timeout 300 python tools/engine_check.py --layers 2 --tokens 12 --time-steps 16 > $O/check_70b_2l.log 2>&1; echo "rc=$?" >> $O/check_70b_2l.log )
tail -6 $O/check_70b_2l.log
if grep -q "ENGINE_CHECK OK" $O/check_70b_2l.log; then
  ( PM355_ENGINE_VERBOSE=1 timeout 400 python tools/engine_check.py --layers 0 --tokens 24 --time-steps 64 > $O/check_70b_80l.log 2>&1; echo "rc=$?" >> $O/check_70b_80l.log )
  tail -6 $O/check_70b_80l.log
  ( PM355_ENGINE_VERBOSE=1 timeout 400 python tools/engine_check.py --model llama3-8b --layers 0 --tokens 24 --time-steps 64 > $O/check_8b.log 2>&1; echo "rc=$?" >> $O/check_8b.log )
  tail -4 $O/check_8b.log
  ( PM355_ENGINE_VERBOSE=1 timeout 400 python tools/engine_check.py --layers 8 --tokens 30 --start-pos 50 --time-steps 0 > $O/check_70b_pos50.log 2>&1; echo "rc=$?" >> $O/check_70b_pos50.log )
  tail -4 $O/check_70b_pos50.log
fi
