#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6y
timeout 900 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -q -x -m gpu -k "small or batch or prefill or prompt" > gpurun_out/r6y/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6y/tests.log
{
echo "== new (cheap look-ahead past the last step, batched scale staging, wv inside the wq | wk launch, attention on the matrix cores from 16 tokens, 33..64 on the prompt GEMM)"
timeout 300 python tools/r5/small_cross.py 2,3,4,8,16,24,32,33,48,64 2>&1 | grep "^T"
echo "== new, PM355_MMQ_DUAL=0"; PM355_MMQ_DUAL=0 timeout 300 python tools/r5/small_cross.py 4,8,16,32 2>&1 | grep "^T"
echo "== new + two tiles in flight (ab/d2.so)"; PM355_LIB=$PWD/ab/d2.so timeout 300 python tools/r5/small_cross.py 4,8 2>&1 | grep "^T"
echo "== before (ab/abl.so: round-5 kernel)"; PM355_LIB=$PWD/ab/abl.so timeout 300 python tools/r5/small_cross.py 2,4,8,16,32 2>&1 | grep "^T"
echo "== new again"; timeout 300 python tools/r5/small_cross.py 4,8,16,32 2>&1 | grep "^T"
echo "== per launch"; PROBE_T=8,32 PROBE_SMALL_ONLY=1 timeout 300 python tools/small_batch_probe.py 2>&1 | grep small
echo "== per launch, two tiles in flight"; PM355_LIB=$PWD/ab/d2.so PROBE_T=8 PROBE_SMALL_ONLY=1 timeout 300 python tools/small_batch_probe.py 2>&1 | grep small
} > gpurun_out/r6y/cross.log 2>&1
timeout 300 bash tools/batch_step_summary.sh "8 32" > gpurun_out/r6y/tables.log 2>&1
tail -5 gpurun_out/r6y/tests.log; cat gpurun_out/r6y/cross.log
