#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6s2
timeout 900 python -u -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "small_batch" > gpurun_out/r6s2/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6s2/tests.log
{
for rep in 1 2; do
echo "== scale in the operand (17..32 tokens, Q4_K)"; timeout 300 python tools/r5/small_cross.py 17,24,32 2>&1 | grep "^T"
echo "== scale on the integer tile (ab/nosplit.so)"; PM355_LIB=$PWD/ab/nosplit.so timeout 300 python tools/r5/small_cross.py 17,24,32 2>&1 | grep "^T"
done
echo "== per launch"; PROBE_T=32 PROBE_SMALL_ONLY=1 timeout 300 python tools/small_batch_probe.py 2>&1 | grep small
echo "== per launch, ab/nosplit.so"; PM355_LIB=$PWD/ab/nosplit.so PROBE_T=32 PROBE_SMALL_ONLY=1 timeout 300 python tools/small_batch_probe.py 2>&1 | grep small
} > gpurun_out/r6s2/cross.log 2>&1
tail -3 gpurun_out/r6s2/tests.log; cat gpurun_out/r6s2/cross.log
