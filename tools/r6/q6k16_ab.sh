#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6s3
timeout 900 python -u -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "small_batch" > gpurun_out/r6s3/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6s3/tests.log
{
for rep in 1 2; do
echo "== new (Q6_K 9..16 tokens in the one-job multi-job form)"; timeout 300 python tools/r5/small_cross.py 12,16 2>&1 | grep "^T"; PROBE_MODEL=qwen timeout 300 python tools/r5/small_cross.py 12,16 2>&1 | grep "^T"
echo "== ab/nosplit.so (before)"; PM355_LIB=$PWD/ab/nosplit.so timeout 300 python tools/r5/small_cross.py 12,16 2>&1 | grep "^T"; PM355_LIB=$PWD/ab/nosplit.so PROBE_MODEL=qwen timeout 300 python tools/r5/small_cross.py 12,16 2>&1 | grep "^T"
done
} > gpurun_out/r6s3/cross.log 2>&1
tail -3 gpurun_out/r6s3/tests.log; cat gpurun_out/r6s3/cross.log
