#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6q
timeout 900 python -u -m pytest tests/test_gpu_shapes.py tests/test_gpu_ops.py -q -x -m gpu -k "shaped or small_batch" > gpurun_out/r6q/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6q/tests.log
{
echo "== Qwen2.5-72B Q6_K, new"; PROBE_MODEL=qwen timeout 300 python tools/r5/small_cross.py 3,8,16,32,48 2>&1 | grep "^T"
echo "== Qwen2.5-72B Q6_K, before (ab/abl.so)"; PM355_LIB=$PWD/ab/abl.so PROBE_MODEL=qwen timeout 300 python tools/r5/small_cross.py 3,8,16,32,48 2>&1 | grep "^T"
} > gpurun_out/r6q/cross_qwen.log 2>&1
tail -4 gpurun_out/r6q/tests.log; cat gpurun_out/r6q/cross_qwen.log
