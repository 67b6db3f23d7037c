#!/bin/bash
# round 6: RoPE + KV store in the epilogue of the small-batch wq | wk | wv launch (PM355_SMALL_ROPE_EPI=0: the separate rope_kv_store launch), interleaved on one box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6re
timeout 1200 python -u -m pytest tests/test_gpu_shapes.py tests/test_gpu_engine.py tests/test_gpu_ops.py -q -x -m gpu -k "shaped or small or batch or engine" > gpurun_out/r6re/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6re/tests.log
{
for rep in 1 2; do
echo "== epilogue"; timeout 300 python tools/r5/small_cross.py 2,4,8,16,32 2>&1 | grep "^T"
echo "== separate launch (PM355_SMALL_ROPE_EPI=0)"; PM355_SMALL_ROPE_EPI=0 timeout 300 python tools/r5/small_cross.py 2,4,8,16,32 2>&1 | grep "^T"
[ -f ab/nosplit.so ] && { echo "== ab/nosplit.so (two commits back)"; PM355_LIB=$PWD/ab/nosplit.so timeout 300 python tools/r5/small_cross.py 2,4,8,16,32 2>&1 | grep "^T"; }
done
echo "== Qwen (NEOX: separate launch either way)"; PROBE_MODEL=qwen timeout 300 python tools/r5/small_cross.py 8,32 2>&1 | grep "^T"
[ -f ab/nosplit.so ] && PM355_LIB=$PWD/ab/nosplit.so PROBE_MODEL=qwen timeout 300 python tools/r5/small_cross.py 8,32 2>&1 | grep "^T"
} > gpurun_out/r6re/cross.log 2>&1
tail -4 gpurun_out/r6re/tests.log; cat gpurun_out/r6re/cross.log
