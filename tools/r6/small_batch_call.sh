#!/bin/bash
# round 6: where T-token steps (2..128) stand on the integer small-batch path and on the prompt GEMM with 64- / 128-token tiles
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6v
timeout 600 python -u -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "prefill_gemm or prompt_gemm" > gpurun_out/r6v/pf_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6v/pf_tests.log
{
echo "== integer path (default)"; timeout 300 python tools/r5/small_cross.py 2,4,8,16,24,32,40,48,64 2>&1 | grep "^T"
echo "== prompt GEMM beyond 32 tokens"; PM355_MMQ_MAX_TOKENS=32 timeout 300 python tools/r5/small_cross.py 33,40,48,64,96,128 2>&1 | grep "^T"
echo "== prompt GEMM beyond 16 tokens"; PM355_MMQ_MAX_TOKENS=16 timeout 300 python tools/r5/small_cross.py 17,24,32 2>&1 | grep "^T"
echo "== prompt GEMM beyond 32 tokens, 128-token tiles"; PM355_GEMM_PF_NT=4 PM355_MMQ_MAX_TOKENS=32 timeout 300 python tools/r5/small_cross.py 33,64 2>&1 | grep "^T"
} > gpurun_out/r6v/cross.log 2>&1
timeout 600 bash tools/batch_step_summary.sh "8 32" > gpurun_out/r6v/tables.log 2>&1
PM355_MMQ_MAX_TOKENS=32 timeout 300 bash tools/batch_step_summary.sh "64" PM355_MMQ_MAX_TOKENS=32 > gpurun_out/r6v/tables64.log 2>&1
tail -3 gpurun_out/r6v/pf_tests.log; cat gpurun_out/r6v/cross.log
