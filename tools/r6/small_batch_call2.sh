#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6w
{
echo "== prompt GEMM beyond 32 tokens, two workgroups per CU (default) / one"; PM355_MMQ_MAX_TOKENS=32 timeout 300 python tools/r5/small_cross.py 33,48,64 2>&1 | grep "^T"
PM355_GEMM_PF_CORESIDENT=0 PM355_MMQ_MAX_TOKENS=32 timeout 300 python tools/r5/small_cross.py 33,48,64 2>&1 | grep "^T"
echo "== prompt GEMM beyond 16 tokens"; PM355_MMQ_MAX_TOKENS=16 timeout 300 python tools/r5/small_cross.py 17,24,32 2>&1 | grep "^T"
echo "== integer path, attention per (head, token) / matrix-core kernel from 8 tokens"; timeout 300 python tools/r5/small_cross.py 8,12,16,24,32 2>&1 | grep "^T"
PM355_SMALL_ATTN_MFMA_MIN=8 timeout 300 python tools/r5/small_cross.py 8,12,16,24,32 2>&1 | grep "^T"
} > gpurun_out/r6w/cross.log 2>&1
PM355_MMQ_MAX_TOKENS=32 timeout 300 bash tools/batch_step_summary.sh "64" PM355_MMQ_MAX_TOKENS=32 > gpurun_out/r6w/tables64.log 2>&1
timeout 300 bash tools/batch_step_summary.sh "16 32" PM355_SMALL_ATTN_MFMA_MIN=8 > gpurun_out/r6w/tables_attn.log 2>&1
cat gpurun_out/r6w/cross.log; grep -h "attn\|gemm_pf\|kernel time" gpurun_out/r6w/tables64.log gpurun_out/r6w/tables_attn.log | cut -c1-150
