#!/bin/bash
# Round 6, small batches: step times of T-token batches through a 16-layer 70B- (and Qwen2.5-72B-) shaped window on the default routing, the prompt-GEMM crossover,
# and per-kernel tables of 8- / 32- / 64-token steps. A/B partners are separate builds selected with PM355_LIB (python -m prima_cpp_amd.build <tag> -D<FLAG>,
# -> ab/<tag>.so) or the A/B switches of INTEGRATION.md (PM355_MMQ_DUAL, PM355_MMQ_RGB8_MIN, PM355_SMALL_ATTN_MFMA_MIN, PM355_MMQ_MAX_TOKENS, PM355_GEMM_PF_NT);
# the numbers of the round are in profiles/r06_small_batch.txt.   usage (GPU box): bash tools/r6/small_batch_measure.sh [ab/<tag>.so]
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6sb
AB=${1:-}
{
echo "== default routing"; timeout 300 python tools/r5/small_cross.py 2,3,4,8,16,24,32,33,48,64 2>&1 | grep "^T"
[ -n "$AB" ] && { echo "== $AB"; PM355_LIB=$PWD/$AB timeout 300 python tools/r5/small_cross.py 2,3,4,8,16,24,32,33,48,64 2>&1 | grep "^T"; }
echo "== integer path up to 64 tokens (round-5 routing)"; PM355_MMQ_MAX_TOKENS=64 timeout 300 python tools/r5/small_cross.py 33,40,48,64 2>&1 | grep "^T"
echo "== prompt GEMM from 17 tokens"; PM355_MMQ_MAX_TOKENS=16 timeout 300 python tools/r5/small_cross.py 17,24,32 2>&1 | grep "^T"
echo "== Qwen2.5-72B Q6_K"; PROBE_MODEL=qwen timeout 300 python tools/r5/small_cross.py 3,8,16,32,48 2>&1 | grep "^T"
echo "== per launch"; PROBE_T=8,32 PROBE_SMALL_ONLY=1 timeout 300 python tools/small_batch_probe.py 2>&1 | grep small
} > gpurun_out/r6sb/cross.log 2>&1
timeout 600 bash tools/batch_step_summary.sh "8 32 64" > gpurun_out/r6sb/tables.log 2>&1
PROBE_MODEL=qwen timeout 300 bash tools/batch_step_summary.sh "8 32" PROBE_MODEL=qwen > gpurun_out/r6sb/tables_qwen.log 2>&1
cat gpurun_out/r6sb/cross.log
