#!/bin/bash
# round 6: the prompt GEMM's tail-only K split (PM355_GEMM_PF_MIXED=0: the uniform split of the whole launch), interleaved on one box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6mx
timeout 900 python -u -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "prefill_gemm or prompt_gemm" > gpurun_out/r6mx/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6mx/tests.log
{
for rep in 1 2; do
  for s in qkv gate gate6 gateq; do
    echo -n "tail split: "; timeout 120 python tools/gemm_probe.py 2048 $s 2>/dev/null
    echo -n "uniform:    "; PM355_GEMM_PF_MIXED=0 timeout 120 python tools/gemm_probe.py 2048 $s 2>/dev/null
  done
done
} > gpurun_out/r6mx/ab.log 2>&1
tail -3 gpurun_out/r6mx/tests.log; cat gpurun_out/r6mx/ab.log
