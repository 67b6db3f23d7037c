#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6x
timeout 900 python -u -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "small_batch" > gpurun_out/r6x/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6x/tests.log
{
echo "== new (two tiles in flight for Q4_K <= 8 tokens, 5..7 row groups in one round)"; timeout 300 python tools/r5/small_cross.py 2,3,4,8,16,32 2>&1 | grep "^T"
echo "== new, PM355_MMQ_RGB8_MIN=8"; PM355_MMQ_RGB8_MIN=8 timeout 300 python tools/r5/small_cross.py 4,8,16,32 2>&1 | grep "^T"
echo "== before (ab/abl.so: round-5 kernel)"; PM355_LIB=$PWD/ab/abl.so timeout 300 python tools/r5/small_cross.py 2,3,4,8,16,32 2>&1 | grep "^T"
echo "== new again"; timeout 300 python tools/r5/small_cross.py 4,8,32 2>&1 | grep "^T"
echo "== per launch"; PROBE_T=4,8,16,32 PROBE_SMALL_ONLY=1 timeout 300 python tools/small_batch_probe.py 2>&1 | grep small
echo "== per launch, before"; PM355_LIB=$PWD/ab/abl.so PROBE_T=4,8,16,32 PROBE_SMALL_ONLY=1 timeout 300 python tools/small_batch_probe.py 2>&1 | grep small
echo "== ablation of the round-5 kernel, 8 tokens"; for s in ffn_gate ffn_down attn_q; do echo $s; timeout 300 bash tools/mmq_i8_ablation.sh $s 8; done
} > gpurun_out/r6x/cross.log 2>&1
tail -3 gpurun_out/r6x/tests.log; cat gpurun_out/r6x/cross.log
