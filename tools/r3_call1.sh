#!/bin/bash
# round 3, GPU call 1: regression + new parity fixtures (small shape) + seam anatomy + QKV-epilogue A/B
mkdir -p gpurun_out
export PM355_8D_SIZES=small
(time timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60) > gpurun_out/r3_c1_pytest.log 2>&1
for epi in 1 0; do
  PM355_QKV_EPI=$epi PM355_LIB=ab/ts.so timeout 300 python tools/seam_anatomy.py > gpurun_out/r3_c1_anatomy_epi$epi.txt 2>&1
done
for epi in 1 0 1 0; do
  PM355_QKV_EPI=$epi timeout 300 python bench.py --no-extras --no-cpu-baseline --prefill 0 --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('70b epi=$epi', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])" >> gpurun_out/r3_c1_ab.txt 2>&1
done
for epi in 1 0; do
  PM355_QKV_EPI=$epi timeout 300 python bench.py --model llama3-8b --no-extras --no-cpu-baseline --prefill 0 --steps 100 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8b epi=$epi', d['value'], d['ms_per_step'])" >> gpurun_out/r3_c1_ab.txt 2>&1
done
cat gpurun_out/r3_c1_ab.txt
tail -25 gpurun_out/r3_c1_anatomy_epi1.txt
tail -30 gpurun_out/r3_c1_pytest.log
