#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_shapes.py -m gpu -q -x 2>&1 | tail -15) > gpurun_out/r3_c4_pytest.log 2>&1
tail -5 gpurun_out/r3_c4_pytest.log
rm -f gpurun_out/r3_c4_ab.txt
for rep in 1 2; do for pre in 1 0; do
  PM355_PRE4=$pre timeout 300 python bench.py --no-extras --no-cpu-baseline --prefill 0 --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('70b pre4=$pre', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])" >> gpurun_out/r3_c4_ab.txt 2>&1
done; done
for pre in 1 0; do
  PM355_PRE4=$pre timeout 300 python bench.py --model llama3-8b --no-extras --no-cpu-baseline --prefill 0 --steps 100 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8b pre4=$pre', d['value'], d['ms_per_step'])" >> gpurun_out/r3_c4_ab.txt 2>&1
  PM355_PRE4=$pre timeout 300 python bench.py --model qwen2.5-72b --no-extras --no-cpu-baseline --prefill 0 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qwen pre4=$pre', d['value'], d['ms_per_step'])" >> gpurun_out/r3_c4_ab.txt 2>&1
done
cat gpurun_out/r3_c4_ab.txt
PM355_LIB=ab/ts.so timeout 300 python tools/seam_anatomy.py > gpurun_out/r3_c4_anatomy.txt 2>&1
tail -12 gpurun_out/r3_c4_anatomy.txt
