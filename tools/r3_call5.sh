#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_shapes.py -m gpu -q -x 2>&1 | tail -15) > gpurun_out/r3_c5_pytest.log 2>&1
tail -5 gpurun_out/r3_c5_pytest.log
rm -f gpurun_out/r3_c5_ab.txt
b() { timeout 300 python bench.py "$@" --no-extras --no-cpu-baseline --prefill 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo "70b late  $(b --steps 40 --warmup 8)" >> gpurun_out/r3_c5_ab.txt
  echo "70b pre2  $(PM355_PRE4=0 b --steps 40 --warmup 8)" >> gpurun_out/r3_c5_ab.txt
  echo "70b early $(PM355_LIB=ab/early.so b --steps 40 --warmup 8)" >> gpurun_out/r3_c5_ab.txt
done
echo "8b late  $(b --model llama3-8b --steps 100 --warmup 8)" >> gpurun_out/r3_c5_ab.txt
echo "8b pre2  $(PM355_PRE4=0 b --model llama3-8b --steps 100 --warmup 8)" >> gpurun_out/r3_c5_ab.txt
cat gpurun_out/r3_c5_ab.txt
PM355_LIB=ab/ts.so timeout 300 python tools/seam_anatomy.py > gpurun_out/r3_c5_anatomy.txt 2>&1
tail -12 gpurun_out/r3_c5_anatomy.txt
