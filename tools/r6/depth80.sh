#!/bin/bash
# the depth test at all 80 layers of the 70B shape (42.5 GB GGUF through the reference's quantizer): tokens, logits and the per-layer l_out distances
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6d80
PM355_8D_DEPTH=80 timeout 1250 python -u -m pytest tests/test_gpu_parity_8d.py -q -x -m gpu -s -k "at_depth" > gpurun_out/r6d80/depth80.log 2>&1; echo "rc=$?" >> gpurun_out/r6d80/depth80.log
grep "^\[8d\|plug-in :\|ref AVX2:\|passed\|failed\|rc=" gpurun_out/r6d80/depth80.log | cut -c1-1200
