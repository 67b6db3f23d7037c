#!/bin/bash
mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_gpu_ring.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -40) > gpurun_out/r3_c10_ring.log 2>&1
tail -40 gpurun_out/r3_c10_ring.log
