#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6e2
timeout 1200 python -u -m pytest tests/test_gpu_experiments.py tests/test_gpu_engine.py tests/test_gpu_shapes.py -q -m gpu > gpurun_out/r6e2/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6e2/tests.log
tail -6 gpurun_out/r6e2/tests.log
