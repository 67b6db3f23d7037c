#!/bin/bash
mkdir -p gpurun_out
export PM355_8D_SIZES=small
(time timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_8d.py tests/test_gpu_llama_decode.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -60) > gpurun_out/r3_c8_pytest.log 2>&1
tail -40 gpurun_out/r3_c8_pytest.log
