#!/bin/bash
mkdir -p gpurun_out
export PM355_8D_SIZES=small
(time timeout 900 python -m pytest tests -m gpu -q -s -x --deselect tests/test_gpu_parity_8d.py 2>&1 | tail -40) > gpurun_out/r3_c2_pytest.log 2>&1
(time timeout 900 python -m pytest tests/test_gpu_parity_8d.py -q -s 2>&1 | tail -60) > gpurun_out/r3_c2_parity8d.log 2>&1
tail -30 gpurun_out/r3_c2_pytest.log; tail -50 gpurun_out/r3_c2_parity8d.log
