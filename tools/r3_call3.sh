#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_shapes.py -m gpu -q -s -k "qkv_epilogue or model_shaped" 2>&1 | tail -120) > gpurun_out/r3_c3_pytest.log 2>&1
tail -120 gpurun_out/r3_c3_pytest.log
