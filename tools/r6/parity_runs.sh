#!/bin/bash
# round 6 parity records (run ON THE GPU BOX from the repo root): 8200-token prompt at the Llama-3-70B head shape (64 / 8 / 128, 8 layers), F16 cache;
# the 16-layer depth test that now runs by default. Output -> gpurun_out/r6_parity.log (copied into profiles/r06_parity_long_context.txt)
cd "$(dirname "$0")/../.."
( PM355_8D_LONG=70b8 timeout 3000 python -m pytest tests/test_gpu_parity_8d.py -x -q -s -k "8k_prompt" 2>&1 | grep -v "^$" | tail -25
  timeout 1500 python -m pytest tests/test_gpu_parity_8d.py -x -q -s -k "at_depth" 2>&1 | grep -v "^$" | tail -15 ) > gpurun_out/r6_parity.log 2>&1
