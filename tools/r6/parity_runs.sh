#!/bin/bash
# round 6 parity records (run ON THE GPU BOX from the repo root): 8200-token prompt at the Llama-3-70B head shape (64 / 8 / 128; two layers, narrow ffn - the
# CPU reference must get through the prompt), F16 cache, plug-in and engine. Output -> gpurun_out/r6_parity_long.log (copied into profiles/r06_parity_long_context.txt)
cd "$(dirname "$0")/../.."
PM355_8D_LONG=70bh timeout 600 python -u -m pytest tests/test_gpu_parity_8d.py -x -q -s -k "8k_prompt" > gpurun_out/r6_parity_long.log 2>&1
grep -v "^$" gpurun_out/r6_parity_long.log | tail -25
