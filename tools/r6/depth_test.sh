mkdir -p gpurun_out/r6p
timeout 1700 python -u -m pytest tests/test_gpu_parity_8d.py -q -x -m gpu -s -k "at_depth" > gpurun_out/r6p/depth.log 2>&1; echo "rc=$?" >> gpurun_out/r6p/depth.log
grep -v "^llama\|^ggml\|^llm_\|^\.\.\." gpurun_out/r6p/depth.log | tail -30 | cut -c1-400
