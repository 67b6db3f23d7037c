mkdir -p gpurun_out/r6u
timeout 900 python -u -m pytest tests/test_gpu_llama_decode.py -q -x -m gpu > gpurun_out/r6u/llama_decode.log 2>&1; echo "rc=$?" >> gpurun_out/r6u/llama_decode.log
timeout 600 bash tools/r6/decode_env_ab.sh > gpurun_out/r6u/env_ab.log 2>&1
timeout 1500 bash tools/collect_profiles.sh r06 > gpurun_out/r6u/collect.log 2>&1
timeout 900 python bench.py > gpurun_out/r6u/bench.json 2> gpurun_out/r6u/bench.err
tail -3 gpurun_out/r6u/llama_decode.log; cat gpurun_out/r6u/env_ab.log; tail -c 600 gpurun_out/r6u/bench.json
