mkdir -p gpurun_out/r6z
timeout 1500 python -u -m pytest tests -q -x -m gpu > gpurun_out/r6z/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/r6z/gputest.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6z/bench.json 2> gpurun_out/r6z/bench.err
tail -5 gpurun_out/r6z/gputest.log
