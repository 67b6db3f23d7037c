#!/bin/bash
# the round's closing run: the whole -m gpu suite (no -x: every failure is listed), then the default bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6z
timeout 1800 python -u -m pytest tests -q -m gpu > gpurun_out/r6z/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/r6z/gputest.log
timeout 900 python bench.py > gpurun_out/r6z/bench.json 2> gpurun_out/r6z/bench.err
tail -8 gpurun_out/r6z/gputest.log
