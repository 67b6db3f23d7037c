#!/bin/bash
# the round's closing evidence on the final code: smoke(), tools/collect_profiles.sh r06 (decode kernel table + timeline, PMC traffic of the dominant decode kernel,
# prompt GEMM rates and counters) - copy gpurun_out/r06/* into profiles/ afterwards; tools/r6/suite_and_bench.sh is the other half (the -m gpu suite + the bench line)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6u
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6u/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/r6u/smoke.log
timeout 1500 bash tools/collect_profiles.sh r06 > gpurun_out/r6u/collect.log 2>&1
tail -3 gpurun_out/r6u/smoke.log; tail -5 gpurun_out/r6u/collect.log
