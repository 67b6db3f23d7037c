#!/bin/bash
# round 6: runtime knobs that could move the ~2.9 us dependent-kernel boundary of the decode step (VERDICT r5 weak #9c), interleaved on one box:
# where the kernel arguments live (HIP_FORCE_DEV_KERNARG), the graph's kernarg handling (DEBUG_HIP_GRAPH_*), signal / wait policy
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for e in "BASE=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=1" "HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0"; do
    echo -n "$e: "; env $e python tools/r5/decode_time.py 64 2>&1 | grep DECODE_TIME
  done
done
