#!/bin/bash
# round 6, second pass over the runtime's knobs around the dependent-kernel boundary of the decode graph (names from `strings libamdhip64.so`)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for e in "BASE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "ROC_SYSTEM_SCOPE_SIGNAL=0" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "ROC_USE_FGS_KERNARG=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "AMD_DIRECT_DISPATCH=0"; do
    echo -n "$e: "; env $e timeout 120 python tools/r5/decode_time.py 64 2>&1 | grep DECODE_TIME || echo failed
  done
done
