#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c4; mkdir -p $O
timeout 60 tools/r5/dma_probe > $O/dma_probe.log 2>&1; echo "rc=$?" >> $O/dma_probe.log; cat $O/dma_probe.log
