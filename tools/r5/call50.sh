#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c50; mkdir -p $O
for reuse in 0 1; do
  echo "#### LLAMA_MI355_GRAPH_REUSE=$reuse" | tee -a $O/host.log
  ( LLAMA_MI355_GRAPH_REUSE=$reuse timeout 300 python tools/plugin_host_probe.py 8b 70b16 2>&1 | grep -v amdgpu.ids | grep -E "^==|steady state|per token" | cut -c1-700 | tee -a $O/host.log )
done
