#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c17; mkdir -p $O
timeout 60 tools/r5/unaligned_probe 2>&1 | tail -2
for parts in 0 8 16 32 64; do
  echo "== PM355_FLASH_PARTS=$parts"
  ( PM355_FLASH_PARTS=$parts timeout 200 python tools/attn_long_probe.py 64 8 128 4096,8192,16384,32768 2>&1 | grep -v amdgpu.ids | tail -5 ) | tee -a $O/spans_$parts.log
done
