#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 200 python tools/attn_long_probe.py 64 8 128 264,328,392,520,648,1032 2>&1 | grep "n_kv"
