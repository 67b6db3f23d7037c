#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 200 python tools/r5/attn_short_probe.py 2>&1 | grep "n_kv" 
