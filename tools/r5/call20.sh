#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c20; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > $O/full.log 2>&1; echo "rc=$?" >> $O/full.log )
tail -30 $O/full.log
