#!/bin/bash
# per-kernel table of a few T-token steps of an 8-layer 70B-shaped window (rocprofv3 --kernel-trace --stats over tools/batch_step_profile.py)
# usage (GPU box, repo root): bash tools/batch_step_summary.sh "2 8" [ENV=VALUE ...]
export TMPDIR=/tmp
R=$PWD
TS=${1:-"2 8"}; shift
cd /tmp
for T in $TS; do
  rm -rf /tmp/bs$T
  env "$@" rocprofv3 --kernel-trace --stats -d /tmp/bs$T --output-format csv -- python $R/tools/batch_step_profile.py $T 6 > /dev/null 2>&1
  f=$(find /tmp/bs$T -name "*kernel_stats.csv" | head -1)
  echo "== T=$T $*"
  python $R/tools/prof_table.py "$f" 7 8
done
