#!/bin/bash
# prompt-GEMM A/B on one box: second generation (mmq.hip kernel2) vs third (mmq_pf.hip), the shapes of a Llama-3-70B / Qwen2.5-72B layer
# usage: tools/gemm_ab.sh [T ...]
cd "$(dirname "$0")/.."
for T in ${@:-2048 512}; do
  for shape in gate gate6 down down4 wo wk; do
    for k in 2 3; do
      echo -n "kernel $k: "; PM355_GEMM_KERNEL=$k python tools/gemm_probe.py $T $shape 2>&1 | tail -1
    done
  done
done
