#!/bin/bash
# where an 8-token small-batch launch spends its time: ablation build of mmq_i8.hip (python -c "from prima_cpp_amd import build as b; b.build(tag='abl', extra=['-DPM_MMQ_ABLATE'])")
# usage (GPU box, repo root): bash tools/mmq_i8_ablation.sh ffn_down [T]
# PM355_MMQ_ABL bits: 1 no activation loads in the loop | 2 no weight loads in the loop | 4 no MFMA / VALU work
S=${1:-ffn_down}; T=${2:-8}
for a in 0 1 2 3 4 5 6 7; do
  echo -n "ABL=$a  "
  PM355_LIB=$PWD/ab/abl.so PM355_MMQ_ABL=$a PROBE_T=$T PROBE_SMALL_ONLY=1 python tools/small_batch_probe.py $S 2>&1 | tail -1
done
