for S in 2 3 4; do echo "== forced split $S"; PM355_GEMM_PF_SPLITK=$S python -m pytest tests/test_gpu_ops.py -x -q -k "mfma_prefill or prompt_gemm" 2>&1 | tail -2; done
echo "== auto"; python -m pytest tests/test_gpu_ops.py -x -q -k "mfma_prefill or prompt_gemm" 2>&1 | tail -2
for sh in wo down down4 gate wk; do for S in 1 0; do echo -n "T=512 $sh split=$S (0 = auto): "; PM355_GEMM_PF_SPLITK=$S python tools/gemm_probe.py 512 $sh 2>&1 | tail -1; done; done
for sh in wo wk; do for S in 1 0; do echo -n "T=2048 $sh split=$S: "; PM355_GEMM_PF_SPLITK=$S python tools/gemm_probe.py 2048 $sh 2>&1 | tail -1; done; done
