mkdir -p gpurun_out/r6q
PROBE_MODEL=qwen timeout 300 bash tools/batch_step_summary.sh "8 32" PROBE_MODEL=qwen > gpurun_out/r6q/tables_qwen.log 2>&1
grep -v "fill_random\|fix_scales\|set_i32\|rocclr\|at::native" gpurun_out/r6q/tables_qwen.log | cut -c1-180
