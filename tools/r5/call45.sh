#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
( timeout 900 python bench.py > $O/r05_bench_builder_run.json 2> $O/bench_stderr.log; echo "rc=$?" >> $O/bench_stderr.log )
tail -2 $O/bench_stderr.log; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/r05_bench_builder_run.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("avg_launch_us"), "traffic", d.get("roofline", {}).get("traffic"))
print("parity", d["config"].get("greedy_tokens_match_reference"), d.get("parity_check", {}).get("tokens_identical"), d.get("parity_check", {}).get("logits_nmse"))
print("cpu", d.get("cpu_baseline", {}).get("value"))
print("extra", [(e.get("workload", "")[:30], e.get("tokens_per_s")) for e in d.get("extra_configs", [])])
pf = d.get("prefill", {}); print("prefill", pf.get("tokens_per_s"), pf.get("n_ubatch_512"), pf.get("roofline", {}).get("achieved"), pf.get("small_batch"))
print("qwen prefill", d.get("prefill_qwen25_72b_q6k", {}).get("tokens_per_s"), d.get("prefill_qwen25_72b_q6k", {}).get("roofline", {}).get("achieved"))
lc = d.get("long_context", {}); print("long", lc.get("contexts"), lc.get("attention_kernel", {}).get("cells"), lc.get("attention_kernel_q8_0_kv", {}).get("cells"))
print("long qwen", d.get("long_context_qwen25_72b", {}).get("contexts"))
pd = d.get("plugin_decode", {}); print("plugin 8b", pd.get("tokens_per_s"), pd.get("graph_reuse_patch"), pd.get("flash_attn"))
p7 = d.get("plugin_decode_70b", {}); print("plugin 70b", p7.get("tokens_per_s_80_layers"), p7.get("frac_of_engine"), p7.get("graph_reuse_patch", {}).get("tokens_per_s_80_layers"), p7.get("graph_reuse_patch", {}).get("frac_of_engine"))
print("stream", d.get("weight_streaming"))
PY
bash tools/collect_profiles.sh r05 2>&1 | tail -5
ls gpurun_out/r05
