#!/bin/bash
mkdir -p gpurun_out
(time timeout 1700 python -m pytest tests/test_gpu_parity_8d.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -80) > gpurun_out/r3_c9_parity8d.log 2>&1
python - > gpurun_out/r3_c9_plugin_ab.txt 2>&1 <<'PY'
import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from _bind import run_llama_driver
from prima_cpp_amd import gguf as G
path = "/tmp/l8b.gguf"
G.write_synthetic_model(path, arch=0, n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256)
prompt = np.random.default_rng(1234).integers(0, 128256, 16)
for rep in range(2):
    for epi in ("1", "0"):
        for fa in ([], ["-fa"]):
            t, l, st = run_llama_driver(path, prompt, 64, ngl=99, n_ctx=4096, threads=8, extra_args=["--keep-out-in-cuda"] + fa, env={"GGML_MI355_QKV_EPI": epi}, timeout=600)
            print(f"plugin 8B epi={epi} fa={bool(fa)}: {st['decode_tok_s']:.1f} tok/s best {1000/st['decode_ms_min']:.1f}", flush=True)
os.unlink(path)
PY
cat gpurun_out/r3_c9_plugin_ab.txt | tail -10
tail -70 gpurun_out/r3_c9_parity8d.log
