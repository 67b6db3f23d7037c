"""Short prompts (8 / 16 / 32 / 64 tokens = one llama_decode batch each) through the reference's llama_decode + the plug-in on an 8B-shaped synthetic
GGUF: prompt tokens/s with the small-batch mat-mul (mmq_i8.hip) and with GGML_MI355_NO_MMQ_I8=1 (mat-vec per column / F16 GEMM from 16)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from prima_cpp_amd import gguf as G  # noqa: E402
from _bind import run_llama_driver  # noqa: E402

p = "/tmp/l8b.gguf"
G.write_synthetic_model(p, arch=0, n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256)
rng = np.random.default_rng(1)
for n in (8, 16, 32, 64):
    prompt = rng.integers(0, 128256, n)
    row = []
    for env in ({}, {"GGML_MI355_NO_MMQ_I8": "1"}):
        t, l, st = run_llama_driver(p, prompt, 4, ngl=99, n_ctx=512, threads=8, extra_args=["--keep-out-in-cuda"], env=env, timeout=300)
        row.append((st["prompt_tok_s"], st["decode_tok_s"], list(t)))
    same = row[0][2] == row[1][2]
    print(f"prompt {n:3d} tokens: {row[0][0]:9.1f} tok/s with mmq_i8, {row[1][0]:9.1f} without; decode {row[0][1]:.1f} / {row[1][1]:.1f} tok/s; same greedy tokens: {same}", flush=True)
