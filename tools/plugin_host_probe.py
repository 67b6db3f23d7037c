import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from prima_cpp_amd import gguf as G
from _bind import run_llama_driver
p = '/tmp/l8b.gguf'
G.write_synthetic_model(p, arch=0, n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256)
prompt = np.random.default_rng(1).integers(0, 128256, 16)
t, l, st = run_llama_driver(p, prompt, 128, ngl=99, n_ctx=4096, threads=16, extra_args=['--keep-out-in-cuda'], env={'GGML_MI355_STATS': '1'}, timeout=900)
print('decode tok/s', st['decode_tok_s'], 'ms', st['decode_ms_avg'])
print('\n'.join(ln for ln in st['stderr'].splitlines() if 'ggml-mi355' in ln and ('stats' in ln or 'host' in ln)))
