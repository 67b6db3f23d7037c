"""8B-shaped GGUF through llama_decode + the plug-in at a long context: prompt of N tokens, then greedy decode; default graphs vs --flash-attn."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from prima_cpp_amd import gguf as G
from _bind import run_llama_driver
p = '/tmp/l8b.gguf'
G.write_synthetic_model(p, arch=0, n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
prompt = np.random.default_rng(1).integers(0, 128256, N)
# (GGML_MI355_ATTN_MFMA=0: the round-2 flash-decoding kernel instead of the matrix-core one - the A/B of the row-major-V operand path of round 4)
for extra, env in (([], {}), (['-fa'], {}), (['-fa'], {'GGML_MI355_ATTN_MFMA': '0'})):
    t, l, st = run_llama_driver(p, prompt, 48, ngl=99, n_ctx=N + 256, threads=16, extra_args=['--keep-out-in-cuda'] + extra, env=dict({'GGML_MI355_STATS': '1'}, **env), timeout=200, chunk=2048)
    print(extra, env, 'n_kv ~', N, 'prompt tok/s %.0f' % st['prompt_tok_s'], 'decode tok/s %.1f (%.3f ms)' % (st['decode_tok_s'], st['decode_ms_avg']))
