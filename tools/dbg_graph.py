import os, sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from _bind import run_llama_driver, write_gguf_from_arrays
z = np.load('tests/golden/tiny_llama_decode.npz')
path = write_gguf_from_arrays('/tmp/tiny.gguf', z)
t, l, st = run_llama_driver(path, z['prompt'], 3, ngl=99, n_ctx=64, extra_args=['--keep-out-in-cuda'], env={'GGML_MI355_DEBUG_PLAN': '1', 'GGML_MI355_DEBUG_PLAN_STEPS': '1'})
open('gpurun_out/r2_graph_llama.txt', 'w').write(st['stderr'])
z = np.load('tests/golden/tiny_qwen2_decode.npz')
path = write_gguf_from_arrays('/tmp/tinyq.gguf', z)
t, l, st = run_llama_driver(path, z['prompt'], 3, ngl=99, n_ctx=64, extra_args=['--keep-out-in-cuda'], env={'GGML_MI355_DEBUG_PLAN': '1', 'GGML_MI355_DEBUG_PLAN_STEPS': '1'})
open('gpurun_out/r2_graph_qwen2.txt', 'w').write(st['stderr'])
