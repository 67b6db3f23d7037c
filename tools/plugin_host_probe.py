"""Where a plug-in token's wall time goes: llama_decode + the MI355 plug-in with GGML_MI355_STATS=1.

Prints, per shape, the wall time per token, the host time inside graph_compute split by phase (ordering behind uploads, fingerprint,
plan lookup, KV-cell patch, hipGraph launch), the time inside synchronize (= waiting for the device) and the remainder, which is
libllama's own host work per token (llama_build_graph, ggml_backend_sched_alloc_graph, input upload, sampling in the driver).
    python tools/plugin_host_probe.py [8b] [70b8] [70b16]
"""
import os, re, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from prima_cpp_amd import gguf as G
from _bind import run_llama_driver

SHAPES = {
    '8b':    dict(n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256),
    '70b8':  dict(n_layer=8,  n_embd=8192, n_head=64, n_head_kv=8, n_ff=28672, n_vocab=128256),
    '70b16': dict(n_layer=16, n_embd=8192, n_head=64, n_head_kv=8, n_ff=28672, n_vocab=128256),
}
N_GEN = 128

def one(name):
    p = f'/tmp/probe_{name}.gguf'
    G.write_synthetic_model(p, arch=0, **SHAPES[name])
    prompt = np.random.default_rng(1).integers(0, 128256, 16)
    t, l, st = run_llama_driver(p, prompt, N_GEN, ngl=99, n_ctx=4096, threads=16, extra_args=['--keep-out-in-cuda'],
                                env={'GGML_MI355_STATS': '1'}, timeout=900)
    os.remove(p)
    err = st['stderr']
    print(f"== {name}: decode {st['decode_tok_s']:.1f} tok/s, {st['decode_ms_avg'] * 1e3:.1f} us per token")
    for ln in err.splitlines():
        if "ggml-mi355" in ln and any(k in ln for k in ("stats", "host time", "phases", "steady state")): print('  ' + ln.strip())
    m = re.search(r'graph_compute ([\d.]+) ms total \(([\d.]+) us per call\).*synchronize ([\d.]+) ms in (\d+) calls', err)
    n = re.search(r'stats: graph_compute (\d+)', err)
    if m and n:
        calls, per_call, sync_ms, n_sync = int(n.group(1)), float(m.group(2)), float(m.group(3)), int(m.group(4))
        tokens = N_GEN + 1                                   # one prompt batch + N_GEN single-token graphs (all counted in the totals)
        wall = st['decode_ms_avg'] * 1e3
        gc = per_call * calls / tokens
        sy = sync_ms * 1e3 / tokens
        print(f"  per token: wall {wall:.1f} us = graph_compute {gc:.1f} ({calls / tokens:.2f} calls) + synchronize (device wait) {sy:.1f} "
              f"+ libllama / driver host work {wall - gc - sy:.1f}")

for name in (sys.argv[1:] or ['8b', '70b8', '70b16']): one(name)
