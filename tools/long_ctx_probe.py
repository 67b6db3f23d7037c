"""Llama-3-70B Q4_K_M decode tokens/s at long contexts (engine): the prompt is prefilled in 2048-token chunks on the MFMA path, then
single tokens are timed through the captured graph. Roofline = 8 TB/s over weights + the F16 KV bytes a token reads at that context."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.engine as E  # noqa: E402
from bench import layer_bytes, model_cfg  # noqa: E402
from prima_cpp_amd.lib import Q6_K, row_size  # noqa: E402

hp, mixture, name = model_cfg(sys.argv[1] if len(sys.argv) > 1 else "llama3-70b")
ctxs = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["512", "2048", "8192", "32768"])]
n_ctx = max(ctxs) + 256
w = E.Window(hp, n_ctx=n_ctx)
w.fill_synthetic(mixture, seed=1234)
w.finalize(max_tokens=2048)
total_w = sum(layer_bytes(hp, mixture)) + row_size(Q6_K, hp["n_embd"]) * hp["n_vocab"] + hp["n_embd"] * 4
rng = np.random.default_rng(3)
pos = 0
for ctx in ctxs:
    while pos < ctx:                                  # extend the cache to `ctx` cells
        n = min(2048, ctx - pos)
        toks = torch.from_numpy(rng.integers(0, hp["n_vocab"], n).astype(np.int32)).cuda()
        if n >= 16:
            w.decode(tokens=toks, pos0=pos, want_hidden=False, want_logits=False)
        else:
            for i in range(n):
                w.decode(tokens=toks[i:i + 1], pos0=pos + i, want_hidden=False, want_logits=False)
        pos += n
    io = torch.zeros(64, dtype=torch.int32, device="cuda")
    io[0] = 7
    w.generate(io, pos, 8, use_graph=True)            # warm-up (captures the graph of this regime)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w.generate(io, pos + 8, 24, use_graph=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 24
    pos += 32
    kv = hp["n_layer"] * 2 * hp["head_dim"] * hp["n_head_kv"] * 2 * pos
    roof = 8e12 / (total_w + kv)
    print(f"{name} n_kv ~{pos:6d}: {1 / dt:7.2f} tok/s ({dt * 1e3:.3f} ms)  KV {kv / 1e9:5.2f} GB/token  roofline {roof:6.1f} tok/s -> {1 / dt / roof * 100:4.1f} %")
w.close()
