"""Step time of T-token batches through the whole 70B-shaped model (16-layer window x 5 -> 80 layers), for the crossover between the integer
small-batch path (mmq_i8.hip, T <= PM355_MMQ_MAX_TOKENS) and the F16 GEMM path (mmq.hip).   python tools/r5/small_cross.py 33,40,48,64"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import prima_cpp_amd.engine as E

Ts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "33,40,48,64").split(",")]
QWEN = os.environ.get("PROBE_MODEL") == "qwen"
hp = dict(E.QWEN25_72B if QWEN else E.LLAMA3_70B)
hp["n_layer"] = 16
from bench import model_cfg
_, mixture, _ = model_cfg("qwen2.5-72b" if QWEN else "llama3-70b")
w = E.Window(hp, n_ctx=1024)
w.fill_synthetic(mixture, seed=1)
w.finalize(max_tokens=max(Ts))
toks = torch.randint(0, hp["n_vocab"], (max(Ts),), dtype=torch.int32, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for T in Ts:
    w.kv_clear()
    w.decode(tokens=toks[:T], pos0=0, want_hidden=False, want_logits=False)
    torch.cuda.synchronize()
    e0.record()
    for i in range(4):
        w.decode(tokens=toks[:T], pos0=T * (i + 1), want_hidden=False, want_logits=False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 4 / 16
    print(f"T {T:3d}: {us:8.1f} us per layer  -> {us * 80 / 1e3:6.2f} ms per 80-layer step  (PM355_MMQ_MAX_TOKENS={os.environ.get('PM355_MMQ_MAX_TOKENS', '64')}, PROMPT_I8={os.environ.get('PM355_PROMPT_I8', '0')})")
w.close()
