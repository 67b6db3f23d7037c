"""Time per T-token step of a 70B-shaped window (16 layers, no head), HIP events around 6 steps after a warm-up - per layer, so that box-to-box
A/B comparisons of the small-batch path do not need the whole model.   usage: python tools/batch_step_time.py [T ...]   (env toggles apply)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prima_cpp_amd.engine as E  # noqa: E402

Ts = [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4, 8, 16]
NL = 16
QWEN = os.environ.get("PROBE_MODEL") == "qwen"                 # Qwen2.5-72B "Q6_K" file type: every matrix Q6_K, ffn_down (K = 29568) Q8_0
hp = dict(E.QWEN25_72B if QWEN else E.LLAMA3_70B); hp["n_layer"] = NL
win = E.Window(hp, lo=0, hi=NL, flags=0, n_ctx=1024)
win.fill_synthetic(E.q6_k_types if QWEN else E.q4_k_m_types, seed=7)
win.finalize(max_tokens=max(Ts), n_seq=1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for T in Ts:
    x = torch.randn(T, hp["n_embd"], device="cuda") * 0.1
    win.kv_clear()
    for i in range(2): win.decode(x_in=x, pos0=T * i, want_hidden=True, want_logits=False)
    torch.cuda.synchronize()
    e0.record()
    for i in range(6): win.decode(x_in=x, pos0=T * (i + 2), want_hidden=True, want_logits=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 6 / NL * 1e3
    print(f"T={T:2d}: {us:7.1f} us per layer  (x 80 layers = {us * 80 / 1e3:6.2f} ms per step)", flush=True)
win.close()
