"""Round 5: the persistent decode engine (csrc/decode_engine.hip) against the five-launch path on the SAME synthetic window: per-token logits of a
teacher-forced token sequence (NMSE; the two forms differ only in ffn_down's summation order, ~1e-13), the engine's watchdog word, and the
graph-replayed step time of both forms.   python tools/engine_check.py [--model llama3-70b] [--layers 4] [--tokens 40] [--time-steps 64]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.engine as E  # noqa: E402


def run(hp, mix, layers, tokens, engine, time_steps, n_ctx, start_pos):
    os.environ["PM355_ENGINE"] = "1" if engine else "0"
    hp = dict(hp)
    full = hp["n_layer"]
    if layers:
        hp["n_layer"] = layers
    win = E.Window(hp, n_ctx=n_ctx)
    if layers:
        win.hp_dict = dict(hp, n_layer=full)        # keep the type mixture of the full model's first layers
    win.fill_synthetic(mix, seed=1234)
    win.finalize(max_tokens=1)
    tok = torch.zeros(1, dtype=torch.int32, device="cuda")
    lg = torch.empty(hp["n_vocab"], dtype=torch.float32, device="cuda")
    xo = torch.empty((1, hp["n_embd"]), dtype=torch.float32, device="cuda")
    win.set_pos(start_pos)
    out = []
    for t in tokens:
        tok[0] = int(t)
        win.step(token=tok, x_out=xo, logits=lg, advance=1, use_graph=True)
        torch.cuda.synchronize()
        out.append((xo.cpu().numpy().copy(), lg.cpu().numpy().copy()))
    rc = win.check()
    us = None
    if time_steps:
        am = torch.zeros(1, dtype=torch.int32, device="cuda")
        win.set_pos(start_pos)
        for _ in range(4):
            win.step(token=tok, argmax=am, advance=1, use_graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(time_steps):
            win.step(token=tok, argmax=am, advance=1, use_graph=True)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / time_steps * 1e6
        rc = rc or win.check()
    win.close()
    return out, rc, us


def nmse(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-70b")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=40)
    ap.add_argument("--time-steps", type=int, default=64)
    ap.add_argument("--n-ctx", type=int, default=4096)
    ap.add_argument("--start-pos", type=int, default=0)
    a = ap.parse_args()
    hp = {"llama3-70b": E.LLAMA3_70B, "llama3-8b": E.LLAMA3_8B, "qwen2.5-72b": E.QWEN25_72B}[a.model]
    mix = E.q6_k_types if a.model.startswith("qwen") else E.q4_k_m_types
    rng = np.random.default_rng(7)
    tokens = rng.integers(0, hp["n_vocab"], a.tokens)
    ref, rc0, us0 = run(hp, mix, a.layers, tokens, False, a.time_steps, a.n_ctx, a.start_pos)
    eng, rc1, us1 = run(hp, mix, a.layers, tokens, True, a.time_steps, a.n_ctx, a.start_pos)
    print("per-token hidden NMSE:", " ".join(f"{nmse(e[0], r[0]):.1e}" for e, r in zip(eng, ref)))
    worst_h = max(nmse(e[0], r[0]) for e, r in zip(eng, ref))
    worst_l = max(nmse(e[1], r[1]) for e, r in zip(eng, ref))
    same = sum(int(np.array_equal(e[1], r[1])) for e, r in zip(eng, ref))
    fin = all(np.isfinite(e[1]).all() for e in eng)
    am = sum(int(e[1].argmax() == r[1].argmax()) for e, r in zip(eng, ref))
    print(f"{a.model} {a.layers or hp['n_layer']} layers, {a.tokens} tokens from position {a.start_pos}: engine vs five launches - worst hidden-row NMSE {worst_h:.2e}, "
          f"worst logits NMSE {worst_l:.2e}, bit-identical logits {same}/{a.tokens}, same argmax {am}/{a.tokens}, finite {fin}; watchdog launches {rc0} engine {rc1}")
    if us0 and us1:
        print(f"step time (graph replay, host clock, {a.time_steps} steps): five launches {us0:.1f} us, engine {us1:.1f} us  ({us0 / us1:.3f} x)")
    ok = rc0 == 0 and rc1 == 0 and fin and worst_l < 1e-9
    print("ENGINE_CHECK", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
