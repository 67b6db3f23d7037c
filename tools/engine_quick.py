"""A few decode steps of an n-layer window of a named shape (debug helper): python tools/engine_quick.py <model> <layers> [graph=1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prima_cpp_amd.engine as E
name, L = sys.argv[1], int(sys.argv[2]); graph = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
hp = dict({"llama3-70b": E.LLAMA3_70B, "llama3-8b": E.LLAMA3_8B, "qwen2.5-72b": E.QWEN25_72B}[name]); full = hp["n_layer"]; hp["n_layer"] = L
mix = E.q6_k_types if name.startswith("qwen") else E.q4_k_m_types
w = E.Window(hp, n_ctx=512); w.hp_dict = dict(hp, n_layer=full)
w.fill_synthetic(mix, seed=1234); w.finalize(max_tokens=1)
tok = torch.zeros(1, dtype=torch.int32, device="cuda"); am = torch.zeros(1, dtype=torch.int32, device="cuda")
w.set_pos(0)
xo = torch.zeros(1, hp["n_embd"], device="cuda"); lg = torch.zeros(hp["n_vocab"], device="cuda")
for i in range(4):
    w.step(token=tok, x_out=xo, logits=lg, argmax=am, use_graph=graph); torch.cuda.synchronize()
    print("step", i, int(am.item()), "hidden max", float(xo.abs().nan_to_num(0, 0, 0).max()), "nan", int(xo.isnan().sum()), "inf", int(xo.isinf().sum()),
          "logits nan", int(lg.isnan().sum()), "inf", int(lg.isinf().sum()), "max", float(lg.nan_to_num(0, 0, 0).abs().max()), flush=True)
w.close(); print("done")
