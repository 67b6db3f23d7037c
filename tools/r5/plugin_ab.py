"""Interleaved A/B of plug-in BUILDS on one box (the r03 -> r04 regression 94.9 -> 91.9 % / 514 -> 496 tok/s): the same driver binary and
libllama, libggml-mi355.so + libprima_mi355.so taken from ab/<tag>lib through LD_LIBRARY_PATH (the driver's RUNPATH comes after it).
   python tools/r5/plugin_ab.py r03,r04,cur 2"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _bind as B
from prima_cpp_amd import gguf as G

tags = (sys.argv[1] if len(sys.argv) > 1 else "r03,r04,cur").split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tmp = os.environ.get("TMPDIR", "/tmp")
p8 = os.path.join(tmp, "ab_8b.gguf"); p70 = os.path.join(tmp, "ab_70b16.gguf")
G.write_synthetic_model(p8, arch=0, n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256)
G.write_synthetic_model(p70, arch=0, n_layer=16, n_embd=8192, n_head=64, n_head_kv=8, n_ff=28672, n_vocab=128256, is_70b=True)
prompt = np.random.default_rng(1234).integers(0, 128256, 16); prompt[0] = 128000
for r in range(rounds):
    for tag in tags:
        env = {} if tag == "cur" else {"LD_LIBRARY_PATH": os.path.join(ROOT, "ab", tag + "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", "")}
        out = []
        for name, path in (("8B", p8), ("70B x 16 layers", p70)):
            _, _, st = B.run_llama_driver(path, prompt, 64, ngl=99, n_ctx=4096, threads=16, extra_args=["--keep-out-in-cuda"], timeout=300, env=env)
            out.append(f"{name}: {st['decode_tok_s']:7.2f} tok/s ({st['decode_ms_avg']:.4f} ms, best {st['decode_ms_min']:.4f})")
        print(f"round {r} {tag:4s} " + " | ".join(out), flush=True)
os.unlink(p8); os.unlink(p70)
