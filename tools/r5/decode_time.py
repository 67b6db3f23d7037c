"""Decode time per token of the resident engine with whatever library PM355_LIB names (only entry points that exist since round 2):
fill a synthetic Llama-3-70B Q4_K_M window, 16-token warm-up, then `n` greedy tokens by pm355_model_generate (captured graph).   python tools/r5/decode_time.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import prima_cpp_amd.engine as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
start = int(sys.argv[2]) if len(sys.argv) > 2 else 24           # position the timed tokens start at (cells before it: whatever the cache holds)
w = E.Window(E.LLAMA3_70B, n_ctx=4096)
w.fill_synthetic(E.q4_k_m_types, seed=1234)
w.finalize(max_tokens=1)
io = torch.zeros(256, dtype=torch.int32, device="cuda"); io[0] = 7
w.generate(io, 0, 24, use_graph=True)
if start > 24:
    w.generate(io, start - 8, 8, use_graph=True)               # (captures the graph of that regime)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    w.generate(io, start + rep * n, n, use_graph=True)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / n)
print(f"DECODE_TIME {os.path.basename(os.path.dirname(os.environ.get('PM355_LIB', 'cur/x')))} {best * 1e3:.4f} ms per token ({1 / best:.2f} tok/s)")
w.close()
