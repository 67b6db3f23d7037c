"""A few T-token steps of a 70B-shaped window (8 layers, no head) for rocprofv3 --kernel-trace --stats: which launches a small batch costs.
usage: rocprofv3 --kernel-trace --stats -d gpurun_out/bs -- python tools/batch_step_profile.py [T] [steps]; then tools/batch_step_summary below
prints per-kernel time per step."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prima_cpp_amd.engine as E  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
QWEN = os.environ.get("PROBE_MODEL") == "qwen"
hp = dict(E.QWEN25_72B if QWEN else E.LLAMA3_70B); hp["n_layer"] = 8
win = E.Window(hp, lo=0, hi=8, flags=0, n_ctx=1024)
win.fill_synthetic(E.q6_k_types if QWEN else E.q4_k_m_types, seed=7)
win.finalize(max_tokens=T, n_seq=1)
x = torch.randn(T, hp["n_embd"], device="cuda") * 0.1
for i in range(steps + 1):
    win.decode(x_in=x, pos0=T * i, want_hidden=True, want_logits=False)
torch.cuda.synchronize()
win.close()
