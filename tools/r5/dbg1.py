import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import prima_cpp_amd.ops as P
from _bind import rand_blocks, Q4_K
rng = np.random.default_rng(1)
K, N = 1024, 512
blocks = rand_blocks(Q4_K, N, K, rng)
w = P.upload_weight(Q4_K, blocks, K, N)
x = torch.from_numpy(rng.normal(0, 1.0, (1, K)).astype(np.float32)).cuda()
want = P.mul_mat_vec_fused([w], x)[0]
e = P.EngineRun()
(got,) = e.matvec([w], x)
e.run()
print("want[:8]", want[:8].cpu().numpy())
print("got [:8]", got[:8].cpu().numpy())
d = (got - want).abs()
print("max diff", d.max().item(), "n wrong", int((d > 1e-4).sum()), "of", N, "first wrong rows", torch.nonzero(d > 1e-4)[:16].flatten().cpu().numpy())
img = w.data.cpu().numpy().view(np.uint32)
print("W row0 first dword %08x  hdr0 %08x" % (img[0], img[(K // 256 * 128) // 4]))

xq = P.quantize_act(x)
y2, ip = P.mul_mat_vec_dbg(w, xq)
print('launch dbg row0 units 0..15 (isum, msum):', ip[0, :16].cpu().numpy().tolist())
print('y2[0]', y2[0].item())

raw = xq.cpu().numpy().view(np.uint8).reshape(-1)
q = raw[:K].view(np.int8)
d = raw[K:K + 4 * (K // 256)].view(np.float32)
print('standalone q[0..31]:', q[:32].tolist())
print('standalone d[0..3]:', d[:4].tolist())
print('x[0..7]:', x[0, :8].cpu().numpy().tolist())
