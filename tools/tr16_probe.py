"""What ds_read_b64_tr_b16 returns (gfx950): per 16-lane group a [4 rows][16 columns] block of 16-bit values in LDS, lane i supplies the address of
(row i / 4, columns 4 (i % 4) .. + 3). Expected: lane c receives column c, rows 0..3 - the transposed read the row-major-V attention operand needs."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from prima_cpp_amd import lib as L
P = L.load_probe()
P.pm355_probe_tr16.restype = C.c_int
P.pm355_probe_tr16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
g, r, c = torch.meshgrid(torch.arange(4), torch.arange(4), torch.arange(16), indexing="ij")
inp = (g * 1000 + r * 100 + c).to(torch.int16).reshape(-1).cuda()           # value = 1000 group + 100 row + column
out = torch.zeros(256, dtype=torch.int16, device="cuda")
assert P.pm355_probe_tr16(inp.data_ptr(), out.data_ptr(), None) == 0
torch.cuda.synchronize()
o = out.cpu().reshape(64, 4)
ok = True
for lane in range(64):
    want = [1000 * (lane // 16) + 100 * e + (lane % 16) for e in range(4)]
    if o[lane].tolist() != want:
        ok = False
        print("lane", lane, "got", o[lane].tolist(), "want", want)
print("ds_read_b64_tr_b16: lane c of a 16-lane group receives column c, rows 0..3 of the group's [4][16] block:", "CONFIRMED" if ok else "NOT what the hardware does")
