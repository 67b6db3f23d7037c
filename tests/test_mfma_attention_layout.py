"""Host-side model of the operand layouts of csrc/attn_flash_mfma.hip (no GPU): v_mfma_f32_16x16x32_f16 is emulated lane by lane with the
register maps of the CDNA4 ISA (A: lane l holds row l % 16, k = 8 (l / 16) .. + 7; B: column l % 16, same k; C / D: column l % 16, rows
4 (l / 16) + r), the kernel's operand construction is restated in numpy, and the claims its header makes are checked:

  * with score-tile rows permuted (row i of tile A = key 8 (i / 4) + i % 4, tile B = that + 4) the 8 accumulator values a lane holds after
    the two score MFMAs are the scores of keys 8 g .. 8 g + 7 of ITS column's head - contiguous keys, so the key-validity test and the mask
    load of the kernel index the right cells;
  * the same 8 values, rounded to F16, are exactly the B operand the P.V MFMA wants from that lane when the A operand is ONE 16-byte load
    of a transposed-V row at key offset 8 g: O^T = V^T P^T comes out right without any exchange between lanes;
  * the O^T accumulator of lane (head, g), register r of tile d is head dimension 16 d + 4 g + r (what the wave merge writes to LDS).

This is the layout argument of the kernel, executable; the numerics are covered on the GPU by tests/test_gpu_ops.py."""
import numpy as np


def mfma_16x16x32(a_regs, b_regs, c_regs):
    """a_regs, b_regs: [64 lanes][8]; c_regs: [64][4] -> D = A B + C in the same layout"""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        for j in range(8):
            A[l % 16, 8 * (l // 16) + j] = a_regs[l, j]
            B[8 * (l // 16) + j, l % 16] = b_regs[l, j]
    D = A @ B
    out = c_regs.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[4 * (l // 16) + r, l % 16]
    return out


def test_key_permutation_makes_p_the_b_operand_of_the_pv_product():
    rng = np.random.default_rng(5)
    DH, R, n_ctx, kt = 128, 8, 96, 32
    K = rng.normal(0, 1, (n_ctx, DH)).astype(np.float16).astype(np.float64)          # cache rows of one KV head
    Vt = rng.normal(0, 1, (DH, n_ctx)).astype(np.float16).astype(np.float64)         # transposed V cache of that head
    Q = rng.normal(0, 1, (R, DH)).astype(np.float16).astype(np.float64)
    lanes = np.arange(64)
    col, lg = lanes % 16, lanes // 16
    krow_i = 8 * (col // 4) + (col % 4)                                               # the kernel's krow_i
    # ---- S^T = K Q^T: two 16-row tiles, DH / 32 k-steps; B operand = q[head][32 kk + 8 lg ..]
    acc = [np.zeros((64, 4)), np.zeros((64, 4))]
    for t in range(2):
        for kk in range(DH // 32):
            a = np.stack([K[kt + krow_i[l] + 4 * t, 32 * kk + 8 * lg[l]: 32 * kk + 8 * lg[l] + 8] for l in lanes])
            b = np.stack([Q[min(col[l], R - 1), 32 * kk + 8 * lg[l]: 32 * kk + 8 * lg[l] + 8] * (col[l] < R) for l in lanes])
            acc[t] = mfma_16x16x32(a, b, acc[t])
    S = K[kt:kt + 32] @ Q.T                                                           # [key][head]
    for l in lanes:
        if col[l] >= R:
            continue
        got = [acc[r >> 2][l, r & 3] for r in range(8)]                                # the kernel's s[r] = acc[r >> 2][r & 3]
        want = [S[8 * lg[l] + r, col[l]] for r in range(8)]                            # ... is key kt + 8 lg + r
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12), l
    # ---- O^T = V^T P^T with P = the lane's own 8 values as B operand, A = one 16-byte load of V^T row (16 d + col) at key kt + 8 lg
    P = np.exp(S - S.max(axis=0, keepdims=True))                                      # any per-head positive weights will do
    o = []
    for d in range(DH // 16):
        a = np.stack([Vt[16 * d + col[l], kt + 8 * lg[l]: kt + 8 * lg[l] + 8] for l in lanes])
        b = np.stack([[P[8 * lg[l] + r, col[l]] if col[l] < R else 0.0 for r in range(8)] for l in lanes])
        o.append(mfma_16x16x32(a, b, np.zeros((64, 4))))
    O = Vt[:, kt:kt + 32] @ P                                                         # [dim][head]
    for l in lanes:
        if col[l] >= R:
            continue
        for d in range(DH // 16):
            for r in range(4):
                assert np.isclose(o[d][l, r], O[16 * d + 4 * lg[l] + r, col[l]], rtol=1e-12, atol=1e-12), (l, d, r)


def test_rows_of_the_two_score_tiles_cover_every_key_of_the_tile_once():
    col = np.arange(16)
    a = 8 * (col // 4) + col % 4
    assert sorted(np.concatenate([a, a + 4]).tolist()) == list(range(32))


def test_merger_partition_covers_every_output_and_partial_exactly_once():
    """The span merge of attn_flash_mfma.hip, as index arithmetic: NM = min(8, nact) mergers, merger `me` owns float4 items [me ipm, me ipm + cnt),
    a thread unit u = (item, slice) sums the partials [sl nact / SL, (sl + 1) nact / SL) - at most 8 per unit, at most 2 x 256 units per
    merger. Every (output item, partial) pair must be summed exactly once, for every head count, head_dim and number of active spans."""
    for DH in (64, 128):
        for R in range(1, 17):
            items = R * DH // 4
            for nact in range(1, 65):
                NM = min(8, nact)
                ipm = (items + NM - 1) // NM
                seen = np.zeros((items, nact), dtype=np.int32)
                for me in range(NM):
                    i0 = me * ipm
                    cnt = max(0, min(ipm, items - i0))
                    SL = min(8, nact)
                    if cnt * SL > 512:
                        SL = 512 // cnt
                    assert SL >= 1 and cnt * SL <= 512, (DH, R, nact, me)
                    for u in range(cnt * SL):
                        item, sl = u % cnt, u // cnt
                        c_lo, c_hi = sl * nact // SL, (sl + 1) * nact // SL
                        assert c_hi - c_lo <= 8, (DH, R, nact, me, sl)
                        seen[i0 + item, c_lo:c_hi] += 1
                assert (seen == 1).all(), (DH, R, nact)
