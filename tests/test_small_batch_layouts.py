"""Index arithmetic of the round-4 small-batch path, restated in Python and checked exhaustively (no GPU): the producers and consumers of the
operand-ordered activation table agree on every byte, the K slices interleaved over the waves cover every step pair exactly once, the Q8_0
tail never counts a block twice, and the multi-column mat-vec's idle slot repeats a column without storing it. The formulas are the ones in
prima_cpp_amd/csrc/quantize.hip (q8k_block_from_regs), mmq_i8.hip (mmq_prep_kernel, mmq_prep_q80_kernel, ActSrc, MT<Q6_K>::operands, the
kernel's pb / pe / pstep) and mmvq_device.h (stage_finish, write_out); tests/test_gpu_ops.py runs the kernels themselves against the oracle."""
import itertools

import pytest


def table_offset_producer(row, blk, lane, i):
    """quantize.hip: lane `lane` of the wave that quantizes 256-block `blk` of activation row `row` holds values 4 lane .. 4 lane + 3 (byte i)."""
    return (row >> 5, blk * 8192 + (lane >> 3) * 1024 + ((((lane >> 2) & 1) * 32 + (row & 31)) * 16) + 4 * (lane & 3) + i)


def table_offset_prep(t, sb, s, g, b):
    """mmq_i8.hip mmq_prep_kernel: thread (t, g) copies the 16 bytes k = 32 s + 16 g .. of super-block sb."""
    return sb * 8192 + s * 1024 + (g * 32 + t) * 16 + b


def consumer_k(sb, s, t, g, b):
    """ActSrc::ld_sub(sb, s) of operand lane (t, g), byte b: the k index the matrix instruction of sub-block s expects there (A operand of
    v_mfma_i32_32x32x32_i8: lane (t, g) = 16 consecutive int8, k = 16 g .. 16 g + 15 of the 32-k slice)."""
    return 256 * sb + 32 * s + 16 * g + b


def test_quantizers_and_prologue_write_the_table_the_kernel_reads():
    for row in (0, 5, 31, 32, 40, 63):
        for blk in (0, 3):
            seen = {}
            for lane, i in itertools.product(range(64), range(4)):
                k = 256 * blk + 4 * lane + i
                pas, off = table_offset_producer(row, blk, lane, i)
                assert pas == row // 32
                seen[off] = k
            assert len(seen) == 256                                   # a row's block lands on 256 distinct bytes ...
            t = row & 31
            for s, g, b in itertools.product(range(8), range(2), range(16)):
                off = table_offset_prep(t, blk, s, g, b)              # ... the prologue kernel's and the quantizers' images coincide ...
                assert seen[off] == consumer_k(blk, s, t, g, b)       # ... and every byte is the k the operand lane wants
    # different rows of a pass never collide
    offs = {table_offset_producer(r, 0, lane, i)[1] for r in range(32) for lane in range(64) for i in range(4)}
    assert len(offs) == 32 * 256


def test_q6k_halves_come_out_of_the_16_byte_loads_by_the_lane_swap():
    """MT<Q6_K>::operands: lane (t, 0) / (t, 1) hold groups 2 s / 2 s + 1 whole; swapping the upper dwords of lane group 0 with the lower dwords of lane
    group 1 (v_permlane32_swap vdst = LO, src0 = HI) leaves, per 16-k product over group G, bytes 8 g .. 8 g + 7 of G in lane (t, g)."""
    for s in range(8):
        held = {g: [32 * s + 16 * g + b for b in range(16)] for g in (0, 1)}          # k indices of the 16 loaded bytes, per lane group
        lo = {g: held[g][:8] for g in (0, 1)}
        hi = {g: held[g][8:] for g in (0, 1)}
        # swap(vdst = LO, src0 = HI): vdst of group 1 <- src0 of group 0; src0 of group 0 <- vdst of group 1
        vdst = {0: lo[0], 1: hi[0]}
        src0 = {0: lo[1], 1: hi[1]}
        for g in (0, 1):
            even, odd = vdst[g], src0[g]
            assert even == [16 * (2 * s) + 8 * g + b for b in range(8)]
            assert odd == [16 * (2 * s + 1) + 8 * g + b for b in range(8)]


@pytest.mark.parametrize("ks_log2", [0, 1, 2, 3])
def test_interleaved_k_slices_cover_every_step_pair_once(ks_log2):
    KS = 1 << ks_log2
    for npairs in range(1, 70):
        seen = []
        for ks in range(KS):
            pb, pstep = ks, KS
            pe = ks if npairs <= ks else ks + ((npairs - ks + KS - 1) // KS) * KS
            pr = pb
            while pr < pe:
                prn = min(pr + pstep, pe - pstep)                      # the prefetch target stays one of the wave's own pairs
                assert pb <= prn < pe and (prn - pb) % KS == 0 and prn < npairs
                seen.append(pr)
                pr += pstep
        assert sorted(seen) == list(range(npairs)), (npairs, KS)


@pytest.mark.parametrize("K", [512, 608, 1184, 4096, 29568])
def test_q8_0_blocks_are_counted_once_also_past_the_row(K):
    """Q8_0: a step = 8 blocks, a "super-block" = 4; the second half of the last step may lie past the row - its blocks read as zeros (out-of-range
    buffer loads), they are never a clamped copy of real ones."""
    nb = K // 32
    nsb = (K + 127) // 128
    npairs = (nsb + 1) // 2
    counted = []
    for pr in range(npairs):
        for sb in (2 * pr, 2 * pr + 1):                                # (sb1 is NOT clamped for this type)
            for j in range(4):
                blk = 4 * sb + j
                in_range = blk * 1024 < nb * 1024 and blk * 128 < nb * 128     # the two buffer resources' bounds
                if in_range:
                    counted.append(blk)
    assert sorted(counted) == list(range(nb))
    # the table the prologue writes is the Q8_K one at block granularity: byte (blk, lane = 32 g + t, b) = k 32 blk + 16 g + b
    for blk, g, b in itertools.product((0, nb - 1), (0, 1), (0, 15)):
        assert table_offset_prep(7, blk // 8, blk % 8, g, b) == blk * 1024 + (g * 32 + 7) * 16 + b


@pytest.mark.parametrize("ncols,slots", [(2, 2), (3, 4), (4, 4), (8, 8)])
def test_idle_column_slot_repeats_a_column_and_is_not_stored(ncols, slots):
    read = [min(c, ncols - 1) for c in range(slots)]                    # stage_finish: column index a slot is filled from
    stored = [c for c in range(slots) if c < ncols]                     # write_out: slots that reach memory
    assert read[:ncols] == list(range(ncols)) and all(r == ncols - 1 for r in read[ncols:])
    assert stored == list(range(ncols))


@pytest.mark.parametrize("dh", [64, 128])
def test_row_major_v_image_feeds_the_transposing_read(dh):
    """attn_flash_mfma.hip, VROW: a wave writes its 32-key tile of row-major V into DH / 16 sub-tiles [32 rows][16 dims] (row = permuted key) and reads operand
    d of lane (col, lg) with two ds_read_b64_tr_b16 (lane i of a 16-lane group addresses row i / 4, columns 4 (i % 4) .. of a [4][16] block and RECEIVES column i,
    rows 0..3 - tools/tr16_probe.py). Modelled here: every (key, dim) is written once, and the operand comes out as keys 8 lg .. 8 lg + 7 of dim 16 d + col."""
    SUB, CPK, DT = 32 * 32 + 32, dh // 8, dh // 16
    image = {}
    for n in range(DT):                                               # the writer: load n, lane -> (key, 16-byte piece)
        for lane in range(64):
            kl, ch = n * (64 // CPK) + lane // CPK, lane % CPK
            pos = (16 + 4 * (kl >> 3) if kl & 4 else 4 * (kl >> 3)) + (kl & 3)
            base = (ch >> 1) * SUB + pos * 32 + (ch & 1) * 16
            for b in range(8):                                        # 8 halves of the piece: dims 8 ch .. 8 ch + 7
                assert base + 2 * b not in image
                image[base + 2 * b] = (kl, 8 * ch + b)
    assert len(image) == 32 * dh and sorted(k for k, _ in image.values()) == sorted(list(range(32)) * dh)
    for lg in range(4):
        for d in range(DT):
            for half, off in ((0, 0), (1, 512)):
                # addresses the 16 lanes of group lg supply, and the [4][16] block they describe
                block = {}
                for i in range(16):
                    a = d * SUB + (4 * lg + (i >> 2)) * 32 + (i & 3) * 8 + off
                    for e in range(4):
                        block[(i >> 2, 4 * (i & 3) + e)] = image[a + 2 * e]
                for col in range(16):                                 # lane `col` receives column col, rows 0..3
                    got = [block[(r, col)] for r in range(4)]
                    assert got == [(8 * lg + 4 * half + r, 16 * d + col) for r in range(4)], (lg, d, half, col, got)


# ---- round 6: the wq | wk | wv epilogue of the small-batch launch (mmq_i8.hip EPI instantiations) and the two-part grid (mmq_i8_dual_kernel)

def epi_slices(N, G):
    """mmq_i8_body<.., EPI = true>: workgroup w of G owns virtual rows [r0, r1), both forced even."""
    return [(((N * w) // G) & ~1, ((N * (w + 1)) // G) & ~1) for w in range(G)]


@pytest.mark.parametrize("N,G", [(9216, 224), (9216, 220), (10240, 256), (10240, 255), (1280, 40), (8192 + 1024, 37), (2304, 9)])
def test_rope_pairs_never_straddle_two_workgroups_and_every_row_is_owned_once(N, G):
    sl = epi_slices(N, G)
    assert sl[0][0] == 0 and sl[-1][1] == N                           # (N even: the last slice ends on N)
    owner = {}
    for w, (r0, r1) in enumerate(sl):
        assert r0 % 2 == 0 and r1 % 2 == 0 and r0 <= r1
        for row in range(r0, r1):
            assert row not in owner
            owner[row] = w
    assert len(owner) == N
    for row in range(0, N, 2):                                         # NORM rope: rows (2 i, 2 i + 1) of a head rotate together
        assert owner[row] == owner[row + 1]


def test_epilogue_thread_finds_its_pair_in_the_neighbouring_lane():
    """The reduction loop's thread o = (row group gi, result register v, lane l) holds (row, token) = (r0 + 32 (rg0 + gi) + l % 32, 8 (v / 4) + 4 (l / 32) + v % 4);
    lane l ^ 1 of the same (gi, v) holds row ^ 1 of the SAME token when r0 is even - the operand of __shfl_xor(s, 1)."""
    for r0 in (0, 2, 46, 1000):
        for rg in range(3):
            for v in range(16):
                for l in range(64):
                    row = r0 + 32 * rg + (l & 31)
                    t = 8 * (v >> 2) + 4 * (l >> 5) + (v & 3)
                    lp = l ^ 1
                    rowp = r0 + 32 * rg + (lp & 31)
                    tp = 8 * (v >> 2) + 4 * (lp >> 5) + (v & 3)
                    assert rowp == row ^ 1 and tp == t
    # every (row of a 32-row group, token of 32) appears exactly once over (v, l)
    seen = {(l & 31, 8 * (v >> 2) + 4 * (l >> 5) + (v & 3)) for v in range(16) for l in range(64)}
    assert len(seen) == 32 * 32


def dual_split(total_a, stride_a, Nb, stride_b, cus=256):
    """pm_launch_mmq_i8_dual: the device's workgroups divided by weight bytes; neither part gets more workgroups than it has 32-row groups."""
    ba, bb = total_a * stride_a, Nb * stride_b
    gb = int(cus * bb / (ba + bb) + 0.5)
    gb = max(gb, 1)
    gb = min(gb, (Nb + 31) // 32)
    ga = min(cus - gb, (total_a + 31) // 32)
    return ga, gb


@pytest.mark.parametrize("Na,Nb,K", [(9216, 1024, 8192), (4096 + 1024, 1024, 4096), (5120 + 1024, 1024, 5120), (300 + 70, 75, 1024), (2100 + 260, 300, 2048)])
def test_two_part_grid_gives_every_part_workgroups_and_whole_row_groups(Na, Nb, K):
    for stride_b in (K // 256 * 210, K // 256 * 176):                  # Q6_K / Q5_K rows against Q4_K's 144 bytes per super-block
        ga, gb = dual_split(Na, K // 256 * 144, Nb, stride_b)
        assert ga >= 1 and gb >= 1 and ga + gb <= 256
        assert gb <= (Nb + 31) // 32 and ga <= (Na + 31) // 32
        # the 70B shape: wq | wk take the bulk, wv (1024 rows) about one workgroup per 32 rows (Q6_K: 35.7 by bytes, capped at 32; Q5_K: 31)
        if (Na, Nb, K) == (9216, 1024, 8192):
            assert (ga, gb) == ((224, 32) if stride_b == K // 256 * 210 else (225, 31))
