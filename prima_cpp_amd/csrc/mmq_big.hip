// mmq_big.hip — prompt-sized (>= 128 tokens) quantized mat-mul on the INTEGER matrix cores: Q4_K / Q6_K weights x Q8_K activations.
//
// What it replaces: ggml_compute_forward_mul_mat (ggml/src/ggml.c:12377) running ggml_vec_dot_q4_K_q8_K / ggml_vec_dot_q6_K_q8_K
// (ggml-quants.c:7713 / :8918) once per (row, token) over activations it quantized with quantize_row_q8_K (:3785); the reference's CUDA
// plug-in serves the same batches with int8 MMA tiles over quantized activations (ggml-cuda/mmq.cuh:2583, quantize.cu:41-126). The
// arithmetic is the CPU reference's own - exact int32 sums  sum_s scale_s * (q_w . q_a)  and  sum_s min_s * bsum_s  per 256-weight
// super-block, ONE f32 multiply-add per super-block with d_w * d_a, super-blocks added in k order - so a prompt pass agrees with the CPU
// to f32 summation order, where the F16 GEMM (mmq.hip: dequantized F16 weights x F16 activations) is a different rounding of the problem.
// It is also the faster formulation on this chip: v_mfma_i32_32x32x32_i8 has twice the F16 rate, and the per-sub-block scale is ONE
// v_mad_i32_i24 per result (results of a lane belong to ONE weight row: the scale is a lane-local scalar), no dequantization at all.
//
// Mapping (same as mmq_i8.hip, whose operand layouts these are): one MFMA = 32 tokens x 32 weight rows x one 32-weight sub-block
// (Q4_K) / one 16-weight scale group (Q6_K, v_mfma_i32_32x32x16_i8); A = activations, lane (t, g) holds 16 (8) consecutive int8 of token t;
// B = weights, lane (r, g) the same k of row r; result lane (r, g) holds tokens 8 (v / 4) + 4 g + v % 4 of row r. The min / -32 terms are
// one F16 MFMA per super-block over the activations' 16-value sums (exact: integers < 2^11 in F16, f32 accumulation).
//
// Tiling: workgroup = 8 waves = 128 tokens x 256 rows (Q4_K; 128 x 128 for Q6_K whose stage is 1.5 x as large), wave tile 64 tokens x 64
// (32) rows, one K step = ONE super-block. Every byte of a stage travels global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging
// registers, no ds_write), double-buffered: the loads of super-block b + 1 fly while b is multiplied, one barrier per super-block.
// A DMA instruction writes 1 KiB of CONTIGUOUS LDS, so rows cannot be padded against bank conflicts; instead the 16-byte chunks are
// XOR-swizzled - the lane that fills LDS chunk p of a row fetches global chunk p ^ f(row) - such that the 16 lanes ds_read_b128 serves
// together hit 16 different bank groups:
//   activations  [128 tokens][16 chunks]:  chunk c of token t at t * 16 + (c ^ (t & 15))
//   weight nibbles (qa / qb; la / lb / qh)  [rows][4 chunks]:  chunk j of row n at n * 4 + (j ^ ((n >> 2) & 3))
// plus per stage the 16-byte headers / scales of the rows, and of the activation tables (pm_q8k_tables: the F16 group sums in A-operand order
// and the transposed block scales, written by the Q8_K quantizers as a second output) the four 32-token passes of this workgroup.
// Rows / tokens beyond N / T are clamped on the way in and not stored.
#include "pm355_device.h"
#include "pm355_kernels.h"
#include <mutex>

namespace {

typedef int      i32x16 __attribute__((ext_vector_type(16)));
typedef int      i32x4  __attribute__((ext_vector_type(4)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void * lds_ptr;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int TM = 128, NTHR = 512;

struct BigP {
    const uint8_t * W; long row_stride; int N, K, T;
    const uint8_t * xq; long xq_stride;                 // row-SoA Q8_K activations [T]
    const uint8_t * tab; long tab_bytes; int n_pass;    // activation tables: pass p (tokens 32 p ..) at tab + p * tab_bytes: [nsb][64][8 f16] | [nsb][32] f32
    float * y; long y_stride; const float * bias; const float * resid;
    int nt_t;                                           // token tiles (grid: consecutive workgroups of an XCD sweep the token tiles of one row tile)
};

__device__ __forceinline__ i32x16 mfma_i8x32(u32x4 a, u32x4 b, i32x16 c) {
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ long pk2(u32x2 a) { return (long) (((uint64_t) a[1] << 32) | a[0]); }
// 1 KiB of LDS at `dst` (wave-uniform) <- 64 x 16 bytes, lane l from its own global address `src`
__device__ __forceinline__ void dma16(const uint8_t * src, uint8_t * dst) {
    __builtin_amdgcn_global_load_lds((const PM_G void *) src, (lds_ptr) dst, 16, 0, 0);
}

template <int TYPE> struct BT;
template <> struct BT<PM_Q4_K> {
    static constexpr int NTN = 2, TN = 4 * 32 * NTN, NSTREAM = 2;          // row tiles per wave, rows per workgroup, 64-byte nibble streams per row
    static __device__ __forceinline__ long hdr_off(int nsb, int b) { return (long) nsb * 128 + (long) b * 16; }      // hdr stream at nb * 128 + b * 16
};
template <> struct BT<PM_Q6_K> {
    static constexpr int NTN = 1, TN = 4 * 32 * NTN, NSTREAM = 3;
    static __device__ __forceinline__ long hdr_off(int nsb, int b) { return (long) pm_q6k_sc_off((uint32_t) nsb, (uint32_t) b); }   // int8 scales (d: pm_q6k_d_off, read by the lanes)
};

// LDS stage layout
template <int TYPE> struct Stage {
    typedef BT<TYPE> B;
    static constexpr int ACT = 0, BS = TM * 256, DD = BS + 4 * 1024, WQ = DD + 512, HD = WQ + B::NSTREAM * B::TN * 64, BYTES = HD + B::TN * 16;
};

// ---- one stage = super-block b of a workgroup tile -> LDS buffer `buf`, by LDS-DMA; instruction q of the list is issued by wave q % 8.
//      Addresses = wave-uniform 64-bit base (SGPRs) + 32-bit lane offset, formed anew for every stage from an opaque copy of the lane id:
//      hoisted out of the k loop, the ~10 per-lane 64-bit addresses of a wave's instructions lived across the whole kernel (spills).
template <int TYPE>
__device__ __forceinline__ void issue_stage(const BigP & p, int b, uint8_t * buf, int T0, int N0, int P0, int nsb, int wave, int lane) {
    typedef BT<TYPE> B; typedef Stage<TYPE> S;
    constexpr int TN = B::TN;
    constexpr int NQ_ACT = TM / 4, NQ_W = B::NSTREAM * TN / 16, NQ_HD = TN / 64;
    constexpr int NQ = NQ_ACT + 4 + 1 + NQ_W + NQ_HD;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const uint8_t * xb = p.xq + (long) b * 256, * tbs = p.tab + (long) b * 1024, * tdd = p.tab + (long) nsb * 1024 + (long) b * 128;
    const uint8_t * wb = p.W + (long) b * 64, * hb = p.W + B::hdr_off(nsb, b);
    for (int q = wave; q < NQ; q += 8) {
        if (q < NQ_ACT) {                                   // 4 tokens x 256 bytes
            const int t = 4 * q + (ln >> 4), c = (ln & 15) ^ (t & 15);
            const uint32_t tg = (uint32_t) min(T0 + t, p.T - 1);
            dma16(xb + (tg * (uint32_t) p.xq_stride + (uint32_t) (c * 16)), buf + S::ACT + q * 1024);
        } else if (q < NQ_ACT + 4) {                        // F16 group sums of one 32-token pass, already in A-operand order
            const int i = q - NQ_ACT; const uint32_t ps = (uint32_t) min(P0 + i, p.n_pass - 1);
            dma16(tbs + (ps * (uint32_t) p.tab_bytes + (uint32_t) (ln * 16)), buf + S::BS + i * 1024);
        } else if (q == NQ_ACT + 4) {                       // block scales of the four passes: 4 x 128 bytes
            if (ln < 32) {
                const uint32_t ps = (uint32_t) min(P0 + (ln >> 3), p.n_pass - 1);
                dma16(tdd + (ps * (uint32_t) p.tab_bytes + (uint32_t) ((ln & 7) * 16)), buf + S::DD);
            }
        } else if (q < NQ_ACT + 5 + NQ_W) {                 // 16 rows x 64 bytes of one nibble stream
            const int i = q - (NQ_ACT + 5), stream = i / (TN / 16), rb = i % (TN / 16);
            const int n = 16 * rb + (ln >> 2), j = (ln & 3) ^ ((n >> 2) & 3);
            const uint32_t ng = (uint32_t) min(N0 + n, p.N - 1);
            dma16(wb + (long) stream * nsb * 64 + (ng * (uint32_t) p.row_stride + (uint32_t) (j * 16)), buf + S::WQ + stream * (TN * 64) + rb * 1024);
        } else {                                            // 64 rows x 16 bytes: Q4_K d | dmin | scales[12]; Q6_K int8 scales[16]
            const int i = q - (NQ_ACT + 5 + NQ_W);
            const uint32_t ng = (uint32_t) min(N0 + 64 * i + ln, p.N - 1);
            dma16(hb + ng * (uint32_t) p.row_stride, buf + S::HD + i * 1024);
        }
    }
}

// block id -> tile of the workgroup: XCD x = id % 8, slot = id / 8; an XCD's consecutive slots sweep the token tiles of one row tile, its row tiles are
// x, x + 8, ... (the workgroups that run together on an XCD share two row tiles of weights and the activations in its L2). false: no tile (grid padding)
__device__ __forceinline__ bool tile_of_block(const BigP & p, int tn, int & tile_t, int & tile_n) {
    const int nt_n = (p.N + tn - 1) / tn;
    const int id = (int) blockIdx.x, x = id & 7, slot = id >> 3;
    const int per_x = (nt_n + 7 - x) >> 3;                 // row tiles of XCD x
    if (slot >= per_x * p.nt_t) return false;
    tile_n = x + 8 * (slot / p.nt_t); tile_t = slot % p.nt_t;
    return true;
}

template <int TYPE>
__global__ __launch_bounds__(NTHR, 2) void mmq_big_kernel(BigP p) {
    typedef BT<TYPE> B; typedef Stage<TYPE> S;
    constexpr int NTN = B::NTN, TN = B::TN;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int r_ = lane & 31, g_ = lane >> 5;
    const int nsb = p.K / 256;
    // tile of this workgroup: block id -> (XCD x = id % 8, slot = id / 8); an XCD's consecutive slots sweep the token tiles of one row tile, its
    // row tiles are x, x + 8, ... (the workgroups that run together on an XCD share two row tiles of weights and the activations in its L2)
    const int nt_n = (p.N + TN - 1) / TN;
    int tile_t, tile_n;
    {
        const int id = (int) blockIdx.x, x = id & 7, slot = id >> 3;
        const int per_x = (nt_n + 7 - x) >> 3;                 // row tiles of XCD x
        if (slot < per_x * p.nt_t) { tile_n = x + 8 * (slot / p.nt_t); tile_t = slot % p.nt_t; }
        else return;                                           // (grid rounded up to 8 x max slots)
    }
    const int T0 = tile_t * TM, N0 = tile_n * TN;
    const int P0 = T0 >> 5;

    f32x16 out[NTN][2];
#pragma unroll
    for (int a = 0; a < NTN; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) out[a][m][v] = 0.0f;
    const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const f32x16 fz = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // Q6_K: the super-block scale d (fp16, one per row and super-block) is read by the lanes themselves, one stage ahead
    uint16_t dq[NTN] = {}, dq_next[NTN] = {};
    auto load_d = [&](int b, uint16_t (&d)[NTN]) __attribute__((always_inline)) {
        if constexpr (TYPE == PM_Q6_K) {
#pragma unroll
            for (int a = 0; a < NTN; ++a) {
                const int ng = min(N0 + (wn * NTN + a) * 32 + r_, p.N - 1);
                d[a] = *(const PM_G uint16_t *) (p.W + (long) ng * p.row_stride + (long) pm_q6k_d_off((uint32_t) nsb, (uint32_t) b));
            }
        }
    };
    load_d(0, dq_next);
    issue_stage<TYPE>(p, 0, smem, T0, N0, P0, nsb, wave, lane);
    for (int b = 0; b < nsb; ++b) {
        uint8_t * buf = smem + (b & 1) * S::BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's share of stage b (and its d values) has landed ...
        __syncthreads();                                         // ... and everybody's; everybody is done with the other buffer
#pragma unroll
        for (int a = 0; a < NTN; ++a) dq[a] = dq_next[a];
        if (b + 1 < nsb) { issue_stage<TYPE>(p, b + 1, smem + ((b + 1) & 1) * S::BYTES, T0, N0, P0, nsb, wave, lane); load_d(b + 1, dq_next); }
        // ---- multiply stage b
#pragma unroll
        for (int a = 0; a < NTN; ++a) {
            // (an opaque copy of the lane coordinates per row tile: formed once, the swizzled LDS offsets below would be hoisted out of the k loop and
            //  live - ~40 registers - next to the 64 accumulators)
            int r = r_, g = g_;
            asm volatile("" : "+v"(r), "+v"(g));
            const int n = (wn * NTN + a) * 32 + r;               // row inside the workgroup tile
            const int sw = (n >> 2) & 3;
            const u32x4 hd = *(const u32x4 *) (buf + S::HD + n * 16);
            if constexpr (TYPE == PM_Q4_K) {
                // the 8 six-bit scales / mins of the super-block, four per dword (get_scale_min_k4, ggml-quants.c:1898-1906, on whole dwords)
                const uint32_t sc4[2] = {hd[1] & 0x3f3f3f3fu, (hd[3] & 0x0f0f0f0fu) | ((hd[1] >> 2) & 0x30303030u)};
                const uint32_t mn4[2] = {hd[2] & 0x3f3f3f3fu, ((hd[3] >> 4) & 0x0f0f0f0fu) | ((hd[2] >> 2) & 0x30303030u)};
                f16x8 bm;
#pragma unroll
                for (int sb = 0; sb < 8; ++sb) bm[sb] = (_Float16) (float) ((mn4[sb >> 2] >> (8 * (sb & 3))) & 0xFFu);
                const float d = h2f((uint16_t) (hd[0] & 0xFFFF)), dmin = h2f((uint16_t) (hd[0] >> 16));
                // THE SCALE GOES INTO THE B OPERAND. Scaling the 32 x 32 result of every sub-block costs 16 v_mad per MFMA = twice the MFMA's own
                // time on the vector pipe (first version of this kernel: 590 TOP/s, VALU-bound). A 6-bit scale splits into two 3-bit halves,
                // scale = 8 hi + lo, and nibble x half <= 15 x 7 = 105 still IS an int8: two MFMAs per sub-block, B = q * lo and B = q * hi,
                // accumulate over the whole super-block inside the matrix pipe (C input), and  sum_s scale_s (q_s . a_s)  =  acc_lo + 8 acc_hi
                // exactly (|acc| <= 256 x 105 x 127). Four nibbles x half at once: v_pk_mul_lo_u16 on byte pairs (products < 256: no carries).
                const uint32_t lo4[2] = {sc4[0] & 0x07070707u, sc4[1] & 0x07070707u}, hi4[2] = {(sc4[0] >> 3) & 0x07070707u, (sc4[1] >> 3) & 0x07070707u};
                const uint8_t * wrow = buf + S::WQ + g * (TN * 64) + n * 64;      // this lane group's nibble stream of the row: qa (g = 0) / qb (g = 1)
                const int t0 = (wm * 2) * 32 + r, t1 = t0 + 32;                   // (r doubles as the token index of this lane in the A operand)
                const uint8_t * arow0 = buf + S::ACT + t0 * 256, * arow1 = buf + S::ACT + t1 * 256;
                const int x0 = t0 & 15, x1 = t1 & 15;
                i32x16 alo0 = zero, ahi0 = zero, alo1 = zero, ahi1 = zero;
                // (Measured and rejected, round 4: a wave = ONE 32-row group x all 128 tokens, token tiles walked one at a time with two accumulator sets so that
                //  the operand preparation serves four tiles and every tile's super-block epilogue hides behind the next tile's MFMAs - 7.7 instead of 11.9
                //  vector instructions per MFMA on paper, 256 VGPRs + 19 spilled and every wave reading the whole activation tile from LDS in practice:
                //  551 vs 657 TOP/s on ffn_gate.)
                // Software-pipelined by hand, two stages deep. Region j of the unit loop holds (1) the LDS reads of unit j + 1's activation operands
                // and unit j + 2's nibbles, fenced at the top; (2) the vector work that turns unit j + 1's nibbles into B operands, and (3) the eight
                // MFMAs of unit j - (2) and (3) are independent, so the wave's own vector instructions issue in the shadow of its MFMAs. Left alone
                // the scheduler puts every ds_read right in front of its MFMA and the operand preparation in front of the MFMAs that need it:
                // with two waves per SIMD running in phase the matrix and vector pipes then take turns (first measurements of this kernel:
                // time = MFMA time + VALU time).
                struct FragA { u32x4 a00, a01, a10, a11; };
                struct FragB { u32x4 bl0, bh0, bl1, bh1; };
                auto fetch_a = [&](FragA & f, int j) __attribute__((always_inline)) {        // sub-block s = activation chunk 2 s + g
                    f.a00 = *(const u32x4 *) (arow0 + (((4 * j + g) ^ x0) * 16)); f.a01 = *(const u32x4 *) (arow0 + (((4 * j + 2 + g) ^ x0) * 16));
                    f.a10 = *(const u32x4 *) (arow1 + (((4 * j + g) ^ x1) * 16)); f.a11 = *(const u32x4 *) (arow1 + (((4 * j + 2 + g) ^ x1) * 16));
                };
                auto fetch_w = [&](int j) __attribute__((always_inline)) { return *(const u32x4 *) (wrow + ((j ^ sw) * 16)); };   // unit j = sub-blocks 2 j (low nibbles), 2 j + 1 (high)
                auto prep = [&](FragB & f, const u32x4 & w, int j) __attribute__((always_inline)) {
                    // multipliers of sub-blocks 2 j, 2 j + 1 as u16 pairs {s, s}: byte (s & 3) of the scale dword into bytes 0 and 2
                    constexpr uint32_t SEL[4] = {0x0c000c00u, 0x0c010c01u, 0x0c020c02u, 0x0c030c03u};
                    const u16x2 ml0 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, lo4[(2 * j) >> 2], SEL[(2 * j) & 3]));
                    const u16x2 mh0 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, hi4[(2 * j) >> 2], SEL[(2 * j) & 3]));
                    const u16x2 ml1 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, lo4[(2 * j + 1) >> 2], SEL[(2 * j + 1) & 3]));
                    const u16x2 mh1 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, hi4[(2 * j + 1) >> 2], SEL[(2 * j + 1) & 3]));
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const u16x2 ql = __builtin_bit_cast(u16x2, w[c] & 0x0F0F0F0Fu), qh = __builtin_bit_cast(u16x2, (w[c] >> 4) & 0x0F0F0F0Fu);
                        f.bl0[c] = __builtin_bit_cast(uint32_t, (u16x2) (ql * ml0)); f.bh0[c] = __builtin_bit_cast(uint32_t, (u16x2) (ql * mh0));
                        f.bl1[c] = __builtin_bit_cast(uint32_t, (u16x2) (qh * ml1)); f.bh1[c] = __builtin_bit_cast(uint32_t, (u16x2) (qh * mh1));
                    }
                };
                FragA fa[2]; FragB fb[2]; u32x4 wr[2];
                fetch_a(fa[0], 0); wr[0] = fetch_w(0); wr[1] = fetch_w(1);
                prep(fb[0], wr[0], 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < 3) fetch_a(fa[(j + 1) & 1], j + 1);
                    const u32x4 wnext = wr[(j + 1) & 1];                           // nibbles of unit j + 1 (fetched two regions ago)
                    if (j < 2) wr[j & 1] = fetch_w(j + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < 3) prep(fb[(j + 1) & 1], wnext, j + 1);
                    const FragA & f = fa[j & 1]; const FragB & q = fb[j & 1];
                    alo0 = mfma_i8x32(f.a00, q.bl0, alo0); ahi0 = mfma_i8x32(f.a00, q.bh0, ahi0);
                    alo1 = mfma_i8x32(f.a10, q.bl0, alo1); ahi1 = mfma_i8x32(f.a10, q.bh0, ahi1);
                    alo0 = mfma_i8x32(f.a01, q.bl1, alo0); ahi0 = mfma_i8x32(f.a01, q.bh1, ahi0);
                    alo1 = mfma_i8x32(f.a11, q.bl1, alo1); ahi1 = mfma_i8x32(f.a11, q.bh1, ahi1);
                    if (j < 3) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f16x8 bs = *(const f16x8 *) (buf + S::BS + (wm * 2 + m) * 1024 + lane * 16);
                    const f32x16 ms = __builtin_amdgcn_mfma_f32_32x32x16_f16(bs, bm, fz, 0, 0, 0);       // sum_s min_s * (bsum[2s] + bsum[2s+1]), exact
                    const float * yd_lds = (const float *) (buf + S::DD) + (wm * 2 + m) * 32;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 yd = *(const f32x4 *) (yd_lds + 8 * q + 4 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int v = 4 * q + i;
                            const int isum = (m ? alo1[v] : alo0[v]) + 8 * (m ? ahi1[v] : ahi0[v]);
                            out[a][m][v] = fmaf(yd[i] * d, (float) isum, fmaf(-(yd[i] * dmin), ms[v], out[a][m][v]));
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);               // (one row tile at a time)
            } else {
                const float d = h2f(dq[a]);
                // sum_G scale_G * bsum_G (the -32 offset of every weight): B slot s of lane group g = scale[2 s + g]
                f16x8 bm;
#pragma unroll
                for (int s = 0; s < 8; ++s) { const int G = 2 * s + g; bm[s] = (_Float16) (float) (int) (int8_t) (hd[G >> 2] >> (8 * (G & 3))); }
                // the four 16-byte units of the row (u = 2 hh + v2), 8-byte half g of each stream
                u32x2 la[4], lb[4], qh[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int off = (n * 4 + (u ^ sw)) * 16 + 8 * g;
                    la[u] = *(const u32x2 *) (buf + S::WQ + off);
                    lb[u] = *(const u32x2 *) (buf + S::WQ + TN * 64 + off);
                    qh[u] = *(const u32x2 *) (buf + S::WQ + 2 * TN * 64 + off);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int t = (wm * 2 + m) * 32 + r;
                    const uint8_t * arow = buf + S::ACT + t * 256 + 8 * g;
                    i32x16 isum = zero;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int v2 = 0; v2 < 2; ++v2) {
                            const int u = 2 * hh + v2;
                            u32x2 qv[4];
                            qv[0] = (la[u] & 0x0F0F0F0Fu)        | ((qh[u] << 4) & 0x30303030u);
                            qv[1] = (lb[u] & 0x0F0F0F0Fu)        | ((qh[u] << 2) & 0x30303030u);
                            qv[2] = ((la[u] >> 4) & 0x0F0F0F0Fu) | (qh[u] & 0x30303030u);
                            qv[3] = ((lb[u] >> 4) & 0x0F0F0F0Fu) | ((qh[u] >> 2) & 0x30303030u);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int G = 8 * hh + 2 * c + v2;                      // group index inside the super-block = scale index
                                const int sc = (int) (int8_t) (hd[G >> 2] >> (8 * (G & 3)));
                                const u32x2 av = *(const u32x2 *) (arow + ((G ^ (t & 15)) * 16));            // k = 16 G + 8 g ..
                                const i32x16 acc = __builtin_amdgcn_mfma_i32_32x32x16_i8(pk2(av), pk2(qv[c]), zero, 0, 0, 0);
#pragma unroll
                                for (int v = 0; v < 16; ++v) isum[v] = __mul24(sc, acc[v]) + isum[v];        // |acc| <= 16 * 63 * 127
                            }
                        }
                    const f16x8 bs = *(const f16x8 *) (buf + S::BS + (wm * 2 + m) * 1024 + lane * 16);
                    const f32x16 ms = __builtin_amdgcn_mfma_f32_32x32x16_f16(bs, bm, fz, 0, 0, 0);
                    const float * yd_lds = (const float *) (buf + S::DD) + (wm * 2 + m) * 32;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 yd = *(const f32x4 *) (yd_lds + 8 * q + 4 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int v = 4 * q + i;
                            out[a][m][v] = fmaf(yd[i] * d, (float) (isum[v] - 32 * (int) ms[v]), out[a][m][v]);   // the reference's integer sum, then one rounding
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // ---- epilogue: lane (r, g), register v = row N0 + ... + r, token T0 + 32 (2 wm + m) + 8 (v / 4) + 4 g + v % 4
    //      (one quad of results at a time: with every address formed up front the 64 results + 64 addresses spill)
#pragma unroll
    for (int a = 0; a < NTN; ++a) {
        const int row = N0 + (wn * NTN + a) * 32 + r_;
        const bool row_ok = row < p.N;
        const float bias = (p.bias && row_ok) ? ld_g(p.bias + row) : 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t0 = T0 + (wm * 2 + m) * 32 + 8 * q + 4 * g_;
                const long o = (long) t0 * p.y_stride + row;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (row_ok && t0 + i < p.T) {
                        float s = out[a][m][4 * q + i] + bias;
                        if (p.resid) s += ld_g(p.resid + o + i * p.y_stride);
                        st_g(p.y + o + i * p.y_stride, s);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
}

} // namespace

size_t pm_mmq_big_table_bytes(int K, int T) { return (size_t) ((T + 31) / 32) * (size_t) (K / 256) * (1024 + 128); }

// 0 when pm_launch_mmq_big serves the shape
int pm_mmq_big_check(int type, int K, int N, int T) {
    if (type != PM_Q4_K && type != PM_Q6_K) return -1;
    if (T < 1 || K % 256 || K < 256 || N < 1) return -2;
    if (pm_q8k_row_bytes(K) % 16) return -2;                    // (16-byte DMA pieces of the activation rows: K % 1024 == 0)
    if ((size_t) T * pm_q8k_row_bytes(K) >= ((size_t) 1 << 32) || (size_t) N * pm_weight_row_stride(type, K) >= ((size_t) 1 << 32) ||
        pm_mmq_big_table_bytes(K, T) >= ((size_t) 1 << 32)) return -2;                   // (32-bit lane offsets)
    return 0;
}

// Y[t][n] = W[n,:] . x[t,:] (+bias[n]) (+resid[t][n]); xq = T rows of row-SoA Q8_K (quantize.hip), tab = the activation tables the quantizer
// wrote next to them (pm_q8k_tables with base = tab, tab_bytes = (K / 256) * 1152, one table per 32 rows). Y / resid token stride = N.
int pm_launch_mmq_big(int type, const void * W, const void * xq, const void * tab, float * Y, int K, int N, int T,
                      const float * bias, const float * resid, hipStream_t st) {
    const int rc = pm_mmq_big_check(type, K, N, T);
    if (rc) return rc;
    if (!W || !xq || !tab || !Y) return -2;
    BigP p = {};
    p.W = (const uint8_t *) W; p.row_stride = (long) pm_weight_row_stride(type, K); p.N = N; p.K = K; p.T = T;
    p.xq = (const uint8_t *) xq; p.xq_stride = (long) pm_q8k_row_bytes(K);
    p.tab = (const uint8_t *) tab; p.tab_bytes = (long) (K / 256) * (1024 + 128); p.n_pass = (T + 31) / 32;
    p.y = Y; p.y_stride = N; p.bias = bias; p.resid = resid;
    p.nt_t = (T + TM - 1) / TM;
    auto go = [&](auto kern, int tn, size_t lds) {
        const int nt_n = (N + tn - 1) / tn;
        const int slots = ((nt_n + 7) / 8) * p.nt_t;             // slots of the XCD with the most row tiles
        pm_allow_big_lds((const void *) kern, lds);
        hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(NTHR), lds, st, p);
    };
    if (type == PM_Q4_K) go(mmq_big_kernel<PM_Q4_K>, BT<PM_Q4_K>::TN, (size_t) 2 * Stage<PM_Q4_K>::BYTES);
    else                 go(mmq_big_kernel<PM_Q6_K>, BT<PM_Q6_K>::TN, (size_t) 2 * Stage<PM_Q6_K>::BYTES);
    return 0;
}

// ---- f32 activations: quantize to Q8_K (+ tables) into a scratch of this (device, stream), then the mat-mul ----------------------
namespace {
struct BigScr { hipStream_t st; uint8_t * p; size_t bytes; uint64_t use; };
BigScr g_big[16][8] = {};
uint64_t g_big_tick = 0;
std::mutex g_big_mu;
uint8_t * big_scratch(int dev, hipStream_t st, size_t need) {
    std::lock_guard<std::mutex> lk(g_big_mu);
    BigScr * e = nullptr, * lru = &g_big[dev][0];
    for (BigScr & c : g_big[dev]) {
        if (c.p && c.st == st) { e = &c; break; }
        if (!c.p) { if (lru->p) lru = &c; } else if (lru->p && c.use < lru->use) lru = &c;
    }
    if (e && e->bytes >= need) { e->use = ++g_big_tick; return e->p; }
    if (!e) e = lru;
    if (e->p) { (void) hipDeviceSynchronize(); (void) hipFree(e->p); e->p = nullptr; e->bytes = 0; }
    if (hipMalloc((void **) &e->p, need) != hipSuccess) { e->p = nullptr; return nullptr; }
    e->st = st; e->bytes = need; e->use = ++g_big_tick;
    return e->p;
}
} // namespace

// reuse_x != 0: the previous call on this stream had the same activations (x contents, T, K): its quantized copy and tables are reused
int pm_launch_mmq_big_f32(int type, const void * W, const float * X, float * Y, int K, int N, int T, const float * bias, const float * resid,
                          int reuse_x, hipStream_t st) {
    const int rc = pm_mmq_big_check(type, K, N, T);
    if (rc) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    const size_t tabs = (pm_mmq_big_table_bytes(K, T) + 255) & ~(size_t) 255, xrow = pm_q8k_row_bytes(K);
    uint8_t * sc = big_scratch(dev, st, tabs + (size_t) T * xrow + 256);
    if (!sc) return -3;
    pm_q8k_tables tb; tb.base = sc; tb.tab_bytes = (size_t) (K / 256) * (1024 + 128); tb.nsb = K / 256;
    if (!reuse_x) pm_launch_quantize_q8k(X, sc + tabs, K, T, st, tb);
    return pm_launch_mmq_big(type, W, sc + tabs, sc, Y, K, N, T, bias, resid, st);
}
