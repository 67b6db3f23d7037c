// mmq_pf.hip — prompt GEMM, third generation: quantized weights x F16 activations on the F16 matrix cores, every byte by LDS-DMA.
//
// Replaces ggml_compute_forward_mul_mat (ggml/src/ggml.c:12377) for prompt-sized batches (what the reference's CUDA plug-in serves with
// dequantize + cuBLAS / mul_mat_q, ggml-cuda.cu:1187-1285, mmq.cuh:2583):   Y[t][n] = sum_k W[n][k] X[t][k].
// Several matrices that share the activations (wq | wk | wv, ffn_gate | ffn_up) are JOBS of one launch, each with its own quant type.
//
// What the second generation (mmq.hip gemm_q_f16_kernel2) lost, by its own counters (profiles/r05_prefill_pmc.txt: MfmaUtil 48 %, waves in
// s_waitcnt): every thread fetched 16-byte weight pieces of its OWN row straight into registers - 512 cache lines touched per instruction
// round, 16 of 128 bytes used, the lines re-requested from L2 by the next three k-steps - and the compiler parked `s_waitcnt vmcnt(0)` in the
// middle of the MFMA stream; the dequantized F16 tile then went through ds_write + barrier + ds_read. Here:
//   * a wave owns 32 weight rows x ALL tokens of the tile (8 waves = 256 rows x 256 tokens; accumulators 8 x 16 = 128 registers). Its
//     weights never meet another wave: the packed bytes travel HBM -> LDS by LDS-DMA into a wave-private area (128 k per piece, two slots),
//     are read back 8 bytes per lane already in MFMA A-operand order (lane = row, k-half) and dequantized IN REGISTERS - packed-F16 math:
//     v_perm_b32 makes the halfs 1024 + q (low nibbles) / 64 + q (high nibbles, no shift), one packed add and one packed fma apply
//     d * sc and dmin * m. No F16 weight tile in LDS, no ds_write, no barrier on the weight path (the wave's own vmcnt orders it).
//   * the F16 activation tile (256 tokens x 64 k = 32 KiB per k-step) travels by LDS-DMA as whole 128-byte lines into a ring of three
//     buffers; the 16-byte chunks of a token row are XOR-swizzled on the SOURCE side (chunk c of token t lands at c ^ ((t >> 1) & 7)) so the
//     sixteen lanes ds_read_b128 serves together hit sixteen different bank groups. One barrier per k-step; step s + 2 is requested at
//     the top of step s, and the first fragments of step s + 1 are read before that barrier, so the matrix pipe never drains at it.
//   * no VGPR-destination global load in the k loop (Q6_K's 2-byte d excepted: once per 256 k): the compiler has nothing to wait for in
//     the MFMA stream; the only vmcnt wait is the explicit one in front of the barrier.
// Numerics: dequantized weight = F16(q * F16(d * sc) - F16(dmin * m)) with ONE rounding in the fma (kernel2 rounded the product first),
// F16 activations, f32 accumulation in k order - the reference's own GPU large-batch numerics (NMSE <= 5e-4, tests/test-backend-ops.cpp:1660;
// measured ~1e-6).
#include "pm355_device.h"
#include "pm355_kernels.h"
#include <stdlib.h>
#include <stdio.h>
#include <mutex>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void * lds_ptr;

constexpr int PF_NTHR = 512, PF_MAX_JOBS = 4;

struct PfJob {
    const uint8_t * W; float * Y; _Float16 * Yh; const float * bias; const float * resid; const float * silu_gate;
    long row_stride, ldy; int type, N, tile0;      // tile0: this job's first row tile in the launch's numbering of row tiles
};
struct PfP { PfJob job[PF_MAX_JOBS]; int njobs, nt_n, nt_t; const _Float16 * Xh; int K, T; int grp, pair; int splitk, full; float * ws; unsigned * cnt; unsigned long long * trace; };
// Ablations (measurement builds only: -DPM_GEMM_ABLATE=1 adds instantiations of the Q4_K 256-token kernel, PM355_GEMM_EXP=<bits> picks one; results are WRONG when set):
// 1 no activation DMA, 2 no weight DMA, 4 no vmcnt wait / barrier, 8 no B fragment reads, 16 no dequantization, 32 no MFMA, 64 no stores;
// 128: s_memtime stamps of super-block 5's second k-step (waves 0 and 4 of workgroup 0), printed by the launcher: the phase timeline
#ifndef PM_GEMM_ABLATE
#define PM_GEMM_ABLATE 0
#endif

// 1 KiB of LDS at byte address `dst` (wave-uniform) <- 64 x 16 bytes, lane l from base (wave-uniform) + its own 32-bit offset. Inline asm, not
// __builtin_amdgcn_global_load_lds: with an LDS-DMA the compiler can see in flight, hipcc (ROCm 7.2) turns every counted `s_waitcnt lgkmcnt(N)` of the
// loop into lgkmcnt(0) - each slice's MFMAs then wait for the fragment reads just issued for the NEXT slice. Invisible to it, the ds_read ladder
// stays counted; the DMA's completion is this kernel's own business (the explicit vmcnt(0) in front of every barrier).
__device__ __forceinline__ void dma16(const uint8_t * base, uint32_t voff, uint32_t dst) {
    unsigned keep;
    dst = (uint32_t) __builtin_amdgcn_readfirstlane((int) dst);          // (provably scalar for the "s" constraint: the compiler keeps some loop counters in VGPRs)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}
__device__ __forceinline__ half2v h2(uint32_t u) { return __builtin_bit_cast(half2v, u); }
__device__ __forceinline__ uint32_t u2(half2v v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ half2v pk_fma(half2v a, half2v b, half2v c) { return __builtin_elementwise_fma(a, b, c); }

template <int TYPE> struct PfT;
// NSTREAM: 1-KiB DMA pieces per 128-k slot (32 rows x 32 bytes each); EXTRA: bytes behind the two header slots (Q5_K: two super-blocks' qh, Q6_K: 8 super-blocks' d)
template <> struct PfT<PM_Q4_K> { static constexpr int NSTREAM = 2, EXTRA = 0; };
template <> struct PfT<PM_Q5_K> { static constexpr int NSTREAM = 2, EXTRA = 2048; };
template <> struct PfT<PM_Q6_K> { static constexpr int NSTREAM = 3, EXTRA = 512; };
// Q8_0 (row-SoA qa[K/32][16] | qb[K/32][16] | half d[K/32]): slots of 64 k = one k-step (two blocks: 1 KiB of qa, 1 KiB of qb per wave), requested two steps ahead
template <> struct PfT<PM_Q8_0> { static constexpr int NSTREAM = 2, EXTRA = 0; };
template <int TYPE> constexpr int pf_wave_bytes() { return 2 * PfT<TYPE>::NSTREAM * 1024 + 1024 + PfT<TYPE>::EXTRA; }

// the k loop + epilogue of one workgroup tile: rows [n0, n0 + 256) of job jb, tokens [t0, t0 + 32 NT)
// role: 0 plain; 1 / 2 = the gate / up half of a PAIR tile (waves 0-3 multiply ffn_gate's rows [n0, n0 + 128), waves 4-7 the same rows of ffn_up; the gate
// accumulators cross over through LDS after the k loop and the up waves store silu(gate) * up: ffn_gate's result never travels to HBM)
template <int TYPE, int NT, int EXP>
__device__ __forceinline__ void pf_body(const PfP & p, const PfJob & jb, const int n0, const int t0, uint8_t * smem, const int role, const int ks, const int S, const int tile_id) {
    constexpr int BBUF = NT * 32 * 128;                       // one activation buffer: 32 NT tokens x 64 halfs
    constexpr int NSTREAM = PfT<TYPE>::NSTREAM, SLOT = NSTREAM * 1024, AW = pf_wave_bytes<TYPE>();   // per wave: two 128-k slots + two 512-byte header slots + EXTRA
    constexpr bool K45 = TYPE == PM_Q4_K || TYPE == PM_Q5_K, Q8 = TYPE == PM_Q8_0;
    constexpr int NQ = NT >= 4 ? NT / 2 : 1;                  // activation DMA instructions per wave and k-step (NT = 2: 64 token rows = one per wave)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // nb = super-blocks of 256 k the loop walks; Q8_0 rows may end on half of one (K % 256 == 128: Qwen2.5-72B's ffn_down, K = 29568) - the weight operand of
    // the missing half is zeroed, its look-ahead requests are clamped to the last real k-step
    const int K = p.K, nst = K / 64, nb = (K + 255) / 256;
    const bool half_tail = Q8 && (K & 255) != 0;
    constexpr int exp = EXP;
    uint8_t * const aw = smem + 3 * BBUF + wave * AW;
    const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) smem;     // LDS byte address of the tile memory
    const uint32_t aw_l = lds0 + 3 * BBUF + wave * AW;

    // ---- DMA sources. weights: lane (row r, 16-byte piece h of the 32 bytes a 128-k slot holds per row and stream)
    const uint8_t * const Wg = jb.W;
    const int nw = n0 + 32 * (role ? wave & 3 : wave);         // first row of this wave
    const uint32_t wrow = (uint32_t) min(nw + r, jb.N - 1) * (uint32_t) jb.row_stride;
    const uint32_t wsrc = wrow + 16u * (uint32_t) h;
    //      activations: instruction q of this wave = token rows 8 (wave + 8 q) .. + 8, lane (row lane / 8, LDS chunk lane % 8 <- global chunk ^ swizzle)
    const uint8_t * const Xg = (const uint8_t *) p.Xh;
    uint32_t xsrc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int t = 8 * (wave + 8 * q) + (lane >> 3);
        const int c = (lane & 7) ^ ((t >> 1) & 7);
        xsrc[q] = (uint32_t) min(t0 + t, p.T - 1) * (uint32_t) (2 * K) + 16u * (uint32_t) c;
    }
    // ---- fragment reads: B = activations, lane (token r of tile jt, k-half h): chunk 2 j + h of slice j at its swizzled place
    //      (slice j: XOR with j << 5 - the chunk index 2 j + h has j in bits 1..2, and so has the byte offset in bits 5..6)
    const uint32_t boff0 = (uint32_t) (r * 128 + ((h ^ ((r >> 1) & 7)) << 4));
    const uint32_t aoff = (uint32_t) (r * 16 + 8 * h);        // A: 8 bytes of row r in a [32 rows][16] piece image

    // one DMA instruction each (the waits below count them): part q of k-step s's activations, stream st of weight piece m, the header of super-block b
    auto issue_B = [&](int s, uint32_t buf, int q) __attribute__((always_inline)) {
        if (exp & 1) return;
        dma16(Xg, xsrc[q] + (uint32_t) min(s, nst - 1) * 128u, lds0 + buf + (wave + 8 * q) * 1024);
    };
    auto issue_A = [&](int m, int slot, int st) __attribute__((always_inline)) {    // 128-k piece m of the rows -> slot
        if (exp & 2) return;
        if (Q8) {                 // m = k-step: blocks 2 m + h of stream st (0 qa, 1 qb)
            const uint32_t sc = (uint32_t) min(m, nst - 1);
            dma16(Wg, wrow + (st ? (uint32_t) K / 2u : 0u) + (2u * sc + (uint32_t) h) * 16u, aw_l + slot * SLOT + st * 1024);
            return;
        }
        const uint32_t mc = (uint32_t) min(m, 2 * nb - 1);
        if (TYPE == PM_Q5_K)      // native 176-byte blocks: qs at + 48; lane (r, h) = bytes [16 st, + 16) of unit 2 (m % 2) + h
            dma16(Wg, wrow + (mc >> 1) * 176u + 48u + (mc & 1u) * 64u + 32u * (uint32_t) h + 16u * (uint32_t) st, aw_l + slot * SLOT + st * 1024);
        else
            dma16(Wg, wsrc + 32u * mc + (uint32_t) st * (uint32_t) nb * 64u, aw_l + slot * SLOT + st * 1024);
    };
    auto issue_H = [&](int b) __attribute__((always_inline)) {                       // header (Q4_K) / int8 scales (Q6_K) of super-block b
        if (exp & 2) return;
        const uint32_t bc = (uint32_t) min(b, nb - 1);
        const uint32_t off = TYPE == PM_Q4_K ? (uint32_t) nb * 128u + bc * 16u : TYPE == PM_Q5_K ? bc * 176u : Q8 ? (uint32_t) K + bc * 16u /* 8 blocks' d */ : pm_q6k_sc_off((uint32_t) nb, bc);
        if (h == 0) dma16(Wg, wrow + off, aw_l + 2 * SLOT + (b & 1) * 512);
    };
    auto issue_QH = [&](int b) __attribute__((always_inline)) {                      // Q5_K: the 32 qh bytes of super-block b, lane (r, h) = bytes [16 h, + 16)
        if (exp & 2) return;
        dma16(Wg, wrow + (uint32_t) min(b, nb - 1) * 176u + 16u + 16u * (uint32_t) h, aw_l + 2 * SLOT + 1024 + (b & 1) * 1024);
    };
    auto issue_D = [&](int g) __attribute__((always_inline)) {                       // Q6_K: the fp16 d of super-blocks 8 g .. 8 g + 7 (16 bytes per row)
        if (exp & 2) return;
        // (16 bytes from the group's first d: a last group of fewer than 8 super-blocks ends exactly at the 16-byte-rounded row stride, pm_q6k_row_stride)
        if (h == 0) dma16(Wg, wrow + pm_q6k_d_off((uint32_t) nb, (uint32_t) (8 * min(g, (nb - 1) >> 3))), aw_l + 2 * SLOT + 1024);
    };

    // ---- weight path, per 16-k slice i of a super-block (i = 0..15: k-step i / 4, slice j = i % 4; 128-k piece mm = i / 8, k-step ks = (i / 4) % 2 of it)
    struct Raw { u32x2 ql, qh; };
    auto read_raw = [&](int i, int b) __attribute__((always_inline)) {                // slice i of super-block b
        const int mm = (i >> 3) & 1, ks = (i >> 2) & 1, j = i & 3;
        Raw w;
        if (Q8) {
            // block j / 2 of the k-step (slot = the step's parity), its first (qa) or second (qb) sixteen values, bytes [8 h, + 8)
            w.ql = *(const u32x2 *) (aw + ((i >> 2) & 1) * SLOT + (j & 1) * 1024 + (j >> 1) * 512 + aoff);
            w.qh = u32x2{0, 0};
        } else if (K45) {
            // unit ks of the slot: qa = qs[0, 16) / qb = qs[16, 32) of its 64 weights; slices 0, 2 read qa, slices 1, 3 qb
            w.ql = *(const u32x2 *) (aw + mm * SLOT + (j & 1) * 1024 + ks * 512 + aoff);
            w.qh = u32x2{0, 0};
            if (TYPE == PM_Q5_K) w.qh = *(const u32x2 *) (aw + 2 * SLOT + 1024 + (b & 1) * 1024 + (j & 1) * 512 + aoff);   // qh[16 (j & 1) + 8 h, + 8): bit s = 5th bit of sub-block s
        } else {
            // 32-weight group g = 2 ks + j / 2 of the 128-weight half: ql stream (g & 1), piece v = j & 1, nibble ks; qh piece v, bit pair g
            const int v = j & 1, st = j >> 1;
            w.ql = *(const u32x2 *) (aw + mm * SLOT + st * 1024 + v * 512 + aoff);
            w.qh = *(const u32x2 *) (aw + mm * SLOT + 2048 + v * 512 + aoff);
        }
        return w;
    };
    u32x4 hd = {0, 0, 0, 0};                                   // Q4_K: d | dmin | scales[12];  Q6_K: int8 scales[16]
    uint16_t d6h = 0;                                          // Q6_K: the super-block's fp16 d
    auto read_hdr = [&](int b) __attribute__((always_inline)) {
        hd = *(const u32x4 *) (aw + 2 * SLOT + (b & 1) * 512 + r * 16);
        if (TYPE == PM_Q6_K) d6h = *(const uint16_t *) (aw + 2 * SLOT + 1024 + r * 16 + (b & 7) * 2);
    };
    half2v mul = {0, 0}, add = {0, 0}, muln = {0, 0}, addn = {0, 0};
    auto scales = [&](int i, half2v & mul, half2v & add) __attribute__((always_inline)) {   // multiplier / addend of slice i (constant per sub-block / 16-group)
        if (Q8) {
            const int q = i >> 1;                              // block of the super-block: its d is already an F16 number
            const _Float16 dd = __builtin_bit_cast(_Float16, (uint16_t) (hd[q >> 1] >> (16 * (q & 1))));
            mul = half2v{dd, dd};
        } else if (K45) {
            const int s = i >> 1;                              // 32-weight sub-block of the super-block
            int sc, mn;
            k4_scale_min(hd[1], hd[2], hd[3], s, sc, mn);
            const _Float16 ds = (_Float16) (h2f((uint16_t) (hd[0] & 0xFFFF)) * (float) sc), ms = (_Float16) (h2f((uint16_t) (hd[0] >> 16)) * (float) mn);
            mul = half2v{ds, ds}; add = half2v{(_Float16) -ms, (_Float16) -ms};
        } else {
            const int mm = (i >> 3) & 1, ks = (i >> 2) & 1, j = i & 3;
            const int G = 8 * mm + 2 * (2 * ks + (j >> 1)) + (j & 1);               // 16-weight group of the super-block
            const _Float16 dd = (_Float16) (h2f(d6h) * (float) (int) (int8_t) (hd[G >> 2] >> (8 * (G & 3))));
            mul = half2v{dd, dd};
        }
    };
    // 8 weights of slice i -> one A operand. Written stage by stage over the four independent half2 chains (and - perm - add - fma): a dependent vector
    // instruction issues ~8-10 cycles after its producer, four chains side by side hide that
    auto dequant = [&](const Raw & w, int i) __attribute__((always_inline)) {
        const int ks = (i >> 2) & 1, j = i & 3;
        uint32_t v[2];
        half2v t[4];
        if (Q8) {
            // int8 x -> byte x ^ 0x80 = x + 128 -> F16 1024 + 128 + x (exact) -> - 1152 -> * d: one rounding, dequantize_row_q8_0's d * x (ggml-quants.c:1616)
            const half2v bias = half2v{(_Float16) -1152.0f, (_Float16) -1152.0f};
#pragma unroll
            for (int c = 0; c < 2; ++c) v[c] = w.ql[c] ^ 0x80808080u;
#pragma unroll
            for (int c = 0; c < 2; ++c) { t[2 * c] = h2(__builtin_amdgcn_perm(0x64646464u, v[c], 0x05010400u)); t[2 * c + 1] = h2(__builtin_amdgcn_perm(0x64646464u, v[c], 0x07030602u)); }
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = t[c] + bias;
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = t[c] * mul;
        } else if (K45) {
            const bool hi = j >> 1;                            // slices 2, 3: the unit's second sub-block = high nibbles
            const half2v bias = hi ? half2v{(_Float16) -64.0f, (_Float16) -64.0f} : half2v{(_Float16) -1024.0f, (_Float16) -1024.0f};
            uint32_t ex[2] = {hi ? 0x54545454u : 0x64646464u, hi ? 0x54545454u : 0x64646464u};   // F16 exponent byte: 64 + m / 16 (m = nibble << 4) / 1024 + m
#pragma unroll
            for (int c = 0; c < 2; ++c) v[c] = w.ql[c] & (hi ? 0xF0F0F0F0u : 0x0F0F0F0Fu);
            if (TYPE == PM_Q5_K) {
                // the 5th bit (bit s of the qh bytes, s = sub-block): low nibbles - bit 4 of the value byte; high nibbles (value byte = q << 4, ulp 1/16) - it is
                // mantissa bit 8 = bit 0 of the half's HIGH byte, which v_perm takes from the exponent operand
                const int sb = i >> 1;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const uint32_t bit = (w.qh[c] >> sb) & 0x01010101u;
                    if (hi) ex[c] |= bit; else v[c] |= bit << 4;
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) { t[2 * c] = h2(__builtin_amdgcn_perm(ex[c], v[c], 0x05010400u)); t[2 * c + 1] = h2(__builtin_amdgcn_perm(ex[c], v[c], 0x07030602u)); }
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = t[c] + bias;
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = pk_fma(t[c], mul, add);
        } else {
            const int g = 2 * ks + (j >> 1);
            const half2v bias = half2v{(_Float16) -1056.0f, (_Float16) -1056.0f};  // 1024 + q - 32
            uint32_t lo4[2], hi2[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) lo4[c] = ks ? (w.ql[c] >> 4) & 0x0F0F0F0Fu : w.ql[c] & 0x0F0F0F0Fu;
#pragma unroll
            for (int c = 0; c < 2; ++c) hi2[c] = g == 0 ? (w.qh[c] << 4) & 0x30303030u : g == 1 ? (w.qh[c] << 2) & 0x30303030u : g == 2 ? w.qh[c] & 0x30303030u : (w.qh[c] >> 2) & 0x30303030u;
#pragma unroll
            for (int c = 0; c < 2; ++c) v[c] = lo4[c] | hi2[c];
#pragma unroll
            for (int c = 0; c < 2; ++c) { t[2 * c] = h2(__builtin_amdgcn_perm(0x64646464u, v[c], 0x05010400u)); t[2 * c + 1] = h2(__builtin_amdgcn_perm(0x64646464u, v[c], 0x07030602u)); }
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = t[c] + bias;
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = t[c] * mul;
        }
        return __builtin_bit_cast(half8, u32x4{u2(t[0]), u2(t[1]), u2(t[2]), u2(t[3])});
    };

    float16v acc[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[jt][e] = 0.0f;
    half8 bf[NT], af[2];                                       // B fragments of the slice (re-filled right behind its MFMAs), A operand of this / the next slice
    Raw raw;
    auto read_B = [&](uint32_t buf, int j, int jt) __attribute__((always_inline)) {
        return *(const half8 *) (smem + buf + jt * 4096 + (boff0 ^ (uint32_t) (j << 5)));
    };

    // ---- prologue: super-block 0's header, piece 0, k-steps 0 and 1
    uint32_t rd = 0, nx = BBUF, wr = 2 * BBUF;                 // ring: buffer of step s, s + 1, s + 2
    // split K: slice ks of S takes the super-blocks [b0, b1); all indices below stay absolute (slot parities, look-ahead clamps)
    const int b0 = (int) ((long) ks * nb / S), b1 = (int) ((long) (ks + 1) * nb / S);
    issue_H(b0); if (TYPE == PM_Q6_K) issue_D(b0 >> 3); if (TYPE == PM_Q5_K) issue_QH(b0);
#pragma unroll
    for (int st = 0; st < NSTREAM; ++st) { if (Q8) { issue_A(4 * b0, 0, st); issue_A(4 * b0 + 1, 1, st); } else issue_A(2 * b0, 0, st); }
#pragma unroll
    for (int q = 0; q < NQ; ++q) { issue_B(4 * b0, rd, q); issue_B(4 * b0 + 1, nx, q); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_hdr(b0);
    raw = read_raw(0, b0);
    scales(0, mul, add);
    af[0] = dequant(raw, 0);
    if (TYPE == PM_Q6_K) scales(1, mul, add);
    raw = read_raw(1, b0);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) bf[jt] = read_B(rd, 0, jt);
    // Two waves share a SIMD (waves w and w + 4): the second group runs HALF A SLICE behind the first, so that on every SIMD one wave is in the memory
    // half of a slice (DMA issue, fragment reads) while the other is in its MFMA half - in lock-step (first version of this kernel) both issued their
    // DMAs, then both read, then both multiplied, and the pieces added up instead of overlapping (profiles/r06_prefill_ablation.txt).
    const bool late = p.grp == 1 ? (wave & 1) : p.grp == 2 ? ((wave >> 1) & 1) : wave >= 4;
    if (late && !(exp & 4)) __builtin_amdgcn_s_barrier();

    unsigned long long ts[17] = {};
    auto stamp = [&](int k, int b) __attribute__((always_inline)) {
        if ((exp & 128) && b == b0 + 5) asm volatile("s_memtime %0" : "=s"(ts[k]));
    };
    for (int b = b0; b < b1; ++b) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = i & 3, sq = i >> 2;                  // slice of the k-step, k-step of the super-block
            if (sq == 1) stamp(4 * j, b);
            // ---- memory half of slice i.
            // DMA schedule of k-step s (at most two instructions per slice: an issue blocks the wave for 60-180 cycles): slice j carries part j of step
            // s + 2's activations; the next weight piece's streams ride in slices 1, 2, 3 of steps 0 and 2, the next header in slice 1 of step 1 (Q6_K:
            // the d's of the next eight super-blocks in slice 3 of step 0, when due). Everything requested in step s has landed when step s + 1's
            // slice 2 begins: `vmcnt(N)` there, N = what slices 0 and 1 of step s + 1 issued. For the activations a barrier follows before their first
            // reader (slice 3 of step s + 1 reads the fragments of step s + 2's first slice); the weights are the wave's own.
            if (j == 2 && !(exp & 3)) {
                constexpr int N8[4] = {3, 3, 3, 2}, N4[4] = {2, 2, 2, 1}, Q8N8[4] = {2, 3, 2, 2}, Q8N4[4] = {1, 2, 1, 1};
                const int n = Q8 ? (NT == 8 ? Q8N8[sq] : Q8N4[sq]) : NT == 8 ? N8[sq] : N4[sq];
                if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            }
            if (NT == 8 || (NT == 4 && (j & 1) == 0) || j == 0) issue_B(4 * b + sq + 2, wr, NT == 8 ? j : NT == 4 ? j >> 1 : 0);
            // (Q8_0: the 64-k slot of step s is free once its last slice has been read - behind slice 1 - and takes step s + 2's blocks in slices 2 and 3)
            if (Q8) { if (j >= 2) issue_A(4 * b + sq + 2, sq & 1, j - 2); }
            else if ((sq == 0 || sq == 2) && j >= 1 && j - 1 < NSTREAM) issue_A(2 * b + 1 + (sq >> 1), sq == 0 ? 1 : 0, j - 1);
            if (sq == 1 && j == 1) issue_H(b + 1);
            if (TYPE == PM_Q5_K && sq == 1 && j == 2) issue_QH(b + 1);
            // (the slot's last reader - d of super-block 8 g + 7 - ran at slice 13 of super-block 8 g + 6; the first reader of the new group runs at slice 13 of 8 g + 7)
            if (TYPE == PM_Q6_K && sq == 0 && j == 3 && (b & 7) == 7) issue_D((b >> 3) + 1);
            __builtin_amdgcn_sched_barrier(0);
            // the wave's vector work: the F16 operand of slice i + 1. HERE, not between the MFMAs: a packed-F16 instruction next to a wave's own MFMA costs its
            // full 8 cycles of issue (measured: 8 MFMAs + 22 vector instructions = 448 cycles, 56 per MFMA, while this half idled 250 cycles at its barrier)
            // (the multiplier of slice i + 2 is formed next to it, an independent chain: slice i + 1's was formed one slice ago)
            const bool newgrp = TYPE == PM_Q6_K || (i & 1) == 0;
            if (newgrp && !(exp & 16)) scales((i + 2) & 15, muln, addn);
            if (!(exp & 16)) af[(i + 1) & 1] = dequant(raw, (i + 1) & 15);
            if (Q8 && i >= 7 && i < 15 && half_tail && b == nb - 1) af[(i + 1) & 1] = half8{0, 0, 0, 0, 0, 0, 0, 0};      // (k >= K: the row has no such weights)
            if (newgrp) { mul = muln; add = addn; }
            __builtin_amdgcn_sched_barrier(0);
            if (sq == 1) stamp(4 * j + 1, b);
            if (!(exp & 4)) __builtin_amdgcn_s_barrier();
            if (sq == 1) stamp(4 * j + 2, b);
            // ---- MFMA half
            // the slice's MFMAs; fragment jt of slice i + 1 goes into the SAME registers right behind the MFMA that read them (at j == 3: slice 0 of the next
            // step, from the next ring buffer): a 1-KiB read takes the LDS ~13 cycles with four waves reading, well inside an MFMA's 32. The consumers sit
            // behind two barriers, so the scheduler cannot sink the reads to them.
            __builtin_amdgcn_s_setprio(1);
            {
                const uint32_t nbuf = j == 3 ? nx : rd;
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    if (!(exp & 32)) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1], bf[jt], acc[jt], 0, 0, 0);
                    if (!(exp & 8)) bf[jt] = read_B(nbuf, (j + 1) & 3, jt);
                }
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            raw = read_raw((i + 2) & 15, b + ((i + 2) >> 4));                       // (consumed after two barriers: no exposed latency)
            if (i == 13) read_hdr(b + 1);
            if (sq == 1) stamp(4 * j + 3, b);
            if (!(exp & 4)) __builtin_amdgcn_s_barrier();
            if (i == 7) stamp(16, b);
            if (j == 3) { const uint32_t t = rd; rd = nx; nx = wr; wr = t; }
        }
    }

    if ((exp & 128) && p.trace && blockIdx.x == 0 && lane == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        p.trace[34 + wave] = hwid;
    }
    if ((exp & 128) && p.trace && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 17; ++k) p.trace[(wave >> 2) * 17 + k] = ts[k];
    }
    f32x4 * const xlds = (f32x4 *) smem;
    if (role || S > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the last steps' look-ahead DMAs still land in the ring)
        if (!late && !(exp & 4)) __builtin_amdgcn_s_barrier();  // (the second group has passed one barrier more: level)
        __builtin_amdgcn_s_barrier();                           // every wave is out of the k loop: the tile memory is free
    }
    // ---- split K: every slice leaves its partial tile in a slab (lane-linear, 1 KiB per store instruction); the LAST arriver of a tile adds the S partials in
    //      slice order, so the sum does not depend on who arrives when, and runs the epilogue. Hand-off = plain stores, one agent-scope
    //      release before the ticket, one acquire behind it (cdna_hip_programming.md, Guideline 16 counter form); the counter returns to 0 for the next launch.
    if (S > 1) {
        f32x4 * const slab0 = (f32x4 *) p.ws + (size_t) tile_id * S * (8 * NT * 4 * 64);
        f32x4 * const mine = slab0 + (size_t) ks * (8 * NT * 4 * 64);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) *(PM_G f32x4 *) (mine + ((wave * NT + jt) * 4 + g) * 64 + lane) = f32x4{acc[jt][4 * g], acc[jt][4 * g + 1], acc[jt][4 * g + 2], acc[jt][4 * g + 3]};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *(volatile unsigned *) smem = __hip_atomic_fetch_add(p.cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const unsigned ticket = *(volatile unsigned *) smem;
        if (ticket != (unsigned) (S - 1)) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(p.cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        // (all S slabs from memory, the reducer's own included - a second register copy of the tile would not fit; one token tile at a time, fenced:
        //  left alone the scheduler puts every load of the tile in flight and spills the accumulators)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jt][e] = 0.0f;
        for (int q = 0; q < S; ++q) {
            const f32x4 * sl = slab0 + (size_t) q * (8 * NT * 4 * 64) + lane;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                f32x4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] = *(const PM_G f32x4 *) (sl + ((wave * NT + jt) * 4 + g) * 64);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[jt][4 * g + e] += v[g][e];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                        // (the ticket word is tile memory: pair tiles write there next)
    }
    // ---- pair tiles: the gate waves park their accumulators in LDS (same lane, same register index as the up wave that needs them)
    if (role) {
        if (role == 1) {
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int g = 0; g < 4; ++g) xlds[(((wave & 3) * NT + jt) * 4 + g) * 64 + lane] = f32x4{acc[jt][4 * g], acc[jt][4 * g + 1], acc[jt][4 * g + 2], acc[jt][4 * g + 3]};
        }
        __syncthreads();
        if (role == 1) return;
    }
    // ---- epilogue: C[row = (e & 3) + 8 (e >> 2) + 4 h][col = r]; Y[t][n]: 4 consecutive n per float4
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const int t = t0 + 32 * jt + r;
        if (t >= p.T || (exp & 64)) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nw + 8 * g + 4 * h;
            const long o = (long) t * jb.ldy + n;
            if (role == 2) {                                    // silu(gate) * up, the arithmetic of the separate launches (silu_gate epilogue below)
                const f32x4 gg = xlds[(((wave & 3) * NT + jt) * 4 + g) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[jt][4 * g + e] *= gg[e] / (1.0f + expf(-gg[e]));
            }
            if (n + 3 < jb.N) {
                float4 v = {acc[jt][4 * g], acc[jt][4 * g + 1], acc[jt][4 * g + 2], acc[jt][4 * g + 3]};
                if (jb.bias)  { const float4 bb = ld_g((const float4 *) (jb.bias + n)); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
                if (jb.resid) { const float4 rr = ld_g((const float4 *) (jb.resid + o)); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
                if (jb.silu_gate) {
                    const float4 gg = ld_g((const float4 *) (jb.silu_gate + o));
                    v.x *= gg.x / (1.0f + expf(-gg.x)); v.y *= gg.y / (1.0f + expf(-gg.y)); v.z *= gg.z / (1.0f + expf(-gg.z)); v.w *= gg.w / (1.0f + expf(-gg.w));
                }
                if (jb.Yh) *(PM_G half4v *) (jb.Yh + o) = half4v{(_Float16) v.x, (_Float16) v.y, (_Float16) v.z, (_Float16) v.w};
                else *(PM_G f32x4 *) (jb.Y + o) = f32x4{v.x, v.y, v.z, v.w};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < jb.N) {
                    float v = acc[jt][4 * g + e];
                    if (jb.bias) v += jb.bias[n + e];
                    if (jb.resid) v += jb.resid[o + e];
                    if (jb.silu_gate) { const float gg = jb.silu_gate[o + e]; v *= gg / (1.0f + expf(-gg)); }
                    if (jb.Yh) jb.Yh[o + e] = (_Float16) v; else jb.Y[o + e] = v;
                }
            }
        }
    }
}

// block id -> (row tile, token tile): XCD x = id % 8 takes the row tiles x, x + 8, ..; its consecutive slots sweep the token tiles of one row tile, so the
// workgroups that run together on an XCD share a few row tiles of weights and the activation k-slices in its L2 (mmq_big.hip's mapping)
template <int TA, int TB, int NT, int EXP = 0>
__global__ __launch_bounds__(PF_NTHR) void gemm_pf_kernel(PfP p) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t pf_smem[];
    const int id = (int) blockIdx.x, x = id & 7;
    // slots [0, full) of an XCD run whole tiles; the slots behind them are split S ways along K, the slices of a tile neighbours on one XCD (the reducer reads
    // their slabs out of its own L2). full = 0: every tile split (S = 1: none)
    const int q = id >> 3;
    const int S = q < p.full ? 1 : p.splitk;
    const int ks = q < p.full ? 0 : (q - p.full) % p.splitk, slot = q < p.full ? q : p.full + (q - p.full) / p.splitk;
    const int per_x = (p.nt_n + 7 - x) >> 3;
    if (slot >= per_x * p.nt_t) return;
    const int tile_n = x + 8 * (slot / p.nt_t), tile_t = slot % p.nt_t;
    PfJob jb = p.job[0];                                       // (selected field by field with static indices: a run-time index would put the table into scratch)
    int n0, role = 0;
    if (p.pair) {                                              // 128-row tiles of job 0 (waves 0-3) and job 1 (waves 4-7)
        role = 1 + __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 8));
        if (role == 2) jb = p.job[1];
        n0 = tile_n * 128;
    } else {
#pragma unroll
        for (int q = 1; q < PF_MAX_JOBS; ++q) if (q < p.njobs && tile_n >= p.job[q].tile0) jb = p.job[q];
        n0 = (tile_n - jb.tile0) * 256;
    }
    const int t0 = tile_t * 32 * NT;
    const int tile_id = (slot - p.full) * 8 + x;                 // (split tiles only: slab and counter index)
    if (TA == TB || jb.type == TA) pf_body<TA, NT, EXP>(p, jb, n0, t0, pf_smem, role, ks, S, tile_id);
    else pf_body<TB, NT, EXP>(p, jb, n0, t0, pf_smem, role, ks, S, tile_id);
}

template <int TA, int TB, int NT> constexpr size_t pf_lds_bytes() { return (size_t) 3 * NT * 32 * 128 + (size_t) 8 * (pf_wave_bytes<TA>() > pf_wave_bytes<TB>() ? pf_wave_bytes<TA>() : pf_wave_bytes<TB>()); }

void pf_allow_lds(const void * kern, size_t lds) {
    struct E { const void * k; int dev; };
    static E done[64]; static int n_done = 0;
    const int dev = pm_cur_dev();
    for (int i = 0; i < n_done && i < 64; ++i) if (done[i].k == kern && done[i].dev == dev) return;
    (void) hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (n_done < 64) { done[n_done].k = kern; done[n_done].dev = dev; ++n_done; }
}

// split-K workspace of a (device, stream): 64 KiB of tile counters (zero between launches: the reducer resets its own) + the partial-tile slabs
struct PfWs { hipStream_t st; int dev; uint8_t * p; size_t bytes; uint64_t use; };
PfWs g_pf_ws[16] = {};
uint64_t g_pf_tick = 0;
std::mutex g_pf_mu;
uint8_t * pf_workspace(hipStream_t st, size_t need) {
    std::lock_guard<std::mutex> lk(g_pf_mu);
    const int dev = pm_cur_dev();
    PfWs * e = nullptr, * lru = &g_pf_ws[0];
    for (PfWs & c : g_pf_ws) {
        if (c.p && c.st == st && c.dev == dev) { e = &c; break; }
        if (!c.p) lru = &c; else if (lru->p && c.use < lru->use) lru = &c;
    }
    if (e && e->bytes >= need) { e->use = ++g_pf_tick; return e->p; }
    if (!e) e = lru;
    if (e->p) { (void) hipDeviceSynchronize(); (void) hipFree(e->p); e->p = nullptr; e->bytes = 0; }
    need = (need + ((size_t) 32 << 20)) & ~(((size_t) 16 << 20) - 1);     // (grow in steps: a later, larger batch does not reallocate at once)
    if (hipMalloc((void **) &e->p, need) != hipSuccess) { e->p = nullptr; return nullptr; }
    (void) hipMemsetAsync(e->p, 0, 65536, st);
    e->st = st; e->dev = dev; e->bytes = need; e->use = ++g_pf_tick;
    return e->p;
}

} // namespace

bool pm_gemm_pf_enabled() {
    static const int force = [] { const char * e = getenv("PM355_GEMM_KERNEL"); return e ? atoi(e) : 0; }();
    return force == 0 || force == 3;
}

// 0 when pm_launch_gemm_pf serves (type, K, N, T)
int pm_gemm_pf_check(int type, int K, int N, int T) {
    if (type != PM_Q4_K && type != PM_Q5_K && type != PM_Q6_K && type != PM_Q8_0) return -1;
    if ((type == PM_Q8_0 ? K % 128 : K % 256) || K < 512 || N < 1 || N % 4 || T < 1) return -2;
    if (type == PM_Q6_K && PM_Q6K_SCD) return -2;               // (the d's travel as 16-byte pieces = 8 super-blocks of the separate-stream row tail)
    if ((size_t) T * (size_t) K * 2 >= ((size_t) 1 << 32) || (size_t) N * pm_weight_row_stride(type, K) >= ((size_t) 1 << 32)) return -2;   // 32-bit lane offsets
    return 0;
}

// The launch geometry of `tiles` row tiles (256 rows; pair launches: 128 rows of each matrix) x T tokens over K on a device of `cus` CUs in 8 XCDs - host
// arithmetic only (tests/test_gemm_plan.py plans the layer shapes without a device). nt: 32-token sub-tiles per workgroup tile (2 / 4 / 8 = 64 / 128 / 256
// tokens: a dequantized operand serves nt MFMAs; smaller tiles only where the batch has no more tokens). Few tiles - wo / ffn_down / wq | wk | wv at the
// reference's default n_ubatch 512 (common/common.h:178) are 64-80 tiles for 256 CUs - are split along K: S slices per tile, the slice count that minimises
// rounds of workgroups x (super-blocks per slice + ~3 for prologue, slab traffic and the reducer's pass). A launch of R full rounds + a short tail (wq | wk | wv
// at 2048 tokens: 320 tiles = 256 + 64; Qwen2.5-72B's ffn_gate | ffn_up: 1848 = 7 x 256 + 56) runs whole tiles in the full rounds and splits ONLY the tail's
// tiles - as many slices as fill the idle CUs of the last round (`full` whole-tile slots per XCD come first in the grid). Splitting every tile instead sends
// every partial tile through memory: 3 x 320 slabs of 256 KB for the qkv launch, 501 us = 685 TFLOP/s against 406 us = 846.
void pm_gemm_pf_plan(int tiles, int K, int T, int cus, int force_nt, int force_s, int allow_mixed, pm_gemm_pf_plan_t * out) {
    int nt = T <= 64 ? 2 : T <= 128 ? 4 : 8;
    if (force_nt == 2 || force_nt == 4 || force_nt == 8) nt = force_nt;
    const int nt_t = (T + 32 * nt - 1) / (32 * nt);
    const long wgs = (long) tiles * nt_t;
    const int nb = (K + 255) / 256;
    if (cus < 8) cus = 8;
    int S = 1, full = 0;
    const int minL = (tiles / 8) * nt_t, maxL = ((tiles + 7) / 8) * nt_t, cpx = cus / 8;      // per-XCD slot counts (XCD x owns the row tiles x, x + 8, ..)
    double best = (double) ((wgs + cus - 1) / cus) * nb;
    for (int q = 2; q <= 8; ++q) {
        if (nb / q < 4 || wgs * q > 16384) break;
        const double c = (double) ((wgs * q + cus - 1) / cus) * ((double) nb / q + 3.0);
        if (c < best * 0.97) { best = c; S = q; }
    }
    const int F = (minL / cpx) * cpx, rem = maxL - F;
    if (allow_mixed && !(force_s >= 1 && force_s <= 8) && F > 0 && rem > 0 && 2 * rem <= cpx) {
        int q = cpx / rem; if (q > 8) q = 8;
        while (q >= 2 && nb / q < 4) --q;
        if (q >= 2) {
            const double c = (double) (F / cpx) * nb + ((double) nb / q + 3.0);
            if (c < best * 0.97) { best = c; S = q; full = F; }
        }
    }
    if (force_s >= 1 && force_s <= 8 && nb / force_s >= 1) { S = force_s; full = 0; }
    out->nt = nt; out->nt_t = nt_t; out->splitk = S; out->full = full; out->grid = 8 * (full + (maxL - full) * S); out->cost = best;
}

// One launch over njobs <= 4 matrices that share the F16 activations xh [T][K]. Job j: Y_j[t][n] (f32, token stride ldy_j) or Yh_j (F16) =
// W_j . x (+bias)(+resid)(x silu(gate)). 0, or -1 type / -2 shape
int pm_launch_gemm_pf(const pm_gemm_pf_job * jobs, int njobs, const void * xh, int K, int T, hipStream_t st) { return pm_launch_gemm_pf_ex(jobs, njobs, xh, K, T, 0, st); }

// pair != 0: jobs[0] = ffn_gate, jobs[1] = ffn_up (same type, same N; jobs[0]'s outputs are ignored): jobs[1]'s output = silu(W0 . x) * (W1 . x) (+ its own bias / resid
// before the product are NOT applied to the gate). One launch, the gate result stays on the chip.
int pm_launch_gemm_pf_ex(const pm_gemm_pf_job * jobs, int njobs, const void * xh, int K, int T, int pair, hipStream_t st) {
    if (njobs < 1 || njobs > PF_MAX_JOBS || !jobs || !xh) return -2;
    if (pair && (njobs != 2 || jobs[0].type != jobs[1].type || jobs[0].N != jobs[1].N || jobs[0].bias || jobs[0].resid || jobs[1].silu_gate)) return -2;
    PfP p = {};
    p.pair = pair ? 1 : 0;
    int tiles = 0, ta = jobs[0].type, tb = jobs[0].type;
    for (int j = 0; j < njobs; ++j) {
        const int rc = pm_gemm_pf_check(jobs[j].type, K, jobs[j].N, T);
        if (rc) return rc;
        if (jobs[j].type != ta) { if (tb != ta && jobs[j].type != tb) return -1; tb = jobs[j].type; }
        PfJob & o = p.job[j];
        o.W = (const uint8_t *) jobs[j].W; o.Y = jobs[j].Y; o.Yh = (_Float16 *) jobs[j].Yh; o.bias = jobs[j].bias; o.resid = jobs[j].resid; o.silu_gate = jobs[j].silu_gate;
        o.row_stride = (long) pm_weight_row_stride(jobs[j].type, K); o.ldy = jobs[j].ldy ? jobs[j].ldy : jobs[j].N; o.type = jobs[j].type; o.N = jobs[j].N; o.tile0 = tiles;
        tiles += (jobs[j].N + 255) / 256;
    }
    if (pair) tiles = (jobs[0].N + 127) / 128;
    if (ta > tb) { const int t = ta; ta = tb; tb = t; }       // (Q4_K, Q6_K) in that order
    static const int grp = [] { const char * e = getenv("PM355_GEMM_PF_GROUP"); return e ? atoi(e) : 0; }();
    p.grp = grp;
    p.njobs = njobs; p.nt_n = tiles; p.Xh = (const _Float16 *) xh; p.K = K; p.T = T;
    static const int force_nt = [] { const char * e = getenv("PM355_GEMM_PF_NT"); return e ? atoi(e) : 0; }();
    static const int force_s = [] { const char * e = getenv("PM355_GEMM_PF_SPLITK"); return e ? atoi(e) : 0; }();
    static const bool no_mixed = [] { const char * e = getenv("PM355_GEMM_PF_MIXED"); return e && e[0] == '0'; }();
    const int cus = pm_device_cus();
    // 256-token tiles (a dequantized operand serves 8 MFMAs; 128-token tiles only where the batch has no more tokens). Few tiles - wo / ffn_down / wq | wk | wv
    // at the reference's default n_ubatch 512 (common/common.h:178) are 64-80 tiles for 256 CUs - are split along K: S slices per tile, the slice count that
    // minimises  rounds of workgroups x (super-blocks per slice + ~3 for prologue, slab traffic and the reducer's pass)
    pm_gemm_pf_plan_t pl;
    pm_gemm_pf_plan(tiles, K, T, cus, force_nt, force_s, no_mixed ? 0 : 1, &pl);
    const int nt = pl.nt, S = pl.splitk, full = pl.full, maxL = ((tiles + 7) / 8) * pl.nt_t;
    p.nt_t = pl.nt_t;
    p.splitk = S; p.full = full;
    const int split_slots = maxL - full;                       // per XCD
    if (S > 1) {
        const size_t slab = (size_t) 8 * nt * 4 * 64 * 16, need = 65536 + (size_t) 8 * split_slots * S * slab;
        if ((size_t) 8 * split_slots > 16384) return -2;
        uint8_t * w = pf_workspace(st, need);
        if (!w) return -3;
        p.cnt = (unsigned *) w; p.ws = (float *) (w + 65536);
    }
    const int slots = full + split_slots * S;
    auto go = [&](auto kern, size_t lds) {
        pf_allow_lds((const void *) kern, lds);
        hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(PF_NTHR), lds, st, p);
    };
#if PM_GEMM_ABLATE
    static const int exp_sw = [] { const char * e = getenv("PM355_GEMM_EXP"); return e ? atoi(e) : 0; }();
    if (exp_sw && ta == PM_Q4_K && tb == PM_Q4_K && nt == 8) {
        static unsigned long long * tr = nullptr;
        if (!tr) (void) hipMalloc((void **) &tr, 42 * 8);
        p.trace = tr;
        switch (exp_sw) {
#define PF_AB(E) case E: go(gemm_pf_kernel<PM_Q4_K, PM_Q4_K, 8, E>, pf_lds_bytes<PM_Q4_K, PM_Q4_K, 8>()); return 0;
            PF_AB(3) PF_AB(4) PF_AB(8) PF_AB(16) PF_AB(24) PF_AB(27)
#define PF_TR(E) case 128 + E: go(gemm_pf_kernel<PM_Q4_K, PM_Q4_K, 8, 128 + E>, pf_lds_bytes<PM_Q4_K, PM_Q4_K, 8>()); break;
            PF_TR(0) PF_TR(3) PF_TR(8) PF_TR(16) PF_TR(24) PF_TR(27)
#undef PF_TR
            default: return -2;
        }
        unsigned long long hst[42];
        (void) hipStreamSynchronize(st);
        (void) hipMemcpy(hst, tr, sizeof hst, hipMemcpyDeviceToHost);
        for (int g = 0; g < 2; ++g) {
            fprintf(stderr, "pf trace exp %d wave %d:", exp_sw & 127, 4 * g);
            for (int k = 1; k < 17; ++k) fprintf(stderr, " %s%lld", k % 4 == 1 ? "| M " : k % 4 == 2 ? "b " : k % 4 == 3 ? "C " : "b ", (long long) (hst[g * 17 + k] - hst[g * 17 + k - 1]));
            fprintf(stderr, "  (A starts %lld after B)\n", (long long) (hst[0] - hst[17]));
        }
        return 0;
    }
#endif
#define PF_GO(A, B) (nt == 8 ? go(gemm_pf_kernel<A, B, 8>, pf_lds_bytes<A, B, 8>()) : nt == 4 ? go(gemm_pf_kernel<A, B, 4>, pf_lds_bytes<A, B, 4>()) : go(gemm_pf_kernel<A, B, 2>, pf_lds_bytes<A, B, 2>()))
    if (ta == tb) { if (ta == PM_Q4_K) PF_GO(PM_Q4_K, PM_Q4_K); else if (ta == PM_Q5_K) PF_GO(PM_Q5_K, PM_Q5_K); else if (ta == PM_Q8_0) PF_GO(PM_Q8_0, PM_Q8_0); else PF_GO(PM_Q6_K, PM_Q6_K); }
    else if (ta == PM_Q8_0 || tb == PM_Q8_0) return -1;      // (Q8_0 only as a launch of its own: ffn_down of the files whose n_ff is no multiple of 256)
    else if (ta == PM_Q4_K && tb == PM_Q5_K) PF_GO(PM_Q4_K, PM_Q5_K);
    else if (ta == PM_Q4_K && tb == PM_Q6_K) PF_GO(PM_Q4_K, PM_Q6_K);
    else PF_GO(PM_Q5_K, PM_Q6_K);
#undef PF_GO
    return 0;
}
