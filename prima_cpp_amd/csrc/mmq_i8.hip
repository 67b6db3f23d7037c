// mmq_i8.hip — small-batch (up to 32 tokens per pass, 64 per call) quantized mat-mul on the INTEGER matrix cores
// (v_mfma_i32_32x32x32_i8 / v_mfma_i32_32x32x16_i8). Q4_K, Q5_K, Q6_K weights x Q8_K activations.
//
// What it replaces: for 2..64 activation rows the reference runs ggml_compute_forward_mul_mat's vec_dot loop once per (row, token)
// (ggml/src/ggml.c:12377 -> ggml_vec_dot_q4_K_q8_K / ggml_vec_dot_q5_K_q8_K / ggml_vec_dot_q6_K_q8_K, ggml-quants.c:7713 / :8281 / :8918), its
// CUDA plug-in the integer mmq tiles (ggml-cuda/mmq.cuh:2583). The arithmetic here is the reference's own: activations quantized to Q8_K,
// exact int32 sums  sum_s scale_s * (q_w . q_a)  and  sum_s min_s * bsum_s  per 256-weight super-block, then ONE f32 multiply-add per
// super-block with d_w * d_a - only the order in which the super-block terms are added in f32 differs (K split over waves).
//
// Why a third mat-mul kernel: between the mat-vec (1 token: weight-stream bound) and the F16 MFMA GEMM (256-token tiles: a 16-token
// batch pays for 256) sits the regime of speculative decoding, parallel sequences and short prompts. The multi-column mat-vec
// (mmvq_cols.hip) re-reads its activations from LDS for every decoded weight and costs ~10 us per extra column on the ffn_gate shape
// (profiles/r02_small_batch_probe.txt); this kernel streams the weights ONCE per 32 tokens and does the products on the matrix cores.
//
// Mapping (one MFMA = 32 tokens x 32 weight rows x 32 k for Q4_K / Q5_K = one 32-weight sub-block; x 16 k for Q6_K = one 16-weight group):
//   A operand = activations: lane (t = lane % 32, g = lane / 32) holds 16 (8) consecutive int8 of token t
//   B operand = weights    : lane (r = lane % 32, g)            holds the same k of weight row r
//   result                 : lane (r, g) holds, for ITS OWN row r, tokens 8 (v / 4) + 4 g + v % 4, v = 0..15
// so the per-row sub-block scale / min / d of a K-quant are lane-local scalars when the i32 tile is scaled (one v_mad_i32_i24 per result;
// the instantiation keeps 4 / 8 / 16 result registers for <= 8 / 16 / 32 tokens). An MFMA never crosses a scale group. The min / -32
// terms are one more MFMA per super-block in F16 (v_mfma_f32_32x32x16_f16: 6-bit mins or int8 scales x 16-value activation sums <= 2032,
// all exact in F16, f32 accumulation of integers < 2^24 exact).
//
// Data movement: weights keep the mat-vec's HBM layout (repack.hip; Q5_K native). A wave owns 32 rows x 512 k per step: it loads them
// with FULL 128-byte lines per row and stream (8 lanes per line, non-temporal; the per-row headers / scales cached), parks them in its
// private LDS tile and reads them back in MFMA operand order - no barrier, a wave's LDS accesses execute in order. The next step's loads
// are in flight while the current one is computed (VMEM returns in order: the activation loads a step waits for are always issued BEFORE
// the weight prefetch that overlaps it; scheduling barriers keep the compiler from sinking them). Activations come from L2 as ONE coalesced
// kilobyte per 32-value sub-block out of a table in A-operand order; two more small tables hold the activation scales (transposed, -> LDS) and the
// F16 group sums in A-operand order - all three written by the Q8_K quantizers as a second output (pm_q8k_tables) or by a prologue launch.
// [Round 4: the operand-ordered table replaced per-token loads (one cache line per token and instruction, 16 B of it used: at 8 tokens the
//  activation loads cost 11 of the 67 us of an ffn_down launch, at 32 tokens more); Q6_K's 16-k products take their halves from the 16-byte
//  loads through v_permlane32_swap. The loads are buffer loads whose token slots >= T carry an out-of-range offset - unconditional instructions
//  (a branch around them made the compiler's vmcnt bookkeeping wait for the NEXT weight tile before THIS step's products).]
// Work split: workgroup = 8 waves, a balanced slice of rows = up to 8 groups of 32; K is split over the waves that share a row group -
// interleaved pair by pair, so that those waves read adjacent 128-byte pieces of a row at the same time (ffn_down Q6_K: 76 -> 70 us) - and
// their f32 partial tiles are added in a fixed order through LDS (bitwise reproducible). Multi-job form (MJ): up to 3 matrices of one
// type and K that share the activations form one virtual row space (wq | wk | wv, ffn_gate | ffn_up): one launch, one fill / drain.
// Measurements, ablation and what did not help: profiles/r02_small_batch_probe.txt, DESIGN.md section 3.
#include "pm355_device.h"
#include "pm355_kernels.h"
#include <mutex>

namespace {

typedef int      i32x16 __attribute__((ext_vector_type(16)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));

constexpr int BLOCK = 512, NWAVE = 8;
constexpr int PITCH = 144;            // LDS bytes per row of a 128-byte stream tile: 16 consecutive rows -> 16 distinct 16-byte bank slots
constexpr int PITCH_H = 48;           // same for the 2 x 16-byte per-row header / scale tile

struct MmqP {
    const uint8_t * W; long row_stride; int N, K, T, rgb_log2;     // 1 << rgb_log2 row groups per batch (1, 2, 4, 8)
    int t_off;                                                     // first table slot of this pass (16-token passes of a 32-slot table)
    const uint8_t * xq; long xq_stride;                            // row-SoA Q8_K activations [T]
    const uint8_t * bsT; const float * dT;                         // prologue tables: [nsb][64 lanes][8 f16], [nsb][32]
    const uint8_t * qT;                                            // the activations in A-operand order: [nsb][8 sub-blocks][64 lanes][16 B]
    float * y; long y_stride; const float * bias; const float * resid;
    // multi-job launches (MJ kernels): up to 3 matrices of one type and K that share the activations (wq | wk | wv, ffn_gate | ffn_up) form
    // ONE virtual row space [0, N): job j owns rows [start[j], start[j] + nj[j]); y / bias per job, output token stride nj[j]
    const uint8_t * Wj[3]; float * yj[3]; const float * bj[3]; int start[3], nj[3], njobs;
    // wq | wk | wv epilogue (e_on; round 6): RoPE (NORM pairing: rows 2 i, 2 i + 1 of a head) with the per-token cos / sin tables of pm_launch_rope_table, q ->
    // F16-rounded f32 into its y, k -> the F16 K-cache row of cell pos + t, v -> the transposed F16 V cache - what rope_kv_store_kernel (layer_ops.hip) does in a
    // launch of its own, the same expressions (ggml_compute_forward_rope_f32 ggml.c:14224-14237; llm_build_kv_store src/llama.cpp:9688-9716).
    // e_role[j]: 1 q, 2 k, 3 v (single-job launches: e_role[0]). Workgroup row slices start on even rows then.
    int e_on, e_role[3]; const float * e_tab; const int32_t * e_pos, * e_seq; long e_seq_stride; uint16_t * e_kc, * e_vc; int e_dh, e_nctx, e_nrot, e_kvdim;
};

__device__ __forceinline__ long pk(uint32_t a, uint32_t b) { return (long) (((uint64_t) b << 32) | a); }
__device__ __forceinline__ u32x4 ld_c16(const void * p) { return *(const PM_G u32x4 *) p; }
__device__ __forceinline__ u32x2 ld_c8(const void * p)  { return *(const PM_G u32x2 *) p; }
__device__ __forceinline__ i32x16 mfma_i8(long a, long b, i32x16 c) { return __builtin_amdgcn_mfma_i32_32x32x16_i8(a, b, c, 0, 0, 0); }
typedef int i32x4 __attribute__((ext_vector_type(4)));
// gfx950's 2 x K form: lane (i, g) holds 16 consecutive int8 (k = 16 g .. 16 g + 15)
__device__ __forceinline__ i32x16 mfma_i8x32(u32x4 a, u32x4 b, i32x16 c) {
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}

// where the loader lanes of a wave point: lane (rr = lane / 8, c = lane % 8) fetches 16-byte chunk c of rows rr, rr + 8, rr + 16, rr + 24
template <bool MJ> struct Rows;
// lm: all ones, or 0 for a look-ahead request past the wave's last step (the step code has no branches - the compiler's vmcnt bookkeeping stays exact - so
// the request is issued anyway): every lane then asks for the same few addresses at the start of a row instead of re-loading the last tile (round 6: that
// re-load was a third of wo's L2 -> CU traffic - two real tiles per wave and one for nothing)
template <> struct Rows<false> {          // one matrix: wave-uniform base + 32-bit lane offsets
    const uint8_t * base; uint32_t stride, off_d, lm; int lim /*last valid row of the group, relative*/, rr, rh;
    __device__ __forceinline__ const uint8_t * at(int n, uint32_t x) const { return base + (((uint32_t) min(rr + 8 * n, lim) * stride + x) & lm); }
    __device__ __forceinline__ const uint8_t * at_h(uint32_t x) const { return base + (((uint32_t) min(rh, lim) * stride + x) & lm); }
    __device__ __forceinline__ const uint8_t * at_d(uint32_t x) const { return base + ((off_d + x) & lm); }
    __device__ __forceinline__ const uint8_t * at_row(int row, uint32_t x) const { return base + (((uint32_t) min(row, lim) * stride + x) & lm); }   // any row of the group
};
template <> struct Rows<true> {           // several matrices: a lane's rows may sit in different allocations -> 64-bit row pointers per lane
    const uint8_t * rp[4], * rph, * rpd; uint32_t lm;
    __device__ __forceinline__ const uint8_t * at(int n, uint32_t x) const { return rp[n] + (x & lm); }
    __device__ __forceinline__ const uint8_t * at_h(uint32_t x) const { return rph + (x & lm); }
    __device__ __forceinline__ const uint8_t * at_d(uint32_t x) const { return rpd + (x & lm); }
    __device__ __forceinline__ const uint8_t * at_row(int, uint32_t) const { return nullptr; }     // (native-layout types are single-job only)
};

// Where a lane's activation operand comes from. Token slots >= T must not cost memory traffic (the texture path is paid per ACTIVE lane,
// tools/small_batch_probe.py ablation) - but an `if (lane is a token)` around the loads becomes a branch over them, and the compiler's
// s_waitcnt vmcnt bookkeeping then assumes the loads behind it may not have been issued: the wait for THIS step's activations also waited for
// the NEXT step's whole weight tile (ISA: vmcnt(14) where 31 loads were in flight), i.e. the prefetch overlapped half a step. Buffer loads
// instead: unconditional instructions, the lanes without a token carry an out-of-range offset, return 0 and touch no memory.
struct ActSrc {
    __amdgpu_buffer_rsrc_t rx, rb; uint32_t xo, bo;        // xo / bo: this lane's byte offset into the operand-ordered activations / the group-sum table (or out of range)
    // sub-block s of super-block sb: lane (t, g) gets the 16 bytes k = 32 s + 16 g .. + 15 of token t (one coalesced kilobyte per wave and 32 tokens)
    __device__ __forceinline__ u32x4 ld_sub(int sb, int s) const { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int) xo, sb * 8192 + s * 1024, 0)); }
    __device__ __forceinline__ f16x8 ld_bs(int sb) const { return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rb, (int) bo, sb * 1024, 0)); }
    // Q8_0 (32-value blocks): block blk of the operand-ordered values, and the four activation scales of tokens 8 q + 4 g .. + 3 (bo = 16 g)
    __device__ __forceinline__ u32x4 ld_blk(int blk) const { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int) xo, blk * 1024, 0)); }
    __device__ __forceinline__ f32x4 ld_d4(int blk, int q) const { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int) bo, blk * 128 + 32 * q, 0)); }
};

template <int TYPE> struct MT;

// ---------------------------------------------------------------- Q4_K: row = qa[U][16] | qb[U][16] | hdr[nb][16] ------------
template <> struct MT<PM_Q4_K> {
    static constexpr int QA = 0, QB = 32 * PITCH, HD = 64 * PITCH, WAVE_LDS = 64 * PITCH + 32 * PITCH_H;
    struct B { u32x4 qa[4], qb[4], h; };
    struct A { u32x4 q[8]; f16x8 bs; };
    // part n of a step's tile (4 parts: rows rr + 8 n of both nibble streams; part 0 also the headers)
    template <class RW>
    static __device__ __forceinline__ void issue_b_part(B & b, const RW & rw, int nb, int pr, int lane, int n) {
        const uint32_t u = (uint32_t) min(8 * pr + (lane & 7), 4 * nb - 1), sb = (uint32_t) min(2 * pr + (lane & 1), nb - 1);
        b.qa[n] = ld_nt16(rw.at(n, u * 16u));
        b.qb[n] = ld_nt16(rw.at(n, (uint32_t) nb * 64u + u * 16u));
        if (n == 0) b.h = ld_c16(rw.at_h((uint32_t) nb * 128u + sb * 16u));   // cached: 4 steps share the line (as nt loads they re-fetched it from HBM: +29 % traffic)
    }
    static __device__ __forceinline__ void stash(const B & b, uint8_t * L, int lane) {
        const int rr = lane >> 3, c = lane & 7;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            *(u32x4 *) (L + QA + (rr + 8 * n) * PITCH + c * 16) = b.qa[n];
            *(u32x4 *) (L + QB + (rr + 8 * n) * PITCH + c * 16) = b.qb[n];
        }
        *(u32x4 *) (L + HD + (lane >> 1) * PITCH_H + (lane & 1) * 16) = b.h;
    }
    // sub-block s (32 weights) = the 16 bytes at 32 s + 16 g of the super-block
    template <int NV>
    static __device__ __forceinline__ void issue_a(A & a, const ActSrc & x, int sb) {
#pragma unroll
        for (int s = 0; s < 8; ++s) a.q[s] = x.ld_sub(sb, s);
        a.bs = x.ld_bs(sb);
    }
    template <int NV, class F>
    static __device__ __forceinline__ void compute(const A & a, const uint8_t * L, int sbi, int r, int g, const float * yd_lds, int /*t_off*/, f32x16 & out, F && between) {
        const u32x4 hd = *(const u32x4 *) (L + HD + r * PITCH_H + sbi * 16);
        const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        i32x16 isum = zero;
        // the 8 six-bit scales / mins of the super-block, four per dword (get_scale_min_k4, ggml-quants.c:1898-1906, on whole dwords)
        const uint32_t sc4[2] = {hd[1] & 0x3f3f3f3fu, (hd[3] & 0x0f0f0f0fu) | ((hd[1] >> 2) & 0x30303030u)};
        const uint32_t mn4[2] = {hd[2] & 0x3f3f3f3fu, ((hd[3] >> 4) & 0x0f0f0f0fu) | ((hd[2] >> 2) & 0x30303030u)};
        f16x8 bm;
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) bm[sb] = (_Float16) (float) ((mn4[sb >> 2] >> (8 * (sb & 3))) & 0xFFu);
        // (round 6, measured and rejected for 17..32 tokens, where the 16 v_mad_i32_i24 per sub-block are what the wave does: the scale INSIDE the weight operand -
        //  sc = 8 hi + lo, q * lo and q * hi are bytes <= 105, four v_pk_mul_lo_u16 per operand, two MFMA chains that accumulate across the sub-blocks,
        //  sum = lo-chain + 8 * hi-chain: the same integers, 8 instead of 16 vector instructions and 2 instead of 1 MFMA per sub-block. 52-100 B of spills,
        //  ffn_gate 43.6 -> 49.8 us, 202.5 -> 229.8 us per 70B layer at 32 tokens: profiles/r06_small_batch.txt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            between(j);
            const u32x4 w = *(const u32x4 *) (L + (g ? QB : QA) + r * PITCH + (4 * sbi + j) * 16);
            const int sc0 = (int) ((sc4[(2 * j) >> 2] >> (8 * ((2 * j) & 3))) & 0xFFu), sc1 = (int) ((sc4[(2 * j + 1) >> 2] >> (8 * ((2 * j + 1) & 3))) & 0xFFu);
            i32x16 acc = mfma_i8x32(a.q[2 * j], w & 0x0F0F0F0Fu, zero);                    // one 32-k MFMA = one 32-weight sub-block
#pragma unroll
            for (int v = 0; v < NV; ++v) { isum[v] = __mul24(sc0, acc[v]) + isum[v]; asm volatile("" : "+v"(isum[v])); }   // (keeps it ONE v_mad_i32_i24:
            acc = mfma_i8x32(a.q[2 * j + 1], (w >> 4) & 0x0F0F0F0Fu, zero);                //  the re-associated mul, mul, add3 form is 1.5 instructions per term)
#pragma unroll
            for (int v = 0; v < NV; ++v) { isum[v] = __mul24(sc1, acc[v]) + isum[v]; asm volatile("" : "+v"(isum[v])); }
        }
        const f32x16 fz = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const f32x16 ms = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.bs, bm, fz, 0, 0, 0);   // sum_s min_s * (bsum[2s] + bsum[2s+1]), exact
        const float d = h2f((uint16_t) (hd[0] & 0xFFFF)), dmin = h2f((uint16_t) (hd[0] >> 16));
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) {
            const f32x4 yd = *(const f32x4 *) (yd_lds + 8 * q + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = 4 * q + i;
                out[v] = fmaf(yd[i] * d, (float) isum[v], fmaf(-(yd[i] * dmin), ms[v], out[v]));
            }
        }
    }
};

// ---------------------------------------------------------------- Q5_K: native 176-byte blocks (d, dmin | scales[12] | qh[32] | qs[128]) ------
// A step's 2 super-blocks are 352 contiguous bytes per row = 22 chunks of 16: lane l fetches chunks l, l + 64, ... (11 loads) of the 32 x 22
// chunk tile; LDS row pitch 368 B (23 x 16: 16 consecutive rows -> 16 distinct bank slots).
template <> struct MT<PM_Q5_K> {
    static constexpr int RP = 368, WAVE_LDS = 32 * RP;
    struct B { u32x4 c[11]; };
    struct A { u32x4 q[8]; f16x8 bs; };
    template <class RW>
    static __device__ __forceinline__ void issue_b_part(B & b, const RW & rw, int nb, int pr, int lane, int n) {
#pragma unroll
        for (int i = 3 * n; i < (n == 3 ? 11 : 3 * n + 3); ++i) {
            const int ci = lane + 64 * i, row = ci / 22, c = ci - row * 22;
            const int sb = min(2 * pr + (c >= 11), nb - 1);
            b.c[i] = ld_nt16(rw.at_row(row, (uint32_t) sb * 176u + (uint32_t) (c >= 11 ? c - 11 : c) * 16u));
        }
    }
    static __device__ __forceinline__ void stash(const B & b, uint8_t * L, int lane) {
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const int ci = lane + 64 * i, row = ci / 22, c = ci - row * 22;
            *(u32x4 *) (L + row * RP + c * 16) = b.c[i];
        }
    }
    template <int NV>
    static __device__ __forceinline__ void issue_a(A & a, const ActSrc & x, int sb) {
#pragma unroll
        for (int s = 0; s < 8; ++s) a.q[s] = x.ld_sub(sb, s);
        a.bs = x.ld_bs(sb);
    }
    template <int NV, class F>
    static __device__ __forceinline__ void compute(const A & a, const uint8_t * L, int sbi, int r, int g, const float * yd_lds, int /*t_off*/, f32x16 & out, F && between) {
        const uint8_t * blk = L + r * RP + sbi * 176;
        const u32x4 hd = *(const u32x4 *) blk;
        const u32x4 qh = *(const u32x4 *) (blk + 16 + 16 * g);       // bit s of byte l: fifth bit of weight l of sub-block s (l = 16 g + i)
        const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        i32x16 isum = zero;
        const uint32_t sc4[2] = {hd[1] & 0x3f3f3f3fu, (hd[3] & 0x0f0f0f0fu) | ((hd[1] >> 2) & 0x30303030u)};
        const uint32_t mn4[2] = {hd[2] & 0x3f3f3f3fu, ((hd[3] >> 4) & 0x0f0f0f0fu) | ((hd[2] >> 2) & 0x30303030u)};
        f16x8 bm;
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) bm[sb] = (_Float16) (float) ((mn4[sb >> 2] >> (8 * (sb & 3))) & 0xFFu);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            between(j);
            const u32x4 w = *(const u32x4 *) (blk + 48 + 32 * j + 16 * g);
            const int sc0 = (int) ((sc4[(2 * j) >> 2] >> (8 * ((2 * j) & 3))) & 0xFFu), sc1 = (int) ((sc4[(2 * j + 1) >> 2] >> (8 * ((2 * j + 1) & 3))) & 0xFFu);
            i32x16 acc = mfma_i8x32(a.q[2 * j], (w & 0x0F0F0F0Fu) | (((qh >> (2 * j)) & 0x01010101u) << 4), zero);
#pragma unroll
            for (int v = 0; v < NV; ++v) { isum[v] = __mul24(sc0, acc[v]) + isum[v]; asm volatile("" : "+v"(isum[v])); }
            acc = mfma_i8x32(a.q[2 * j + 1], ((w >> 4) & 0x0F0F0F0Fu) | (((qh >> (2 * j + 1)) & 0x01010101u) << 4), zero);
#pragma unroll
            for (int v = 0; v < NV; ++v) { isum[v] = __mul24(sc1, acc[v]) + isum[v]; asm volatile("" : "+v"(isum[v])); }
        }
        const f32x16 fz = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const f32x16 ms = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.bs, bm, fz, 0, 0, 0);
        const float d = h2f((uint16_t) (hd[0] & 0xFFFF)), dmin = h2f((uint16_t) (hd[0] >> 16));
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) {
            const f32x4 yd = *(const f32x4 *) (yd_lds + 8 * q + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = 4 * q + i;
                out[v] = fmaf(yd[i] * d, (float) isum[v], fmaf(-(yd[i] * dmin), ms[v], out[v]));
            }
        }
    }
};

// ---------------------------------------------------------------- Q6_K: row = la[U][16] | lb[U][16] | qh[U][16] | scales[nb][16] | d[nb] ------
template <> struct MT<PM_Q6_K> {
    static constexpr int LA = 0, LB = 32 * PITCH, QH = 64 * PITCH, SC = 96 * PITCH, DD = SC + 32 * PITCH_H, WAVE_LDS = DD + 128;
    struct B { u32x4 la[4], lb[4], qh[4], s; uint16_t d; };
    struct A { u32x4 q[8]; f16x8 bs; };       // as loaded: lane (t, g) holds the WHOLE 16-weight group 2 s + g of sub-block s (operand(): the halves the 16-k products need)
    template <class RW>
    static __device__ __forceinline__ void issue_b_part(B & b, const RW & rw, int nb, int pr, int lane, int n) {
        const uint32_t u = (uint32_t) min(8 * pr + (lane & 7), 4 * nb - 1), sb = (uint32_t) min(2 * pr + (lane & 1), nb - 1);
        b.la[n] = ld_nt16(rw.at(n, u * 16u));
        b.lb[n] = ld_nt16(rw.at(n, (uint32_t) nb * 64u + u * 16u));
        b.qh[n] = ld_nt16(rw.at(n, (uint32_t) nb * 128u + u * 16u));
        if (n == 0) b.s = ld_c16(rw.at_h(pm_q6k_sc_off((uint32_t) nb, sb)));   // cached, like Q4_K's header
        // (d: one 2-byte load per lane (row lane % 32, super-block 2 pr + lane / 32), issued by the kernel with its own row offset)
    }
    static __device__ __forceinline__ void stash(const B & b, uint8_t * L, int lane) {
        const int rr = lane >> 3, c = lane & 7;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            *(u32x4 *) (L + LA + (rr + 8 * n) * PITCH + c * 16) = b.la[n];
            *(u32x4 *) (L + LB + (rr + 8 * n) * PITCH + c * 16) = b.lb[n];
            *(u32x4 *) (L + QH + (rr + 8 * n) * PITCH + c * 16) = b.qh[n];
        }
        *(u32x4 *) (L + SC + (lane >> 1) * PITCH_H + (lane & 1) * 16) = b.s;
        *(uint16_t *) (L + DD + lane * 2) = b.d;                     // [c = lane / 32][r = lane % 32]
    }
    // 16-weight group G = the 8 bytes at 16 G + 8 g of the super-block
    template <int NV>
    static __device__ __forceinline__ void issue_a(A & a, const ActSrc & x, int sb) {
#pragma unroll
        for (int s = 0; s < 8; ++s) a.q[s] = x.ld_sub(sb, s);
        a.bs = x.ld_bs(sb);
    }
    // A operand of the 16-k product over group G: lane (t, g) needs bytes 8 g .. 8 g + 7 of group G. The lane pair (t, 0) / (t, 1) holds groups
    // 2 s / 2 s + 1 whole: v_permlane32_swap hands the upper half of the even group to lane (t, 1) and the lower half of the odd group to (t, 0)
    static __device__ __forceinline__ void operands(const u32x4 & v, u32x2 & even, u32x2 & odd) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x2 r = __builtin_bit_cast(u32x2, __builtin_amdgcn_permlane32_swap(v[i], v[2 + i], false, false));
            even[i] = r[0]; odd[i] = r[1];
        }
    }
    template <int NV, class F>
    static __device__ __forceinline__ void compute(const A & a, const uint8_t * L, int sbi, int r, int g, const float * yd_lds, int /*t_off*/, f32x16 & out, F && between) {
        const u32x4 s16 = *(const u32x4 *) (L + SC + r * PITCH_H + sbi * 16);
        const float d = h2f(*(const uint16_t *) (L + DD + (sbi * 32 + r) * 2));
        const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        i32x16 isum = zero;
        u32x2 aop[16];                                   // (the swaps work in place on the registers of a.q)
#pragma unroll
        for (int s = 0; s < 8; ++s) operands(a.q[s], aop[2 * s], aop[2 * s + 1]);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int v2 = 0; v2 < 2; ++v2) {
                between(2 * hh + v2);
                const int uo = (4 * sbi + 2 * hh + v2) * 16 + 8 * g;
                const u32x2 la = *(const u32x2 *) (L + LA + r * PITCH + uo), lb = *(const u32x2 *) (L + LB + r * PITCH + uo),
                            qh = *(const u32x2 *) (L + QH + r * PITCH + uo);
                u32x2 qv[4];
                qv[0] = (la & 0x0F0F0F0Fu)        | ((qh << 4) & 0x30303030u);
                qv[1] = (lb & 0x0F0F0F0Fu)        | ((qh << 2) & 0x30303030u);
                qv[2] = ((la >> 4) & 0x0F0F0F0Fu) | (qh & 0x30303030u);
                qv[3] = ((lb >> 4) & 0x0F0F0F0Fu) | ((qh >> 2) & 0x30303030u);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int G = 8 * hh + 2 * c + v2;                                   // group index inside the super-block = scale index
                    const int sc = (int) (int8_t) (s16[G >> 2] >> (8 * (G & 3)));
                    const i32x16 acc = mfma_i8(pk(aop[G][0], aop[G][1]), pk(qv[c][0], qv[c][1]), zero);
#pragma unroll
                    for (int v = 0; v < NV; ++v) { isum[v] = __mul24(sc, acc[v]) + isum[v]; if constexpr (NV < 16) asm volatile("" : "+v"(isum[v])); }   // |acc| <= 16*63*127 (NV = 16: the barrier costs 100 B of spills)
                }
            }
        // sum_G scale_G * bsum_G (the -32 offset of every weight): B slot s of lane group g = scale[2 s + g]
        f16x8 bm;
#pragma unroll
        for (int s = 0; s < 8; ++s) { const int G = 2 * s + g; bm[s] = (_Float16) (float) (int) (int8_t) (s16[G >> 2] >> (8 * (G & 3))); }
        const f32x16 fz = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const f32x16 ms = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.bs, bm, fz, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) {
            const f32x4 yd = *(const f32x4 *) (yd_lds + 8 * q + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = 4 * q + i;
                out[v] = fmaf(yd[i] * d, (float) (isum[v] - 32 * (int) ms[v]), out[v]);   // the reference's integer sum, then one rounding
            }
        }
    }
};

// ---------------------------------------------------------------- Q8_0: row = qa[nb][16] | qb[nb][16] | d[nb] (f16), nb = K / 32 ------------
// Weights and activations in 32-value blocks (ggml_vec_dot_q8_0_q8_0, ggml-quants.c:5518: sumf += sumi * d_w * d_a per block): one 32-k MFMA is one
// block, its 32 x 32 integer tile is scaled by d_w (lane-local: the lane's own row) x d_a (the token of the result register). In this kernel's
// terms a "super-block" is 128 weights = 4 blocks and a step 8 blocks - per row the same 128 + 128 + 16 bytes as a Q4_K step, so the tile loader
// and its LDS image are Q4_K's. K need not be a multiple of 256 (Qwen2.5-72B's ffn_down, K = 29568, falls back to this type: src/llama.cpp:19447):
// blocks past the row are clamped weight bytes against out-of-range (= zero) activations and scales.
template <> struct MT<PM_Q8_0> {
    static constexpr int QA = 0, QB = 32 * PITCH, HD = 64 * PITCH, DS = HD + 32 * PITCH_H, WAVE_LDS = DS + 1024;
    struct B { u32x4 qa[4], qb[4], h, ds; };
    struct A { u32x4 q[4]; };
    // (ds: the ACTIVATION scales of the step's 8 blocks, [block][32 token slots] f32 = 1 KiB per wave and step, travel with the weight tile into the wave's
    //  LDS image - held in registers per result token they cost 32 VGPRs per operand set and the 16-token form spilled)
    template <class RW>
    static __device__ __forceinline__ void issue_b_part(B & b, const RW & rw, int nb /*blocks per row*/, int pr, int lane, int n) {
        const uint32_t u = (uint32_t) min(8 * pr + (lane & 7), nb - 1);
        b.qa[n] = ld_nt16(rw.at(n, u * 16u));
        b.qb[n] = ld_nt16(rw.at(n, (uint32_t) nb * 16u + u * 16u));
        if (n == 0) b.h = ld_c16(rw.at_h((uint32_t) nb * 32u + (uint32_t) pr * 16u));   // the step's 8 block scales (row padding keeps the last piece inside the row)
    }
    static __device__ __forceinline__ void issue_ds(B & b, const ActSrc & x, int pr) { b.ds = __builtin_bit_cast(u32x4, x.ld_d4(8 * pr, 0)); }   // (bo = 16 * lane; blocks past the row: 0)
    static __device__ __forceinline__ void stash(const B & b, uint8_t * L, int lane) {
        const int rr = lane >> 3, c = lane & 7;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            *(u32x4 *) (L + QA + (rr + 8 * n) * PITCH + c * 16) = b.qa[n];
            *(u32x4 *) (L + QB + (rr + 8 * n) * PITCH + c * 16) = b.qb[n];
        }
        if ((lane & 1) == 0) *(u32x4 *) (L + HD + (lane >> 1) * PITCH_H) = b.h;
        *(u32x4 *) (L + DS + lane * 16) = b.ds;
    }
    template <int NV>
    static __device__ __forceinline__ void issue_a(A & a, const ActSrc & x, int sb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a.q[j] = x.ld_blk(4 * sb + j);
    }
    template <int NV, class F>
    static __device__ __forceinline__ void compute(const A & a, const uint8_t * L, int sbi, int r, int g, const float *, int t_off, f32x16 & out, F && between) {
        const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const u32x2 d4 = *(const u32x2 *) (L + HD + r * PITCH_H + sbi * 8);              // this row's four block scales (f16)
        const float * ds = (const float *) (L + DS) + t_off;                             // (16-token passes over a 32-slot table start at slot t_off)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            between(j);
            const u32x4 w = *(const u32x4 *) (L + (g ? QB : QA) + r * PITCH + (4 * sbi + j) * 16);
            const uint32_t hb = (d4[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            const float dw = (hb & 0x7C00u) == 0x7C00u ? 0.0f : h2f((uint16_t) hb);       // (a block past the row: whatever the padding holds, times zero)
            const i32x16 acc = mfma_i8x32(a.q[j], w, zero);
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) {
                const f32x4 yd = *(const f32x4 *) (ds + (4 * sbi + j) * 32 + 8 * q + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) out[4 * q + i] = fmaf((float) acc[4 * q + i], dw * yd[i], out[4 * q + i]);
            }
        }
    }
};

// ABL (measurement only, PM355_MMQ_ABL): 1 = no activation loads in the loop, 2 = no weight loads in the loop, 4 = no MFMA / VALU work
// the work of workgroup w of G on the launch p (its own grid, or one part of a two-part launch)
// EPI: the wq | wk | wv epilogue (MmqP::e_*) is compiled in - instantiations of their own: carried by every kernel as a run-time switch it cost the launches
// that never use it 2-6 us per layer (Qwen2.5-72B, NEOX rope: 236 -> 242 us at 8 tokens)
template <int TYPE, int NV, int ABL, bool MJ, bool EPI = false>  // NV result registers per lane in use: 4 (<= 8 tokens), 8 (<= 16), 16 (<= 32)
__device__ __forceinline__ void mmq_i8_body(const MmqP & p, const int w, const int G, uint8_t * smem) {
    typedef MT<TYPE> M;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (a scalar: the K slice, the loop counters and the buffer loads' scalar offsets derive from it)
    constexpr bool Q80 = TYPE == PM_Q8_0;                         // 32-value blocks: "super-block" = 4 blocks, no scale table in LDS (scales travel with the operands)
    const int nsb = Q80 ? (p.K + 127) / 128 : p.K / 256, npairs = (nsb + 1) >> 1;
    const int nbw = Q80 ? p.K / 32 : nsb;                         // what the tile loader counts in: blocks of the row
    float * dTl = (float *) smem;                                 // activation scales [nsb + 1][32] (0 for token slots >= T; last row all 0)
    uint8_t * stage = smem + (Q80 ? 0 : (size_t) (nsb + 1) * 128);   // NWAVE x WAVE_LDS; afterwards the 32 KB reduction buffer
    int r0 = (int) ((long) p.N * w / G), r1 = (int) ((long) p.N * (w + 1) / G);            // (MJ: virtual rows over all jobs)
    if constexpr (EPI) { r0 &= ~1; r1 &= ~1; }                    // rotation pairs stay inside a workgroup (N and the job boundaries are even)
    const int nrg = (r1 - r0 + 31) >> 5;
    const int RGB = 1 << p.rgb_log2, KS = NWAVE >> p.rgb_log2;
    const int rgi = wave & (RGB - 1), ks = wave >> p.rgb_log2;
#ifdef PM_MMQ_KCONTIG
    const int pb = npairs * ks / KS, pe = npairs * (ks + 1) / KS, pstep = 1;
#else
    // K slices INTERLEAVED over the waves that share a row group: at any time those waves read ADJACENT 128-byte pieces of a row (KS x 128 B
    // contiguous per row and stream) instead of pieces a whole slice apart - HBM sees longer bursts per open page
    const int pstep = KS, pb = ks, pe = npairs <= ks ? ks : ks + ((npairs - ks + KS - 1) / KS) * KS;      // pairs pb, pb + KS, ... < pe
#endif
    uint8_t * L = stage + wave * M::WAVE_LDS;
    const int r = lane & 31, g = lane >> 5;
    const bool act = NV == 16 ? true : r < p.T;                                     // token slots >= T: operand bytes are don't-care (scale row 0, never stored)
    ActSrc xs;
    xs.rx = __builtin_amdgcn_make_buffer_rsrc((void *) p.qT, 0, Q80 ? nbw * 1024 : nsb * 8192, 0x00020000);
    xs.rb = __builtin_amdgcn_make_buffer_rsrc((void *) p.bsT, 0, Q80 ? nbw * 128 : nsb * 1024 + 512, 0x00020000);   // (+ t_off slots of the last super-block: inside the table allocation)
    xs.xo = act ? (uint32_t) ((32 * g + r + p.t_off) * 16) : 0x80000000u;
    xs.bo = Q80 ? (uint32_t) (16 * lane) : act ? (uint32_t) ((lane + p.t_off) * 16) : 0x80000000u;   // (Q8_0: lane l carries 16 bytes of the step's [8 blocks][32 slots] activation scales)
    Rows<MJ> rw;
    // (round 6, measured and rejected: TWO weight tiles in flight per wave - a second register set, the steps taking the sets alternately. The forms with the
    //  registers for it, Q4_K up to 8 tokens, went from 170 to 248 VGPRs and - with the look-ahead past the last step made cheap - from 36.7 to 38.5 us on
    //  ffn_gate, 16.5 to 17.6 on wq, 160.5 to 165.7 us per 70B layer: the streaming part of a launch already runs at 6.3 TB/s, what a launch pays is its
    //  fixed ~8 us - profiles/r06_small_batch.txt)
    typename M::B R;
    typename M::A A0 = {}, A1 = {};                              // (inactive token lanes keep these zeros)
    // a step's weight tile goes out in FOUR parts. Issued in one burst (13 x 1 KiB per wave, 100 KB per CU) the wave sat in the issue itself until the
    // CU's miss queues had taken the whole tile - and only then started on the tile it holds: load time and compute time added up
    // (ffn_down Q6_K, 8 tokens: 41 us stream + 19 us compute + 12 fixed = 70). One part in front of every quarter of the first super-block's
    // products keeps both busy.
    auto issue_b_part = [&](typename M::B & R, int pr, int n) __attribute__((always_inline)) {
        M::issue_b_part(R, rw, nbw, pr, lane, n);
        if constexpr (TYPE == PM_Q6_K) if (n == 0) R.d = *(const PM_G uint16_t *) rw.at_d(pm_q6k_d_off((uint32_t) nsb, (uint32_t) min(2 * pr + g, nsb - 1)));   // cached: 32 steps share the line
        if constexpr (Q80) if (n == 0) M::issue_ds(R, xs, pr);
    };
    auto issue_b = [&](typename M::B & R, int pr) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 4; ++n) issue_b_part(R, pr, n);
    };
    // the first weight tile and activation slice of a row group (in flight before anything waits)
    auto first = [&](int rg) __attribute__((always_inline)) {
        const int rbase = r0 + 32 * rg;
        if constexpr (!MJ) {
            rw.base = p.W + (long) rbase * p.row_stride;
            rw.stride = (uint32_t) p.row_stride; rw.lim = r1 - 1 - rbase; rw.rr = lane >> 3; rw.rh = lane >> 1;
            rw.off_d = (uint32_t) min(r, rw.lim) * rw.stride;
        } else {
            auto rowptr = [&](int v) __attribute__((always_inline)) {                     // virtual row -> its matrix row
                v = min(v, r1 - 1);
                const int j = (v >= p.start[1] && p.njobs > 1) + (v >= p.start[2] && p.njobs > 2);
                return (j == 0 ? p.Wj[0] : j == 1 ? p.Wj[1] : p.Wj[2]) + (long) (v - (j == 0 ? 0 : j == 1 ? p.start[1] : p.start[2])) * p.row_stride;
            };
#pragma unroll
            for (int n = 0; n < 4; ++n) rw.rp[n] = rowptr(rbase + (lane >> 3) + 8 * n);
            rw.rph = rowptr(rbase + (lane >> 1)); rw.rpd = rowptr(rbase + r);
        }
        rw.lm = ~0u;
        issue_b(R, pb);
        M::template issue_a<NV>(A0, xs, 2 * pb);
        if constexpr (ABL & 1) M::template issue_a<NV>(A1, xs, 2 * pb);
    };
    if (rgi < nrg && pb < pe) first(rgi);                         // ... including the staging of the scale table:
    if constexpr (!Q80) {
        // (slots >= T masked here: a quantizer that writes the table itself fills rows < T only). Four loads per thread in flight: one load per trip made
        // ffn_down's 112 super-blocks seven dependent round trips to L2 = 5 of the launch's ~18 fixed microseconds (empty-loop ablation 13.3 us at K = 8192,
        // 18.4 at 28672, profiles/r06_small_batch.txt)
        for (int i0 = tid; i0 < nsb * 32; i0 += 4 * BLOCK) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * BLOCK;
                v[u] = (i < nsb * 32 && (i & 31) < p.T) ? *((const PM_G float *) p.dT + (i + p.t_off)) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * BLOCK; if (i < nsb * 32) dTl[i] = v[u]; }
        }
        if (tid < 32) dTl[nsb * 32 + tid] = 0.0f;
    }
    __syncthreads();
    for (int rg0 = 0; rg0 < nrg; rg0 += RGB) {
        const int rg = rg0 + rgi;
        f32x16 out = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (rg < nrg && pb < pe) {
            if (rg0 > 0) first(rg);
            // one step: the tile in `Rc` -> LDS, its two super-blocks' products; on the way the requests for the wave's next tile (into the same registers)
            // and for the next step's activations
            auto step = [&](typename M::B & Rc, int pr) __attribute__((always_inline)) {
                M::stash(Rc, L, lane);                                 // waits for this step's weights only
                const int sb0 = 2 * pr, sb1 = Q80 ? 2 * pr + 1 : min(2 * pr + 1, nsb - 1);   // (Q8_0: blocks past the row read as zeros - no clamp, no double count)
                // Issue order is the design (VMEM returns in order): the compiler's schedulers must not sink the prefetches towards
                // their uses - no conditional code in the step (an odd tail super-block is computed on clamped data with the all-zero
                // scale row) and scheduling barriers around the issue points.
                if constexpr (!(ABL & 1)) M::template issue_a<NV>(A1, xs, sb1);
                const int prn = min(pr + pstep, pe - pstep);             // the tile these registers take next
                rw.lm = pr + pstep < pe ? ~0u : 0u;                      // (past the wave's last step: a request for nothing)
                __builtin_amdgcn_sched_barrier(0);
                // (the 32-token Q6_K form has no registers for it - 404 B of spills, 88 -> 205 us: its tile still goes out in one burst)
                constexpr bool SPREAD = !(TYPE == PM_Q6_K && NV == 16) && !(ABL & 4);
                if constexpr (!SPREAD && !(ABL & 2)) { issue_b(Rc, prn); __builtin_amdgcn_sched_barrier(0); }
                auto part = [&](int n) __attribute__((always_inline)) {   // next tile, part n: unconditional, pinned between the quarters
                    if constexpr (SPREAD && !(ABL & 2)) { __builtin_amdgcn_sched_barrier(0); issue_b_part(Rc, prn, n); __builtin_amdgcn_sched_barrier(0); }
                };
                if constexpr (!(ABL & 4)) M::template compute<NV>(A0, L, 0, r, g, dTl + sb0 * 32, p.t_off, out, part);
                else { for (int s_ = 0; s_ < 8; ++s_) asm volatile("" :: "v"(A0.q[s_])); asm volatile("" :: "v"(A0.bs)); asm volatile("" :: "v"(*(const u32x4 *) (L + 16 * lane))); }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(ABL & 1)) M::template issue_a<NV>(A0, xs, min(2 * prn, nsb - 1));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(ABL & 4)) M::template compute<NV>(A1, L, 1, r, g, dTl + (2 * pr + 1 < nsb ? sb1 : nsb) * 32, p.t_off, out, [](int) {});
                else { for (int s_ = 0; s_ < 8; ++s_) asm volatile("" :: "v"(A1.q[s_])); asm volatile("" :: "v"(A1.bs)); asm volatile("" :: "v"(*(const u32x4 *) (L + 16 * lane + 1024))); }
                __builtin_amdgcn_sched_barrier(0);
            };
            for (int pr = pb; pr < pe; pr += pstep) step(R, pr);
        }
        // fixed-order sum of the K slices through LDS, epilogue, store
        __syncthreads();
        float * red = (float *) stage;
#pragma unroll
        for (int v = 0; v < NV; ++v) red[(wave * 16 + v) * 64 + lane] = out[v];
        __syncthreads();
        for (int o = tid; o < RGB * 1024; o += BLOCK) {
            const int gi = o >> 10, v = (o >> 6) & 15, l = o & 63;
            if (v >= NV) continue;
            float s = 0.0f;
            for (int k = 0; k < KS; ++k) s += red[(((k << p.rgb_log2) + gi) * 16 + v) * 64 + l];
            const int t = 8 * (v >> 2) + 4 * (l >> 5) + (v & 3), row = r0 + 32 * (rg0 + gi) + (l & 31);
            const bool live = t < p.T && row < r1;
            // the element's job: local row, output, bias
            int j = 0, lr = row, nj = p.N; const float * bj = p.bias; float * yj = p.y;
            if constexpr (MJ) {
                j = (row >= p.start[1] && p.njobs > 1) + (row >= p.start[2] && p.njobs > 2);
                lr = row - (j == 0 ? 0 : j == 1 ? p.start[1] : p.start[2]); nj = j == 0 ? p.nj[0] : j == 1 ? p.nj[1] : p.nj[2];
                bj = j == 0 ? p.bj[0] : j == 1 ? p.bj[1] : p.bj[2];
                yj = j == 0 ? p.yj[0] : j == 1 ? p.yj[1] : p.yj[2];
            }
            if (live && bj) s += ld_g(bj + lr);
            if constexpr (EPI) {
                const float sp = __shfl_xor(s, 1);                      // the pair's other row: the neighbouring lane (r0 even), same token
                if (live) {
                    const int role = j == 0 ? p.e_role[0] : j == 1 ? p.e_role[1] : p.e_role[2];
                    const int seq = p.e_seq ? *p.e_seq : 0;
                    const int pos = p.e_pos[seq] + t;
                    const long kv_off = (long) seq * p.e_seq_stride;
                    if (role == 3) p.e_vc[kv_off + (long) lr * p.e_nctx + pos] = f2h(s);
                    else {
                        const int d = lr % p.e_dh;
                        float o = s;
                        if (d < p.e_nrot) {
                            const float c = ld_g(p.e_tab + (long) t * p.e_nrot + (d & ~1)), sn = ld_g(p.e_tab + (long) t * p.e_nrot + (d & ~1) + 1);
                            o = (d & 1) ? sp * sn + s * c : s * c - sp * sn;          // o1 = x0 sin + x1 cos | o0 = x0 cos - x1 sin
                        }
                        const uint16_t hv = f2h(o);
                        if (role == 2) p.e_kc[kv_off + (long) pos * p.e_kvdim + lr] = hv;
                        else st_g(yj + (long) t * nj + lr, role == 1 ? h2f(hv) : o);
                    }
                }
            } else if (live) {
                if (p.resid) s += ld_g(p.resid + (long) t * (MJ ? nj : (int) p.y_stride) + lr);
                st_g(yj + (long) t * (MJ ? nj : (int) p.y_stride) + lr, s);
            }
        }
        __syncthreads();
    }
}

template <int TYPE, int NV, int ABL = 0, bool MJ = false, bool EPI = false>
__global__ __launch_bounds__(BLOCK, 2) void mmq_i8_kernel(MmqP p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    mmq_i8_body<TYPE, NV, ABL, MJ, EPI>(p, (int) blockIdx.x, (int) gridDim.x, smem);
}

// Two launches that share the activations as ONE grid: workgroups [0, ga) are the multi-job launch `a` (wq | wk, type TA), the rest the single matrix `b` of
// another type (wv: Q6_K or Q5_K in the Q4_K_M files, src/llama.cpp:19447 use_more_bits) - by itself a 7 MB matrix is a launch of ~14 us that 32 workgroups
// spend mostly on their fixed costs (profiles/r06_small_batch.txt); here it rides in the shadow of the large part.
struct MmqP2 { MmqP a, b; int ga; };
template <int TA, int TB, int NV, bool EPI = false>
__global__ __launch_bounds__(BLOCK, 2) void mmq_i8_dual_kernel(MmqP2 pp) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if ((int) blockIdx.x < pp.ga) mmq_i8_body<TA, NV, 0, true, EPI>(pp.a, (int) blockIdx.x, pp.ga, smem);
    else mmq_i8_body<TB, NV, 0, TB == PM_Q6_K, EPI>(pp.b, (int) blockIdx.x - pp.ga, (int) gridDim.x - pp.ga, smem);   // (Q6_K: the one-job multi-job form - no spills at 16 tokens)
}

// prologue: per super-block, the activation group sums as F16 in A-operand order and the transposed activation scales
__global__ __launch_bounds__(64) void mmq_prep_kernel(const uint8_t * xq, long xq_stride, int K, int T, uint8_t * bsT, float * dT, uint8_t * qT) {
    const int sb = (int) blockIdx.x, l = (int) threadIdx.x, t = l & 31, g = l >> 5;
    const uint8_t * row = xq + (long) min(t, T - 1) * xq_stride;
    if (t < T) {                                                   // the values in A-operand order (token slots >= T are never loaded)
#pragma unroll
        for (int s = 0; s < 8; ++s) *(u32x4 *) (qT + (size_t) sb * 8192 + s * 1024 + l * 16) = *(const u32x4 *) (row + sb * 256 + 32 * s + 16 * g);
    }
    const int16_t * bsums = (const int16_t *) (row + K + (K / 256) * 4);
    f16x8 h;
#pragma unroll
    for (int s = 0; s < 8; ++s) h[s] = t < T ? (_Float16) (float) (int) bsums[sb * 16 + 2 * s + g] : (_Float16) 0.0f;
    *(f16x8 *) (bsT + ((size_t) sb * 64 + l) * 16) = h;
    if (g == 0) dT[sb * 32 + t] = t < T ? ((const float *) (row + K))[sb] : 0.0f;
}

// Q8_0 activations (row-SoA: int8 qs[K] | f16 d[K / 32], quantize.hip) -> per block the values in A-operand order [blk][64 lanes][16 B] and the
// scales as f32 [blk][32 token slots] (0 for slots >= T)
__global__ __launch_bounds__(64) void mmq_prep_q80_kernel(const uint8_t * xq, long xq_stride, int K, int T, uint8_t * qT, float * dT) {
    const int blk = (int) blockIdx.x, l = (int) threadIdx.x, t = l & 31, g = l >> 5;
    const uint8_t * row = xq + (long) min(t, T - 1) * xq_stride;
    if (t < T) *(u32x4 *) (qT + (size_t) blk * 1024 + l * 16) = *(const u32x4 *) (row + blk * 32 + 16 * g);
    if (g == 0) dT[blk * 32 + t] = t < T ? h2f(((const uint16_t *) (row + K))[blk]) : 0.0f;
}

// row groups a workgroup works on side by side (the other waves split K): 5..7 groups take all 8 wave slots in ONE round (one fill / drain of the step
// pipeline, no K split) instead of two rounds of 4 (ffn_gate | ffn_up of the 70B shape: 7 groups per workgroup); PM355_MMQ_RGB8_MIN=8: the round-5 rule
int rgb_log2_for(int nrg) {
    static const int rgb8_min = [] { const char * e = getenv("PM355_MMQ_RGB8_MIN"); const int v = e ? atoi(e) : 0; return v >= 3 && v <= 8 ? v : 5; }();
    return nrg >= rgb8_min ? 3 : nrg >= 3 ? 2 : nrg == 2 ? 1 : 0;
}

// per-device scratch (grown on demand, like mmq.hip's f16 activation copy): quantized activations of an f32 call + the two tables
// scratch per (device, stream): the tables of a pass and the quantized copy of an f32 call are written and read in stream order, so two
// streams of one device (two ggml backends, the engine next to the plug-in) must not share them
struct Scr { hipStream_t st; uint8_t * p; size_t bytes; uint64_t use; int tab_K; };   // tab_K: the K the tables in p were last written for (0: none)
constexpr int NSCR = 8;
Scr g_scr[16][NSCR] = {};
Scr g_scr80[16][NSCR] = {};             // Q8_0 launches: their own tables (never share bytes with the Q8_K tables a later launch may re-use)
uint64_t g_tick = 0;
std::mutex g_scr_mu;

}  // namespace

size_t pm_mmq_i8_lds_bytes(int type, int K) {
    if (type == PM_Q8_0) return (size_t) NWAVE * MT<PM_Q8_0>::WAVE_LDS;
    const size_t w = type == PM_Q4_K ? MT<PM_Q4_K>::WAVE_LDS : type == PM_Q5_K ? MT<PM_Q5_K>::WAVE_LDS : MT<PM_Q6_K>::WAVE_LDS;
    return (size_t) (K / 256 + 1) * 128 + NWAVE * w;
}

// 0 when pm_launch_mmq_i8 serves this shape
int pm_mmq_i8_check(int type, int K, int N, int T) {
    if (type != PM_Q4_K && type != PM_Q5_K && type != PM_Q6_K && type != PM_Q8_0) return -1;
    if (type == PM_Q8_0) return (T < 1 || T > 64 || K % 32 || K < 512 || N < 1) ? -2 : 0;
    if (T < 1 || T > 64 || K % 256 || K < 512 || N < 1) return -2;
    if (pm_mmq_i8_lds_bytes(type, K) > 150 * 1024) return -4;
    return 0;
}

// Y[t][n] = W[n,:] . x[t,:] (+bias[n]) (+resid[t][n]) for 1 <= T <= 64 tokens (passes of up to 32). xq: activations already in the library's row-SoA Q8_K
// form (quantize.hip), or null and x_f32 [T][K] is quantized first. Y / resid token stride = N.
namespace {
// scratch of (device, stream) for K: tables of two 32-token passes + a quantized copy of up to 64 f32 rows; null: allocation failed (or,
// lookup_only, no scratch of this stream holds tables for K)
size_t scr_q_off(int K) { return ((2 * (size_t) (K / 256) * (1024 + 128) + (size_t) 64 * pm_q8k_row_bytes(K) + 256) + 255) & ~(size_t) 255; }
Scr * scratch(int dev, hipStream_t st, int K, bool lookup_only) {
    const size_t need = scr_q_off(K) + 2 * (size_t) (K / 256) * 8192;          // tables of two passes | quantized copy of 64 rows | operand-ordered values of two passes
    std::lock_guard<std::mutex> lk(g_scr_mu);
    Scr * e = nullptr, * lru = &g_scr[dev][0];
    for (Scr & c : g_scr[dev]) {
        if (c.p && c.st == st) { e = &c; break; }
        if (!c.p) { if (lru->p) lru = &c; } else if (lru->p && c.use < lru->use) lru = &c;
    }
    // lookup_only (a consumer that re-uses tables a producer wrote): the entry of this stream must still exist (not evicted by a ninth
    // stream, not re-allocated for a larger K in between) AND hold tables for this K - else the launch fails instead of reading zeros
    if (lookup_only) {
        if (!e || e->tab_K != K || e->bytes < need) return nullptr;
        e->use = ++g_tick;
        return e;
    }
    if (e && e->bytes >= need) { e->use = ++g_tick; e->tab_K = K; return e; }
    if (!e) e = lru;
    if (e->p) { (void) hipDeviceSynchronize(); (void) hipFree(e->p); e->p = nullptr; e->bytes = 0; }
    if (hipMalloc((void **) &e->p, need) != hipSuccess) { e->p = nullptr; return nullptr; }
    // table slots no quantizer ever wrote stay finite (they are multiplied by scale 0). ON THE OWNING STREAM: a null-stream memset is not ordered
    // against a non-blocking stream and zeroed the tables the first producer had just written (two ranks on one GPU, tests/test_gpu_ring.py)
    (void) hipMemsetAsync(e->p, 0, need, st);
    e->st = st; e->bytes = need; e->use = ++g_tick; e->tab_K = K;
    return e;
}
// activation tables of every 32-token pass
void launch_prep(const Scr * sc, const void * xq, int K, int T, hipStream_t st) {
    const int nsb = K / 256;
    const size_t xrow = pm_q8k_row_bytes(K), tab = (size_t) nsb * (1024 + 128);
    for (int t0 = 0, c = 0; t0 < T; t0 += 32, ++c) {
        uint8_t * bsT = sc->p + c * tab;
        hipLaunchKernelGGL(mmq_prep_kernel, dim3(nsb), dim3(64), 0, st, (const uint8_t *) xq + (size_t) t0 * xrow, (long) xrow, K, T - t0 < 32 ? T - t0 : 32,
                           bsT, (float *) (bsT + (size_t) nsb * 1024), sc->p + scr_q_off(K) + (size_t) c * nsb * 8192);
    }
}
}  // namespace

namespace {
// Q8_0 weights x Q8_0 activations, passes of up to 16 tokens. xq: row-SoA Q8_0 rows (quantize.hip) or null and x_f32 is quantized first. Its tables
// (operand-ordered values + f32 scales of both 32-slot halves, then a quantized copy of 64 f32 rows) are written by a prologue launch per call.
Scr * scratch80(int dev, hipStream_t st, size_t need) {
    std::lock_guard<std::mutex> lk(g_scr_mu);
    Scr * e = nullptr, * lru = &g_scr80[dev][0];
    for (Scr & c : g_scr80[dev]) {
        if (c.p && c.st == st) { e = &c; break; }
        if (!c.p) { if (lru->p) lru = &c; } else if (lru->p && c.use < lru->use) lru = &c;
    }
    if (!e || e->bytes < need) {
        if (!e) e = lru;
        if (e->p) { (void) hipDeviceSynchronize(); (void) hipFree(e->p); e->p = nullptr; e->bytes = 0; }
        if (hipMalloc((void **) &e->p, need) != hipSuccess) { e->p = nullptr; return nullptr; }
        e->st = st; e->bytes = need;
    }
    e->use = ++g_tick;
    return e;
}
size_t q80_need(int K) { return 2 * ((size_t) (K / 32) * 1024 + (size_t) (K / 32) * 128) + (size_t) 64 * pm_q80_row_bytes(K) + 512; }

// reuse_prep: the tables of this (device, stream) already describe these activations (pm_launch_silu_mul_q80_tab wrote them): no quantizer, no prologue launch
int launch_q80(const void * W, const void * xq, const float * x_f32, float * Y, int K, int N, int T, const float * bias, const float * resid, int reuse_prep, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    const int nb = K / 32;
    const size_t xrow = pm_q80_row_bytes(K), qtab = (size_t) nb * 1024, dtab = (size_t) nb * 128;
    Scr * e = scratch80(dev, st, q80_need(K));
    if (!e) return -3;
    uint8_t * base = e->p;
    if (!reuse_prep) {
    if (!xq) {
        uint8_t * q = base + 2 * (qtab + dtab) + 256;
        pm_launch_quantize_q80(x_f32, q, K, T, st);
        xq = q;
    }
    for (int t0 = 0, c = 0; t0 < T; t0 += 32, ++c)
        hipLaunchKernelGGL(mmq_prep_q80_kernel, dim3(nb), dim3(64), 0, st, (const uint8_t *) xq + (size_t) t0 * xrow, (long) xrow, K, T - t0 < 32 ? T - t0 : 32,
                           base + c * (qtab + dtab), (float *) (base + c * (qtab + dtab) + qtab));
    }
    const int cus = pm_device_cus();
    const int grid = N / 32 >= cus ? cus : (N + 31) / 32;
    const int rows = (N + grid - 1) / grid, nrg = (rows + 31) / 32;
    const size_t lds = pm_mmq_i8_lds_bytes(PM_Q8_0, K);
    constexpr int tmax = 16;                         // (the 32-token form spills 236 B: 502 vs 422 us per Qwen2.5-72B layer at 32 tokens - two passes of 16 it is)
    for (int t0 = 0; t0 < T; t0 += tmax) {
        const int tn = T - t0 < tmax ? T - t0 : tmax;
        uint8_t * tab = base + (t0 / 32) * (qtab + dtab);
        MmqP p = {};
        p.t_off = t0 % 32;
        p.W = (const uint8_t *) W; p.row_stride = (long) pm_weight_row_stride(PM_Q8_0, K); p.N = N; p.K = K; p.T = tn;
        p.qT = tab; p.bsT = tab + qtab;                            // (bsT: here the f32 activation scales [blk][32])
        p.y = Y + (size_t) t0 * N; p.y_stride = N; p.bias = bias; p.resid = resid ? resid + (size_t) t0 * N : nullptr;
        p.rgb_log2 = rgb_log2_for(nrg);
        auto go = [&](auto kern) {
            pm_allow_big_lds((const void *) kern, 150 * 1024);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, st, p);
        };
        if (tn <= 8) go(mmq_i8_kernel<PM_Q8_0, 4>); else go(mmq_i8_kernel<PM_Q8_0, 8>);
    }
    return 0;
}
}  // namespace

// Q8_0 weights: where a producer (pm_launch_silu_mul_q80_tab) writes the activation tables of this (device, stream) for K; per 32-token pass qtab + dtab bytes
int pm_mmq_i8_q80_tables(int K, hipStream_t st, void ** tab, size_t * qtab_bytes, size_t * dtab_bytes) {
    if (K % 32 || K < 512) return -2;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    Scr * e = scratch80(dev, st, q80_need(K));
    if (!e) return -3;
    *tab = e->p; *qtab_bytes = (size_t) (K / 32) * 1024; *dtab_bytes = (size_t) (K / 32) * 128;
    return 0;
}

// Where the tables for K live: pm_launch_quantize_q8k / pm_launch_rmsnorm_q8k write them as a second output (<= 64 rows), the mat-mul is
// then launched with reuse_prep = 1 and no prologue launch at all.
int pm_mmq_i8_tables(int K, hipStream_t st, pm_q8k_tables * out) {
    if (K % 256 || K < 512) return -2;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    const Scr * sc = scratch(dev, st, K, false);
    if (!sc) return -3;
    out->base = sc->p; out->nsb = K / 256; out->tab_bytes = (size_t) (K / 256) * (1024 + 128); out->qbase = sc->p + scr_q_off(K);
    return 0;
}

// The prologue alone (tables for xq = T rows of row-SoA Q8_K): lets a caller fork streams between the prologue and several
// pm_launch_mmq_i8(..., reuse_prep = 1, ...) launches that share the activations.
int pm_launch_mmq_i8_prep(const void * xq, int K, int T, hipStream_t st) {
    if (T < 1 || T > 64 || K % 256 || K < 512) return -2;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    const Scr * sc = scratch(dev, st, K, false);
    if (!sc) return -3;
    launch_prep(sc, xq, K, T, st);
    return 0;
}

// reuse_prep != 0: the previous pm_launch_mmq_i8 / pm_launch_mmq_i8_prep on this device had the SAME activations (xq / x_f32 contents, T, K)
// and is ordered before this launch: its quantized copy and tables are still valid, skip the prologue launches (q/k/v and gate/up share
// one activation set).
int pm_launch_mmq_i8(int type, const void * W, const void * xq, const float * x_f32, float * Y, int K, int N, int T,
                     const float * bias, const float * resid, int reuse_prep, hipStream_t st) {
    const int rc = pm_mmq_i8_check(type, K, N, T);
    if (rc) return rc;
    if (type == PM_Q8_0) return launch_q80(W, xq, x_f32, Y, K, N, T, bias, resid, reuse_prep, st);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    const int nsb = K / 256;
    const size_t xrow = pm_q8k_row_bytes(K), tab = (size_t) nsb * (1024 + 128);
    const Scr * sc = scratch(dev, st, K, reuse_prep != 0);
    if (!sc) return -3;
    if (!xq) {
        uint8_t * q = sc->p + 2 * tab;
        if (!reuse_prep) pm_launch_quantize_q8k(x_f32, q, K, T, st);
        xq = q;
    }
    if (!reuse_prep) launch_prep(sc, xq, K, T, st);
    const int cus = pm_device_cus();
    const int grid = N / 32 >= cus ? cus : (N + 31) / 32;
    const int rows = (N + grid - 1) / grid, nrg = (rows + 31) / 32;
    const size_t lds = pm_mmq_i8_lds_bytes(type, K);
    const int tmax = type == PM_Q5_K ? 16 : 32;                    // (Q5_K: 11 weight registers more per step - its 32-token form spills 388 B / lane)
    for (int t0 = 0, c = 0; t0 < T; t0 += tmax, ++c) {
        const int tn = T - t0 < tmax ? T - t0 : tmax;
        uint8_t * bsT = sc->p + (t0 / 32) * tab; float * dT = (float *) (bsT + (size_t) nsb * 1024);    // tables are per 32 tokens
        const uint8_t * xc = (const uint8_t *) xq + (size_t) t0 * xrow;
        MmqP p = {};
        p.t_off = t0 % 32;
        p.W = (const uint8_t *) W; p.row_stride = (long) pm_weight_row_stride(type, K); p.N = N; p.K = K; p.T = tn;
        p.xq = xc; p.xq_stride = (long) xrow; p.bsT = bsT; p.dT = dT; p.qT = sc->p + scr_q_off(K) + (size_t) (t0 / 32) * nsb * 8192;
        p.y = Y + (size_t) t0 * N; p.y_stride = N; p.bias = bias; p.resid = resid ? resid + (size_t) t0 * N : nullptr;
        p.rgb_log2 = rgb_log2_for(nrg);
        auto go = [&](auto kern) {
            pm_allow_big_lds((const void *) kern, 150 * 1024);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, st, p);
        };
#ifdef PM_MMQ_ABLATE      // measurement build (profiles/r02_small_batch_probe.txt): PM355_MMQ_ABL = 1 no activation loads | 2 no weight loads | 4 no compute
        static const int abl = [] { const char * e = getenv("PM355_MMQ_ABL"); return e ? atoi(e) : 0; }();
        if (abl && tn <= 8) {
#define PM_ABL_CASE(T_) switch (abl) { case 1: go(mmq_i8_kernel<T_, 4, 1>); break; case 2: go(mmq_i8_kernel<T_, 4, 2>); break; case 3: go(mmq_i8_kernel<T_, 4, 3>); break; \
                                       case 4: go(mmq_i8_kernel<T_, 4, 4>); break; case 5: go(mmq_i8_kernel<T_, 4, 5>); break; case 6: go(mmq_i8_kernel<T_, 4, 6>); break; \
                                       default: go(mmq_i8_kernel<T_, 4, 7>); break; }
            if (type == PM_Q4_K) PM_ABL_CASE(PM_Q4_K) else PM_ABL_CASE(PM_Q6_K)
#undef PM_ABL_CASE
        } else
#endif
        if (type == PM_Q4_K) { if (tn <= 8) go(mmq_i8_kernel<PM_Q4_K, 4>); else if (tn <= 16) go(mmq_i8_kernel<PM_Q4_K, 8>); else go(mmq_i8_kernel<PM_Q4_K, 16>); }
        else if (type == PM_Q5_K) { if (tn <= 8) go(mmq_i8_kernel<PM_Q5_K, 4>); else go(mmq_i8_kernel<PM_Q5_K, 8>); }
        else if (tn <= 8)    go(mmq_i8_kernel<PM_Q6_K, 4>);
        else if (tn <= 16) {
            // the 16-token Q6_K form with wave-uniform row addressing spills 44 B per lane; the one with per-lane row pointers (the multi-job form) does not
            // (249 registers): a launch of ONE job in that form
            p.njobs = 1; p.y_stride = N;
            for (int j = 0; j < 3; ++j) { p.Wj[j] = p.W; p.yj[j] = p.y; p.bj[j] = p.bias; p.nj[j] = N; p.start[j] = j == 0 ? 0 : N; }
            go(mmq_i8_kernel<PM_Q6_K, 8, 0, true>);
        }
        else                 go(mmq_i8_kernel<PM_Q6_K, 16>);
    }
    return 0;
}

namespace {
// wq | wk | wv epilogue of a launch (MmqP::e_*): NORM-mode rope, transposed F16 V cache, engine mode (cell = position). false: not served
bool fill_epi(MmqP & p, const pm_qkv_epi * e, int r0, int r1, int r2) {
    if (!e) return true;
    if (e->neox || e->v_rowmajor || e->dyn || !e->tab || !e->pos || !e->kc || !e->vc || e->dh < 2 || (e->dh & 1) || (e->n_rot & 1)) return false;
    p.e_on = 1; p.e_role[0] = r0; p.e_role[1] = r1; p.e_role[2] = r2;
    p.e_tab = e->tab; p.e_pos = e->pos; p.e_seq = e->seq; p.e_seq_stride = e->seq_stride; p.e_kc = (uint16_t *) e->kc; p.e_vc = (uint16_t *) e->vc;
    p.e_dh = e->dh; p.e_nctx = e->n_ctx; p.e_nrot = e->n_rot; p.e_kvdim = e->Hkv * e->dh;
    return true;
}
}  // namespace

// Up to 3 matrices of ONE type and K sharing the activations (wq | wk | wv, ffn_gate | ffn_up) in one launch: one fill / drain of the
// step pipeline instead of two or three (~10 us each). T <= 64 in passes of 32 (round 4: with the operand-ordered activation table the 32-token
// instantiations have the registers for per-lane row pointers: 238 / 250 VGPRs); -5: not served, launch the jobs one by one. y[j]: [T][N[j]].
int pm_launch_mmq_i8_multi(int type, int njobs, const void * const * W, const int * N, float * const * Y, const float * const * bias, const void * xq,
                           int K, int T, int reuse_prep, hipStream_t st, const pm_qkv_epi * epi) {
    if (njobs < 2 || njobs > 3 || T > 64 || (type != PM_Q4_K && type != PM_Q6_K)) return -5;
    // epi: the jobs are wq, wk, wv (three jobs, one pass, even row counts whose heads are whole)
    if (epi && (njobs != 3 || T > 32 || N[0] % epi->dh || N[1] != epi->Hkv * epi->dh || N[2] != N[1])) return -5;
    long total = 0;
    for (int j = 0; j < njobs; ++j) { if (pm_mmq_i8_check(type, K, N[j], T)) return -5; total += N[j]; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    const int nsb = K / 256;
    const size_t xrow = pm_q8k_row_bytes(K);
    const Scr * sc = scratch(dev, st, K, reuse_prep != 0);
    if (!sc) return -3;
    if (!reuse_prep) launch_prep(sc, xq, K, T, st);
    const int cus = pm_device_cus();
    const int grid = total / 32 >= cus ? cus : (int) ((total + 31) / 32);
    const int rows = (int) ((total + grid - 1) / grid), nrg = (rows + 31) / 32;
    MmqP p = {};
    p.row_stride = (long) pm_weight_row_stride(type, K); p.N = (int) total; p.K = K; p.T = T;
    p.xq = (const uint8_t *) xq; p.xq_stride = (long) xrow; p.bsT = sc->p; p.dT = (const float *) (sc->p + (size_t) nsb * 1024); p.qT = sc->p + scr_q_off(K);
    p.njobs = njobs;
    int at = 0;
    for (int j = 0; j < 3; ++j) {
        const int jj = j < njobs ? j : njobs - 1;
        p.Wj[j] = (const uint8_t *) W[jj]; p.yj[j] = Y[jj]; p.bj[j] = bias ? bias[jj] : nullptr; p.nj[j] = N[jj];
        p.start[j] = j < njobs ? at : (int) total;
        if (j < njobs) at += N[j];
    }
    p.rgb_log2 = rgb_log2_for(nrg);
    if (!fill_epi(p, epi, 1, 2, 3)) return -5;
    const size_t lds = pm_mmq_i8_lds_bytes(type, K);
    auto go = [&](auto kern) {
        pm_allow_big_lds((const void *) kern, 150 * 1024);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, st, p);
    };
    const size_t tab = (size_t) nsb * (1024 + 128);
    for (int t0 = 0, c = 0; t0 < T; t0 += 32, ++c) {                 // passes of 32 tokens, each with its own tables
        const int tn = T - t0 < 32 ? T - t0 : 32;
        p.T = tn;
        p.bsT = sc->p + c * tab; p.dT = (const float *) (p.bsT + (size_t) nsb * 1024); p.qT = sc->p + scr_q_off(K) + (size_t) c * nsb * 8192;
        for (int j = 0; j < 3; ++j) { const int jj = j < njobs ? j : njobs - 1; p.yj[j] = Y[jj] + (size_t) t0 * N[jj]; }
        if (p.e_on && type == PM_Q4_K) { if (tn <= 8) go(mmq_i8_kernel<PM_Q4_K, 4, 0, true, true>); else if (tn <= 16) go(mmq_i8_kernel<PM_Q4_K, 8, 0, true, true>); else go(mmq_i8_kernel<PM_Q4_K, 16, 0, true, true>); }
        else if (p.e_on)         { if (tn <= 8) go(mmq_i8_kernel<PM_Q6_K, 4, 0, true, true>); else if (tn <= 16) go(mmq_i8_kernel<PM_Q6_K, 8, 0, true, true>); else go(mmq_i8_kernel<PM_Q6_K, 16, 0, true, true>); }
        else if (type == PM_Q4_K) { if (tn <= 8) go(mmq_i8_kernel<PM_Q4_K, 4, 0, true>); else if (tn <= 16) go(mmq_i8_kernel<PM_Q4_K, 8, 0, true>); else go(mmq_i8_kernel<PM_Q4_K, 16, 0, true>); }
        else                 { if (tn <= 8) go(mmq_i8_kernel<PM_Q6_K, 4, 0, true>); else if (tn <= 16) go(mmq_i8_kernel<PM_Q6_K, 8, 0, true>); else go(mmq_i8_kernel<PM_Q6_K, 16, 0, true>); }
    }
    return 0;
}

// wq | wk (na matrices of type ta = Q4_K, as in pm_launch_mmq_i8_multi) AND one matrix of another K-quant type (wv: Q6_K / Q5_K) over the same activations as
// ONE grid (mmq_i8_dual_kernel): the device's workgroups are divided by weight bytes. T <= 32 (Q5_K: 16) - one pass; -5: not served, launch them separately.
int pm_launch_mmq_i8_dual(int ta, int na, const void * const * Wa, const int * Na, float * const * Ya, const float * const * ba,
                          int tb, const void * Wb, int Nb, float * Yb, const float * bb, const void * xq, int K, int T, int reuse_prep, hipStream_t st,
                          const pm_qkv_epi * epi) {
    static const bool off = [] { const char * e = getenv("PM355_MMQ_DUAL"); return e && e[0] == '0'; }();
    if (off || ta != PM_Q4_K || (tb != PM_Q6_K && tb != PM_Q5_K) || na < 1 || na > 3 || T < 1 || T > (tb == PM_Q5_K ? 16 : 32)) return -5;
    long total = 0;
    for (int j = 0; j < na; ++j) { if (pm_mmq_i8_check(ta, K, Na[j], T)) return -5; total += Na[j]; }
    if (pm_mmq_i8_check(tb, K, Nb, T)) return -5;
    // epi: part a = wq, wk; part b = wv
    if (epi && (na != 2 || Na[0] % epi->dh || Na[1] != epi->Hkv * epi->dh || Nb != Na[1])) return -5;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    const int nsb = K / 256;
    const size_t xrow = pm_q8k_row_bytes(K);
    const Scr * sc = scratch(dev, st, K, reuse_prep != 0);
    if (!sc) return -3;
    if (!reuse_prep) launch_prep(sc, xq, K, T, st);
    const int cus = pm_device_cus();
    const double bytes_a = (double) total * pm_weight_row_stride(ta, K), bytes_b = (double) Nb * pm_weight_row_stride(tb, K);
    int gb = (int) (cus * bytes_b / (bytes_a + bytes_b) + 0.5);
    gb = gb < 1 ? 1 : gb;
    if (gb > (Nb + 31) / 32) gb = (Nb + 31) / 32;
    int ga = cus - gb;
    if (ga > (int) ((total + 31) / 32)) ga = (int) ((total + 31) / 32);
    if (ga < 1) return -5;
    MmqP2 pp = {};
    {
        MmqP & p = pp.a;
        p.row_stride = (long) pm_weight_row_stride(ta, K); p.N = (int) total; p.K = K; p.T = T;
        p.xq = (const uint8_t *) xq; p.xq_stride = (long) xrow; p.bsT = sc->p; p.dT = (const float *) (sc->p + (size_t) nsb * 1024); p.qT = sc->p + scr_q_off(K);
        p.njobs = na;
        int at = 0;
        for (int j = 0; j < 3; ++j) {
            const int jj = j < na ? j : na - 1;
            p.Wj[j] = (const uint8_t *) Wa[jj]; p.yj[j] = Ya[jj]; p.bj[j] = ba ? ba[jj] : nullptr; p.nj[j] = Na[jj];
            p.start[j] = j < na ? at : (int) total;
            if (j < na) at += Na[j];
        }
        const int rows = (int) ((total + ga - 1) / ga);
        p.rgb_log2 = rgb_log2_for((rows + 31) / 32);
    }
    {
        MmqP & p = pp.b;
        p.W = (const uint8_t *) Wb; p.row_stride = (long) pm_weight_row_stride(tb, K); p.N = Nb; p.K = K; p.T = T;
        p.xq = (const uint8_t *) xq; p.xq_stride = (long) xrow; p.bsT = pp.a.bsT; p.dT = pp.a.dT; p.qT = pp.a.qT;
        p.y = Yb; p.y_stride = Nb; p.bias = bb;
        p.njobs = 1;
        for (int j = 0; j < 3; ++j) { p.Wj[j] = p.W; p.yj[j] = Yb; p.bj[j] = bb; p.nj[j] = Nb; p.start[j] = j == 0 ? 0 : Nb; }
        const int rows = (Nb + gb - 1) / gb;
        p.rgb_log2 = rgb_log2_for((rows + 31) / 32);
    }
    pp.ga = ga;
    if (!fill_epi(pp.a, epi, 1, 2, 0) || !fill_epi(pp.b, epi, 3, 0, 0)) return -5;
    const size_t la = pm_mmq_i8_lds_bytes(ta, K), lb = pm_mmq_i8_lds_bytes(tb, K), lds = la > lb ? la : lb;
    auto go = [&](auto kern) {
        pm_allow_big_lds((const void *) kern, 150 * 1024);
        hipLaunchKernelGGL(kern, dim3(ga + gb), dim3(BLOCK), lds, st, pp);
    };
    if (epi) {
        if (tb == PM_Q6_K) { if (T <= 8) go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q6_K, 4, true>); else if (T <= 16) go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q6_K, 8, true>); else go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q6_K, 16, true>); }
        else               { if (T <= 8) go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q5_K, 4, true>); else go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q5_K, 8, true>); }
    }
    else if (tb == PM_Q6_K) { if (T <= 8) go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q6_K, 4>); else if (T <= 16) go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q6_K, 8>); else go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q6_K, 16>); }
    else               { if (T <= 8) go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q5_K, 4>); else go(mmq_i8_dual_kernel<PM_Q4_K, PM_Q5_K, 8>); }
    return 0;
}
