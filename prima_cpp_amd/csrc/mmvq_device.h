// mmvq_device.h — device code of the batch-1 quantized mat-vec (shared by mmvq.hip and the multi-column mmvq_cols.hip).
// See mmvq.hip for the design notes.
#pragma once
#include "pm355_device.h"
#include "pm355_kernels.h"
#include "attn_tail_device.h"
#include <limits.h>

#define PM_MAX_ROWS_PER_WG 512

namespace pmv {


struct GemvJob {
    const uint8_t * W; const uint8_t * W2; float * y; const float * bias; const float * resid;
    long row_stride;
    int N, is_b /*uses TB*/, U /*units per row*/;
    int role;    // QKV epilogue (GemvP::epi): 0 none, 1 = query rows (rotate, round to F16), 2 = key rows (rotate, store in the K cache), 3 = value rows (store in the V cache)
    int split;   // items of this job are (row, chunk) steps instead of whole rows: a few long rows spread over all waves (wk / wv next to wq)
    // NEOX rope in the QKV epilogue (build_qwen2): a rotation pair is (i, i + n_rot / 2) of one head (ggml.c:14238-14253), so a workgroup's slice of
    // s = 2^nx_s rows is made of TWO runs of s / 2 rows, n_rot / 2 apart: logical row L of the job (what the item loops count in) is matrix row
    // job_row(L). nx_s == 0: identity (NORM rope pairs adjacent rows; every other job).
    int nx_s, nx_dh /*log2 head_dim*/, nx_hrot /*n_rot / 2*/;
};
// logical -> matrix row of a job (wave-uniform arithmetic: shifts and masks, head_dim and the slice size are powers of two)
__device__ __forceinline__ int job_row(const GemvJob & jb, int L) {
    const int sh = jb.nx_s > 0 ? jb.nx_s : 1, half = 1 << (sh - 1);                // (branch-free: scalar selects, the row loops stay straight-line code)
    const int w = L & ((1 << jb.nx_dh) - 1), l = w & ((1 << sh) - 1), q = w >> sh;
    const int m = (L - w) + q * half + (l < half ? l : jb.nx_hrot + l - half);
    return jb.nx_s > 0 ? m : L;
}
// RoPE + F16 KV store in the epilogue of the wq | wk | wv launch (NORM-mode rope: a pair = two adjacent rows of one workgroup's slice).
// tab[i] = (cos, sin) of rotation pair i at this token's position, built once per token by rope_table_kernel (layer_ops.hip) with the
// reference's running product (ggml_rope_cache_init, ggml.c:14117-14131); cell / sequence as in attn_device.h (dyn: ggml-graph mode).
struct QkvEpi {
    const float * tab; const int32_t * pos_ptr, * seq_ptr, * dyn; long seq_stride;
    uint16_t * kc, * vc; int kv_dim /*Hkv * dh*/, dh, n_ctx, n_rot, v_rowmajor;
    int neox;    // rotation pairs (i, i + n_rot / 2) instead of (2 i, 2 i + 1): the q / k jobs carry the row mapping (GemvJob::nx_s)
    // attention in the tail (pm_qkv_epi::att_out): q = the rotated query rows this launch stores, wgs = workgroups per KV-head group (power of two)
    float * att_out; const float * att_q; unsigned * att_ticket; int * att_err; float att_scale; int att_H, att_wgs;
};
struct GemvP {
    GemvJob job[3];
    const uint8_t * xq;                     // xmode 0: pre-quantized activation row (row-SoA Q8_K / Q8_0)
    const float * xf; const float * norm_w; // xmode 1: f32 activation; xmode 2: rms_norm(xf) * norm_w first; xmode 3: the same with the
                                            // sum of squares of xf taken from the PRODUCER's per-workgroup partials (ss_in[n_ss], see ss_out)
    float eps;
    // producer-side sum of squares (round 5): a single-job launch whose output row is the next launch's rms_norm input stores, per workgroup,
    // the f64 sum of the f32-rounded squares of the rows it wrote (ss_out[blockIdx.x]); the consuming launch (xmode 3) adds the n_ss partials
    // instead of reducing the row itself - its prologue loses the reduction pass and one workgroup barrier (ggml.c:11975-11980: the reference
    // sums (ggml_float)(x[i] * x[i]) in f64; any order of these <= 2^15 terms agrees after the final rounding to f32, as for xmode 2)
    double * ss_out; const double * ss_in; int n_ss;
    int xmode, K;
    int32_t * dbg;
    int ncols;                              // activation columns served by one launch (xmode 0 only when > 1)
    long xq_stride, y_stride;               // bytes between quantized activation rows; floats between output columns
    unsigned long long * ts;                // measurement builds (-DPM_TS): this launch's timestamp slot, else null
    QkvEpi epi;                             // used by the EPI instantiation only
};

// Per-type traits. A unit's NV values come in NV/16 groups of 16 CONTIGUOUS activations; group_base() gives the
// activation index of group g. X holds the quantized activation slice of one unit.
template <int NV> struct XT { uint32_t q[NV / 4]; int gs[NV / 16]; float yd; };

template <int TYPE> struct QT;

// ---------------------------------------------------------------- Q4_K (row-SoA: qa | qb | hdr) -------
// unit u = (block b = u / 4, j = u % 4): the 32 qs bytes [32j, 32j+32) = sub-blocks 2j (low nibbles, values 64j..64j+31)
// and 2j+1 (high nibbles, values 64j+32..64j+63) -> 64 CONTIGUOUS activations, one scale decode per 64 weights.
// HBM row: qa[U][16] (first 16 bytes of every unit) | qb[U][16] (second 16) | hdr[nb][16] (d, dmin, scales[12]): each of
// the three wave-level loads covers ONE contiguous span (1 KB / 1 KB / 256 B), no cache line is touched by two loads.
template <> struct QT<PM_Q4_K> {
    static constexpr int NV = 64, LPB = 4, ABLK = 256;
    typedef XT<NV> X;
    struct Wr { u32x4 q0, q1, h; };
    static __device__ __forceinline__ int group_base(int u, int g) { return (u >> 2) * 256 + 64 * (u & 3) + 16 * g; }
    static __device__ __forceinline__ void issue(Wr & w, const uint8_t * row /*wave-uniform*/, int K, int u) {
        const uint32_t nb = (uint32_t) K / 256;                         // scalar stream bases + 32-bit lane offsets
        w.q0 = ld_nt16(row + (uint32_t) u * 16u);
        w.q1 = ld_nt16(row + nb * 64 + (uint32_t) u * 16u);
        w.h  = ld_nt16(row + nb * 128 + ((uint32_t) u >> 2) * 16u);
    }
    static __device__ __forceinline__ float consume(const Wr & w, const X & x, int u, float acc, int & isum, int & msum) {
        int slo = 0, shi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            slo = dot4(w.q0[i] & 0x0F0F0F0Fu, x.q[i], slo);
            shi = dot4((w.q0[i] >> 4) & 0x0F0F0F0Fu, x.q[8 + i], shi);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            slo = dot4(w.q1[i] & 0x0F0F0F0Fu, x.q[4 + i], slo);
            shi = dot4((w.q1[i] >> 4) & 0x0F0F0F0Fu, x.q[12 + i], shi);
        }
        int sc0, sc1, m0, m1;
        k4_scale_min_pair(w.h[1], w.h[2], w.h[3], u & 3, sc0, sc1, m0, m1);
        isum = __mul24(sc0, slo) + __mul24(sc1, shi);                  // |slo| <= 32*15*127, scales <= 63: 24-bit safe
        msum = __mul24(m0, x.gs[0] + x.gs[1]) + __mul24(m1, x.gs[2] + x.gs[3]);
        const float d = h2f((uint16_t) (w.h[0] & 0xFFFF)), dmin = h2f((uint16_t) (w.h[0] >> 16));
        return fmaf(x.yd * d, (float) isum, fmaf(-(x.yd * dmin), (float) msum, acc));
    }
};

// ---------------------------------------------------------------- Q5_K (native 176-B blocks) -------
template <> struct QT<PM_Q5_K> {
    static constexpr int NV = 32, LPB = 8, ABLK = 256;
    typedef XT<NV> X;
    struct Wr { u32x4 q, h, qh; };
    static __device__ __forceinline__ int group_base(int u, int g) { const int c = u & 7; return (u >> 3) * 256 + 64 * (c >> 1) + 16 * (c & 1) + 32 * g; }
    static __device__ __forceinline__ void issue(Wr & w, const uint8_t * row, int, int u) {
        const uint32_t hb = (uint32_t) (u >> 3) * PM_BS_Q5_K;
        w.q  = ld_nt16(row + (hb + 48u + 16u * (uint32_t) (u & 7)));
        w.qh = ld_nt16(row + (hb + 16u + 16u * (uint32_t) (u & 1)));
        w.h  = ld_nt16(row + hb);
    }
    static __device__ __forceinline__ float consume(const Wr & w, const X & x, int u, float acc, int & isum, int & msum) {
        const int j = (u & 7) >> 1;
        int slo = 0, shi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = (w.q[i] & 0x0F0F0F0Fu) | (((w.qh[i] >> (2 * j)) & 0x01010101u) << 4);
            const uint32_t hi = ((w.q[i] >> 4) & 0x0F0F0F0Fu) | (((w.qh[i] >> (2 * j + 1)) & 0x01010101u) << 4);
            slo = dot4(lo, x.q[i], slo);
            shi = dot4(hi, x.q[4 + i], shi);
        }
        int sc0, sc1, m0, m1;
        k4_scale_min_pair(w.h[1], w.h[2], w.h[3], j, sc0, sc1, m0, m1);
        isum = __mul24(sc0, slo) + __mul24(sc1, shi);
        msum = __mul24(m0, x.gs[0]) + __mul24(m1, x.gs[1]);
        const float d = h2f((uint16_t) (w.h[0] & 0xFFFF)), dmin = h2f((uint16_t) (w.h[0] >> 16));
        return fmaf(x.yd * d, (float) isum, fmaf(-(x.yd * dmin), (float) msum, acc));
    }
};

// ---------------------------------------------------------------- Q6_K (row-SoA) --------------------
// row: la[U][16] | lb[U][16] | qh[U][16] | per 8 blocks: scales[8][16] d[8] (pm355_device.h)   unit u = 4b + 2hh + v = (block b, half hh, 16-col slice v)
//      la = ql[64hh+16v, +16), lb = ql[64hh+32+16v, +16), qh = qh[32hh+16v, +16): every load is one contiguous span per wave
template <> struct QT<PM_Q6_K> {
    static constexpr int NV = 64, LPB = 4, ABLK = 256;
    typedef XT<NV> X;
    struct Wr { u32x4 l0, l1, h; u32x2 s; uint16_t d; };
    static __device__ __forceinline__ int group_base(int u, int g) { return (u >> 2) * 256 + 128 * ((u >> 1) & 1) + 32 * g + 16 * (u & 1); }
    static __device__ __forceinline__ void issue(Wr & w, const uint8_t * row, int K, int u) {
        const uint32_t nb = (uint32_t) K / 256;                         // wave-uniform stream bases, 32-bit lane offsets
        const uint32_t b = (uint32_t) u >> 2, hh = ((uint32_t) u >> 1) & 1;
        w.l0 = ld_nt16(row + (uint32_t) u * 16u);
        w.l1 = ld_nt16(row + nb * 64 + (uint32_t) u * 16u);
        w.h  = ld_nt16(row + nb * 128 + (uint32_t) u * 16u);
        // (round 5, measured: scales and d through the caches instead of non-temporal - so that the four steps sharing the 128-byte line of d hit it -
        //  is 2.1 % SLOWER on the 70B token, with either tail layout: profiles/r05_ab_sumsq_q6k_tail.txt. -DPM_Q6K_TAIL_NT=0 builds the cached loads)
#if !defined(PM_Q6K_TAIL_NT) || PM_Q6K_TAIL_NT
        w.s  = ld_nt8(row + (pm_q6k_sc_off(nb, b) + 8 * hh));
        w.d  = ld_nt2(row + pm_q6k_d_off(nb, b));
#else
        w.s  = *(const PM_G u32x2 *) (row + (pm_q6k_sc_off(nb, b) + 8 * hh));
        w.d  = *(const PM_G uint16_t *) (row + pm_q6k_d_off(nb, b));
#endif
    }
    static __device__ __forceinline__ float consume(const Wr & w, const X & x, int u, float facc, int & isum, int & msum) {
        const int v = u & 1;
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t h = w.h[i];
            acc[0] = dot4((w.l0[i] & 0x0F0F0F0Fu)        | ((h << 4) & 0x30303030u), x.q[i],      acc[0]);
            acc[1] = dot4((w.l1[i] & 0x0F0F0F0Fu)        | ((h << 2) & 0x30303030u), x.q[4 + i],  acc[1]);
            acc[2] = dot4(((w.l0[i] >> 4) & 0x0F0F0F0Fu) | (h & 0x30303030u),        x.q[8 + i],  acc[2]);
            acc[3] = dot4(((w.l1[i] >> 4) & 0x0F0F0F0Fu) | ((h >> 2) & 0x30303030u), x.q[12 + i], acc[3]);
        }
        const uint64_t s8 = ((uint64_t) w.s[1] << 32) | w.s[0];
        isum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int sc = (int) (int8_t) (s8 >> (8 * (v + 2 * k)));
            isum += __mul24(sc, acc[k] - 32 * x.gs[k]);      // sum (q-32)*a = sum q*a - 32*sum a   (|.| <= 16*63*127)
        }
        msum = 0;
        return fmaf(x.yd * h2f(w.d), (float) isum, facc);
    }
};

// ---------------------------------------------------------------- Q8_0 (row-SoA) --------------------
// row: qa[nb32][16] | qb[nb32][16] | d[nb32] (first / second 16 int8 of every block); activations quantized to Q8_0
// (32-blocks, fp16 d).   unit = one 32-block
template <> struct QT<PM_Q8_0> {
    static constexpr int NV = 32, LPB = 1, ABLK = 32;
    typedef XT<NV> X;
    struct Wr { u32x4 q0, q1; uint16_t d; };
    static __device__ __forceinline__ int group_base(int u, int g) { return u * 32 + 16 * g; }
    static __device__ __forceinline__ void issue(Wr & w, const uint8_t * row, int K, int u) {
        w.q0 = ld_nt16(row + (uint32_t) u * 16u);
        w.q1 = ld_nt16(row + (uint32_t) K / 2 + (uint32_t) u * 16u);
        w.d  = ld_nt2(row + (uint32_t) K + (uint32_t) u * 2u);
    }
    static __device__ __forceinline__ float consume(const Wr & w, const X & x, int, float acc, int & isum, int & msum) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { s = dot4(w.q0[i], x.q[i], s); s = dot4(w.q1[i], x.q[4 + i], s); }
        isum = s; msum = 0;
        return fmaf((float) s, h2f(w.d) * x.yd, acc);
    }
};

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + expf(-g)); }

#ifndef PM_GEMV_BLOCK
#define PM_GEMV_BLOCK 1024                // 16 waves: ONE workgroup per CU (measured +2.3 % over 2 x 512: prologue once per CU)
#endif
#define PM_GEMV_NW (PM_GEMV_BLOCK / 64)

// ---- activation prologue: the workgroup quantizes the WHOLE activation row once into LDS ----------------------------
//   xs_q  int8  [K]        quantized values
//   xs_gs int32 [K/16]     sums of each 16 consecutive quantized values (min / -32 terms)
//   xs_d  float [K/ABLK]   block scales (ABLK = 256: Q8_K, float d;  ABLK = 32: Q8_0, fp16-rounded d)
struct XLds { const int8_t * q; const int * gs; const float * d; int col_bytes; };   // column c of a multi-column launch: + c * col_bytes

// wave min, result in every lane (lanes that receive nothing from a DPP step keep their own value)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_keep_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_keepf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ int wave_min_keep(int v) {
    v = min(v, dpp_keep_i<0xB1>(v)); v = min(v, dpp_keep_i<0x4E>(v)); v = min(v, dpp_keep_i<0x141>(v)); v = min(v, dpp_keep_i<0x140>(v));
    v = min(v, dpp_keep_i<0x142, 0xA>(v)); v = min(v, dpp_keep_i<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// Row-of-16-lanes reductions (DPP quad_perm x2, row_half_mirror, row_mirror): every lane of the row gets the result.
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ __forceinline__ int row16_min(int v) {
    v = min(v, dpp_keep_i<0xB1>(v)); v = min(v, dpp_keep_i<0x4E>(v)); v = min(v, dpp_keep_i<0x141>(v)); v = min(v, dpp_keep_i<0x140>(v));
    return v;
}
// f64 wave sum through DPP on the two dword halves (no LDS traffic); total returned in every lane
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_d(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) u, CTRL, ROW_MASK, 0xF, true);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (u >> 32), CTRL, ROW_MASK, 0xF, true);
    return __builtin_bit_cast(double, ((uint64_t) hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_d<0xB1>(v); v += dpp_d<0x4E>(v); v += dpp_d<0x141>(v); v += dpp_d<0x140>(v);
    v += dpp_d<0x142, 0xA>(v); v += dpp_d<0x143, 0xC>(v);
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) u, 63), hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (u >> 32), 63);
    return __builtin_bit_cast(double, ((uint64_t) hi << 32) | lo);
}

// One ROW of 16 lanes = one 256-block (quantize_row_q8_K_ref): lane j holds v[k][i] = x[64k + 4j + i]. Four blocks per
// wave instruction stream: every workgroup quantizes the whole activation row redundantly, so the per-block instruction
// count (not bytes) is what the prologue costs - 16 lanes x 16 values needs 4 DPP steps per reduction and a quarter of
// the instructions of the one-block-per-wave form.
__device__ __forceinline__ float row16_min_f(float v) {
    v = fminf(v, dpp_keepf<0xB1>(v)); v = fminf(v, dpp_keepf<0x4E>(v)); v = fminf(v, dpp_keepf<0x141>(v)); v = fminf(v, dpp_keepf<0x140>(v));
    return v;
}
__device__ __forceinline__ float row16_max_f(float v) {
    v = fmaxf(v, dpp_keepf<0xB1>(v)); v = fmaxf(v, dpp_keepf<0x4E>(v)); v = fmaxf(v, dpp_keepf<0x141>(v)); v = fmaxf(v, dpp_keepf<0x140>(v));
    return v;
}
// Every workgroup repeats this for the whole activation row, on every CU: its VALU instruction count is wall time (seam anatomy: ~1.5 us
// of a 4.5 us prologue at K = 8192, ~3 us at K = 28672). Round 3: signed max / min instead of |x| + an index key per element (the index
// path only where +a and -a of equal magnitude tie for the maximum), v_rndne + v_cvt for nearest_int, v_perm byte packing, v_dot4 for the
// 16-value sums: ~11 -> ~7 instructions per value, same bits.
__device__ __forceinline__ void q8k_rows_to_lds(const float (&v)[4][4], int j, bool valid, int8_t * xs_q, int * xs_gs, float * xs_d, int blk) {
    float mx = v[0][0], mn = v[0][0];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) { mx = fmaxf(mx, v[k][i]); mn = fminf(mn, v[k][i]); }
    mx = row16_max_f(mx); mn = row16_min_f(mn);
    const float amax = fmaxf(mx, -mn);
    bool neg = -mn > mx;                             // the element with the largest |x| (quantize_row_q8_K_ref keeps its SIGNED value) ...
    if (mx == -mn && amax != 0.0f) {                 // ... and on a tie between +a and -a the FIRST of them: lowest index, low bit = its sign
        int key = 0x7fffffff;
#pragma unroll
        for (int k = 3; k >= 0; --k)
#pragma unroll
            for (int i = 3; i >= 0; --i)
                key = fabsf(v[k][i]) == amax ? (((64 * k + 4 * j + i) << 1) | (v[k][i] < 0.0f ? 1 : 0)) : key;
        key = row16_min(key);
        neg = key & 1;
    }
    const float iscale = amax != 0.0f ? -127.f / (neg ? -amax : amax) : 0.0f;
    uint32_t packed[4]; int psum[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (int) fminf(rintf(iscale * v[k][i]), 127.0f);      // nearest_int (RNE, |.| < 2^22), then MIN(127, .)
        const uint32_t p01 = __builtin_amdgcn_perm((uint32_t) q[1], (uint32_t) q[0], 0x0c0c0400u);
        const uint32_t p23 = __builtin_amdgcn_perm((uint32_t) q[3], (uint32_t) q[2], 0x0c0c0400u);
        packed[k] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
        psum[k] = dot4(packed[k], 0x01010101u, 0);
        psum[k] += dpp_i<0xB1>(psum[k]); psum[k] += dpp_i<0x4E>(psum[k]);       // 16 consecutive values = one quad of lanes
    }
    if (valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ((uint32_t *) xs_q)[blk * 64 + 16 * k + j] = packed[k];
            if ((j & 3) == 0) xs_gs[blk * 16 + 4 * k + (j >> 2)] = psum[k];
        }
        if (j == 0) xs_d[blk] = amax != 0.0f ? 1 / iscale : 0.0f;
    }
}

// 8 lanes x 4 values = one 32-block (quantize_row_q8_0_ref)
__device__ __forceinline__ void q80_block_to_lds(const float v[4], int i4 /*index of this float4 in the row*/,
                                                 int8_t * xs_q, int * xs_gs, float * xs_d) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = group8_max(amax);
    const float d = amax / 127;
    const float id = d ? 1.0f / d : 0.0f;
    uint32_t packed = 0; int psum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int q = (int) roundf(v[i] * id); psum += q; packed |= (uint32_t) (q & 0xFF) << (8 * i); }
    ((uint32_t *) xs_q)[i4] = packed;
    psum += dpp_i<0xB1>(psum); psum += dpp_i<0x4E>(psum);
    if ((i4 & 3) == 0) xs_gs[i4 >> 2] = psum;
    if ((i4 & 7) == 0) xs_d[i4 >> 3] = h2f(f2h(d));
}

// The prologue is split in two so that the kernel can put the first steps' WEIGHT loads in flight between them: the
// activation loads are issued first (they return first: VMEM returns in order), the weight loads queue up behind them
// and travel from HBM while the workgroup normalizes / quantizes the activation row.
//   ABLK = 256 (Q8_K): wave w, lane (r = lane / 16, j = lane % 16), pass t handles block 4 (w + 16 t) + r; f[t][k] is the
//                      float4 at element 64 k + 4 j of that block. Two passes (K <= 32768; K <= 16384 with norm weights)
//                      are loaded ONCE and stay in registers between the sum-of-squares and the quantization.
//   ABLK = 32  (Q8_0): thread t owns the float4s t, t + 1024, ... (8 lanes = one 32-block); K <= 16384 held in registers.
// f[0] / f[1]: the two passes of a row of up to 128 blocks; with norm weights (held for <= 64 blocks = one pass) f[1] carries the weights
// Kernel FEATURES that are compiled into their own instantiations (template parameter FEAT of gemv_body / gemv_q_kernel), because their mere presence
// costs the kernels that do not use them (measured, profiles/r05_ab_sumsq_q6k_tail.txt: the sum-of-squares code +0.55 us on every launch, unused):
//   PM_FEAT_SS    producer-side sum of squares: xmode 3 prologues and the ss_out epilogue
//   PM_FEAT_TAIL  attention in the tail of the wq | wk | wv launch (QkvEpi::att_out)
//   PM_FEAT_NEOX  logical -> matrix row mapping of the wq / wk jobs (NEOX rope pairs in the QKV epilogue: build_qwen2); the TAIL instantiations carry it too
constexpr int PM_FEAT_SS = 1, PM_FEAT_TAIL = 2, PM_FEAT_NEOX = 4;
template <bool SS> struct ActRegsT;
template <> struct ActRegsT<false> { float4 f[2][4]; };
template <> struct ActRegsT<true>  { float4 f[2][4]; double ssp[4]; };      // ssp: xmode 3, this lane's share of the producer's partial sums (n_ss <= 256)

template <int ABLK>
__device__ __forceinline__ bool act_held(const GemvP & p) {
    if (p.xmode == 0) return false;
    if (ABLK == 256) return p.K / 256 <= (p.xmode >= 2 ? 64 : 128);
    return p.K / 4 <= 4 * PM_GEMV_BLOCK;
}

template <int ABLK, bool COH, bool SS = false>
__device__ __forceinline__ void stage_issue(const GemvP & p, ActRegsT<SS> & a, int wave, int lane) {
    if constexpr (SS) if (p.xmode == 3) {    // the producer's partial sums: four 8-byte loads per lane, in front of everything else of this wave
#pragma unroll
        for (int i = 0; i < 4; ++i) a.ssp[i] = lane + 64 * i < p.n_ss ? ld_g(p.ss_in + lane + 64 * i) : 0.0;
    }
    if (!act_held<ABLK>(p)) return;
    const float4 * xf4 = (const float4 *) p.xf, * nw4 = (const float4 *) p.norm_w;
    if (ABLK == 256) {
        const int nblk = p.K / 256, r = lane >> 4, j = lane & 15;
#pragma unroll
        for (int t = 0; t < 2; ++t) if (64 * t < nblk) {
            const int B = min(4 * (wave + PM_GEMV_NW * t) + r, nblk - 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) a.f[t][k] = ld_act4<false>(xf4 + (B * 64 + 16 * k + j));
        }
        if (p.xmode >= 2) {
            const int B = min(4 * wave + r, nblk - 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) a.f[1][k] = ld_g(nw4 + (B * 64 + 16 * k + j));
        }
    } else {
        const int tid = threadIdx.x, n4 = p.K / 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * PM_GEMV_BLOCK;
            a.f[0][k] = ld_act4<false>(xf4 + (i < n4 ? i : 0));
            if (p.xmode >= 2) a.f[1][k] = ld_g(nw4 + (i < n4 ? i : 0));
        }
    }
}

__device__ __forceinline__ double sumsq4(const float4 & f) {
    double s = (double) (f.x * f.x); s += (double) (f.y * f.y); s += (double) (f.z * f.z); s += (double) (f.w * f.w);
    return s;
}

template <int ABLK, bool COH, bool SS = false>
__device__ __forceinline__ void stage_finish(const GemvP & p, ActRegsT<SS> & a, int8_t * xs_q, int * xs_gs, float * xs_d, double * nred,
                                             int wave, int lane, int ncols = 1, int col_bytes = 0, unsigned long long * t_norm = nullptr) {
    const int tid = threadIdx.x;
    const int K = p.K;
    if (p.xmode == 0) {
        // pre-quantized row-SoA: int8 qs[K] | scales. Copy + group sums, one LDS region per activation column.
        for (int c = 0; c < ncols; ++c) {
            const uint8_t * xq = p.xq + (size_t) (p.ncols > 0 && c >= p.ncols ? p.ncols - 1 : c) * p.xq_stride;   // (3 columns run the 4-column form: slot 3 repeats column 2, never stored)
            int8_t * cq = xs_q + c * col_bytes; int * cgs = (int *) ((char *) xs_gs + c * col_bytes); float * cd = (float *) ((char *) xs_d + c * col_bytes);
            for (int g = tid; g < K / 16; g += PM_GEMV_BLOCK) {
                const u32x4 t = *(const u32x4 *) (xq + 16 * g);
                *(u32x4 *) (cq + 16 * g) = t;
                int s_ = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) s_ = dot4(t[i], 0x01010101u, s_);
                cgs[g] = s_;
            }
            for (int b = tid; b < K / ABLK; b += PM_GEMV_BLOCK)
                cd[b] = ABLK == 256 ? ((const float *) (xq + K))[b] : h2f(((const uint16_t *) (xq + K))[b]);
        }
        return;
    }
    const float4 * xf4 = (const float4 *) p.xf, * nw4 = (const float4 *) p.norm_w;
    const int n4 = K / 4;
    const bool held = act_held<ABLK>(p);
    const int nblk = K / 256, r = lane >> 4, j = lane & 15;
    float scale = 1.0f;
    bool from_partials = false;
    if constexpr (SS) if (p.xmode == 3) {
        // every wave adds the producer's partials for itself: no reduction over the row, no LDS, no workgroup barrier
        const double tot = wave_sum_f64((a.ssp[0] + a.ssp[1]) + (a.ssp[2] + a.ssp[3]));
        if (t_norm) *t_norm = PM_TS_NOW();
        const float mean = (float) (tot / K);
        scale = 1.0f / sqrtf(mean + p.eps);
        from_partials = true;
    }
    if (!from_partials && p.xmode == 2) {
        // sum of the f32-rounded squares in f64 like the reference (ggml.c:11975-11980); any summation order of <= 2^15
        // f64 terms agrees with the sequential one after the final rounding to f32
        double ss = 0.0;
        if (held && ABLK == 256) {
            if (4 * wave + r < nblk) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ss += sumsq4(a.f[0][k]);
            }
        } else if (held) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (tid + k * PM_GEMV_BLOCK < n4) ss += sumsq4(a.f[0][k]);
        } else {
            for (int i = tid; i < n4; i += PM_GEMV_BLOCK) ss += sumsq4(ld_act4<false>(xf4 + i));
        }
        ss = wave_sum_f64(ss);
        if (lane == 0) nred[wave] = ss;
        __syncthreads();
        if (t_norm) *t_norm = PM_TS_NOW();
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < PM_GEMV_NW; ++k) tot += nred[k];
        const float mean = (float) (tot / K);
        scale = 1.0f / sqrtf(mean + p.eps);
    }
    if (ABLK == 256) {
        auto rows = [&](const float4 (&f)[4], const float4 (&g)[4], int B) __attribute__((always_inline)) {
            float v[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k][0] = f[k].x; v[k][1] = f[k].y; v[k][2] = f[k].z; v[k][3] = f[k].w;
                if (p.xmode >= 2) { v[k][0] = v[k][0] * scale * g[k].x; v[k][1] = v[k][1] * scale * g[k].y; v[k][2] = v[k][2] * scale * g[k].z; v[k][3] = v[k][3] * scale * g[k].w; }
            }
            q8k_rows_to_lds(v, j, B < nblk, xs_q, xs_gs, xs_d, B);
        };
        if (held) {
#pragma unroll
            for (int t = 0; t < 2; ++t) if (4 * (wave + PM_GEMV_NW * t) < nblk) rows(a.f[t], a.f[1], 4 * (wave + PM_GEMV_NW * t) + r);
        } else {
            for (int B0 = 4 * wave; B0 < nblk; B0 += 4 * PM_GEMV_NW) {
                const int Bc = min(B0 + r, nblk - 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) { a.f[0][k] = ld_act4<false>(xf4 + (Bc * 64 + 16 * k + j)); if (p.xmode >= 2) a.f[1][k] = ld_g(nw4 + (Bc * 64 + 16 * k + j)); }
                rows(a.f[0], a.f[1], B0 + r);
            }
        }
    } else {
        for (int i0 = tid; i0 < n4; i0 += 4 * PM_GEMV_BLOCK) {
            if (!held) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * PM_GEMV_BLOCK;
                    a.f[0][k] = ld_act4<false>(xf4 + (i < n4 ? i : 0));
                    if (p.xmode >= 2) a.f[1][k] = ld_g(nw4 + (i < n4 ? i : 0));
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * PM_GEMV_BLOCK;
                if (i - lane < n4) {                 // wave-uniform (n4 is a multiple of 8 = one 32-block)
                    const float4 f = a.f[0][k], g = a.f[1][k];
                    float v[4] = {f.x, f.y, f.z, f.w};
                    if (p.xmode >= 2) { v[0] = v[0] * scale * g.x; v[1] = v[1] * scale * g.y; v[2] = v[2] * scale * g.z; v[3] = v[3] * scale * g.w; }
                    if (i < n4) q80_block_to_lds(v, i, xs_q, xs_gs, xs_d);
                }
            }
        }
    }
}

template <int TYPE>
__device__ __forceinline__ void load_x_lds(typename QT<TYPE>::X & x, const XLds & xs, int u, int c = 0) {
    typedef QT<TYPE> T;
    const int8_t * q = xs.q + c * xs.col_bytes;
    const int * gs = (const int *) ((const char *) xs.gs + c * xs.col_bytes);
    const float * d = (const float *) ((const char *) xs.d + c * xs.col_bytes);
#pragma unroll
    for (int g = 0; g < T::NV / 16; ++g) {
        const int base = T::group_base(u, g);
        const u32x4 t = *(const u32x4 *) (q + base);
#pragma unroll
        for (int i = 0; i < 4; ++i) x.q[4 * g + i] = t[i];
        x.gs[g] = gs[base >> 4];
    }
    x.yd = d[T::group_base(u, 0) / T::ABLK];
}

// A wave processes ITEMS: R consecutive rows of one job. Lanes stride over the row's units in chunks of CH units per
// lane; the activation slice of every unit comes from LDS. No barrier and no cross-wave reduction inside an item.
// NX: the job's logical rows go through job_row() (NEOX rope in the QKV epilogue). A template parameter, not a runtime select: the dozen scalar
// instructions in front of every weight-row address cost the NORM-rope models 1.25 % of the 70B token when they sat in every kernel (round 4 shipped
// that; found in round 5 with the round-3 library beside it: profiles/r05_ab_sumsq_q6k_tail.txt, second correction)
template <int TYPE, bool PAIR, int NC = 1, bool NX = false> struct Item {
    typedef QT<TYPE> T;
    static constexpr int NM = PAIR ? 2 : 1;
#ifndef PM_CH32
#define PM_CH32 2
#endif
#ifndef PM_CH64
#define PM_CH64 1
#endif
#ifndef PM_RSINGLE
#define PM_RSINGLE 1
#endif
#ifndef PM_RPAIR
#define PM_RPAIR 1
#endif
    static constexpr int CH = T::NV == 64 ? PM_CH64 : PM_CH32;   // units per lane per row and register set (two sets in flight)
#ifndef PM_RCOLS
#define PM_RCOLS(NC_) 1
#endif
    // rows in flight per wave. Multi-column launches (mmvq_cols.hip): two rows share every activation slice fetched from LDS - the column loop is
    // bound by those LDS reads (4 columns: ffn_gate 51 -> 40 us, ffn_down Q6_K 52 -> 46 us; 8 columns x 2 rows spill at 256 VGPRs and keep 1)
    static constexpr int R  = PAIR ? PM_RPAIR : (NC > 1 ? (TYPE == PM_Q5_K && NC > 2 ? 1 : PM_RCOLS(NC)) : PM_RSINGLE);   // (Q5_K: 4 columns x 2 rows spill 200 B)
    struct Regs { typename T::Wr w[R][NM][CH]; };

    // unconditional loads (row / unit clamped): conditional loads make the compiler drain the VMEM queue
    static __device__ __forceinline__ void issue(Regs & g, const GemvP & p, const GemvJob & jb, int row, int r1, int c0, int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int rr = max(min(row + r, r1 - 1), 0);   // (a workgroup whose slice is empty still pre-issues: row 0 always exists)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int u = min(lane + 64 * (c0 + i), jb.U - 1);
                T::issue(g.w[r][0][i], jb.W + (long) (NX ? job_row(jb, rr) : rr) * jb.row_stride, p.K, u);
                if (PAIR) T::issue(g.w[r][NM - 1][i], jb.W2 + (long) rr * jb.row_stride, p.K, u);
            }
        }
    }
    template <bool DBG>
    static __device__ __forceinline__ void consume(const Regs & g, float (&acc)[R][NM][NC], const GemvP & p, const GemvJob & jb,
                                                   const XLds & xs, int row, int r1, int c0, int lane) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int uu = lane + 64 * (c0 + i);
            const bool uv = uu < jb.U;
            const int u = min(uu, jb.U - 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) {           // the nibble / scale decoding of T::consume is common to all columns (CSE)
                typename T::X x;
                load_x_lds<TYPE>(x, xs, u, c);
                x.yd = uv ? x.yd : 0.0f;             // a clamped (out-of-row) unit contributes exactly 0
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        int isum, msum;
                        acc[r][m][c] = T::consume(g.w[r][m][i], x, u, acc[r][m][c], isum, msum);
                        if (DBG) if (uv && row + r < r1) {
                            int32_t * o = p.dbg + ((long) (m * jb.N + row + r) * jb.U + u) * 2;
                            o[0] = isum; o[1] = msum;
                        }
                    }
            }
        }
    }
    // All items of ONE job that belong to this wave (item ids first, first+NW, ... < n_job_items), flattened into STEPS
    // (item, chunk) and software-pipelined with statically named register sets: the loads of step s+1 are in flight
    // while step s is consumed. The steady-state loop body is straight-line code - every issue() in it is unconditional -
    // so the compiler can use counted s_waitcnt vmcnt(N); the last one or two steps are peeled off behind the loop.
    // (All waves start in lock-step after the prologue barrier: without the overlap the whole chip would alternate
    //  between "only loading" and "only computing".)
    // NPRE (0, 1, 2): the kernel already issued the loads of this job's first NPRE steps into ga (, gb) before the activation prologue
    // (step 1 a clamped copy where the wave has one step only). [Round 3, measured and rejected: 3 or 4 pre-issued steps (12-VGPR Q4_K
    // sets fit next to the activation registers once the norm weights share the second pass's registers) - the extra 12-25 MB per launch sit
    // in every CU's load queue in front of the activation rows of the waves behind them: the prologue grows by what the extra steps cover
    // (QKV 4.5 -> 5.6-6.0 us, wo 2.3 -> 3.7-3.8 us, also when they are issued only after the wave's own activation row has arrived),
    // 116.0 vs 118.0 tok/s on the 70B shape, 588 vs 598 on the 8B shape.]
    // c_start / acc0 (decode_engine.hip): the first item starts at chunk c_start with the per-lane partial sums acc0[NM] of its chunks 0 .. c_start - 1,
    // which the caller consumed from another source (the engine's LDS-prefetched steps) with the same consume() calls: the fmaf chain of the row is
    // the one the plain call builds. A pre-issuing caller starts its cursor at (row of item `first`, c_start) too.
    template <bool DBG, int NPRE>
    static __device__ __forceinline__ void run_job(Regs & ga, Regs & gb, const GemvP & p, const GemvJob & jb, const XLds & xs,
                                                   float * out /*job slice*/, int first, int n_job_items, int r0, int r1, int lane,
                                                   int c_start = 0, const float * acc0 = nullptr) {
        static_assert(NPRE >= 0 && NPRE <= 2, "pre-issue depth");
        if (first >= n_job_items) return;
        const int upl = (jb.U + 63) >> 6;            // units per lane
        const int cpr = (upl + CH - 1) / CH;         // chunks (steps) per item
        const int n_my = (n_job_items - first + PM_GEMV_NW - 1) / PM_GEMV_NW;
        const int S = n_my * cpr - c_start;
        float acc[R][NM][NC];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[r][m][c] = (acc0 && r == 0 && c == 0) ? acc0[m] : 0.0f;
        // step cursors: (row, chunk) of the step being ISSUED and of the step being CONSUMED
        int irow = r0 + first * R, ic = c_start, crow = irow, cc = c_start;
        auto next = [&](int & row, int & c) __attribute__((always_inline)) { if (++c == cpr) { c = 0; row += PM_GEMV_NW * R; } };
        auto finish = [&]() __attribute__((always_inline)) {           // after a step was consumed: end of item?
            if (cc == cpr - 1) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        float o[NM];
#pragma unroll
                        for (int m = 0; m < NM; ++m) { o[m] = wave_sum(acc[r][m][c]); acc[r][m][c] = 0.0f; }
                        if (lane == 0 && crow + r < r1) out[(crow + r - r0) * NC + c] = PAIR ? silu_f(o[0]) * o[NM - 1] : o[0];
                    }
            }
            next(crow, cc);
        };
        auto eat = [&](Regs & g) __attribute__((always_inline)) { consume<DBG>(g, acc, p, jb, xs, crow, r1, cc * CH, lane); finish(); };
        auto put = [&](Regs & g) __attribute__((always_inline)) { issue(g, p, jb, irow, r1, ic * CH, lane); next(irow, ic); };
        {
            if (NPRE == 0) issue(ga, p, jb, irow, r1, ic * CH, lane);
            next(irow, ic);
        }
        int s_ = 0;
        if (NPRE >= 2) {                             // steps 0 AND 1 are already in flight (gb holds step 1, clamped if S == 1)
            if (S == 1) { eat(ga); return; }
            next(irow, ic);
            eat(ga);
            if (S == 2) { eat(gb); return; }
            put(ga);
            eat(gb);
            s_ = 2;
        }
        for (; s_ + 2 < S; s_ += 2) {
            put(gb);
            eat(ga);
            put(ga);
            eat(gb);
        }
        if (s_ + 1 < S) {
            issue(gb, p, jb, irow, r1, ic * CH, lane);
            eat(ga);
            eat(gb);
        } else {
            eat(ga);
        }
    }
    // SPLIT jobs: an item is ONE step (row, chunk); every step's partial sum goes to out[item] and write_out adds the chunks of a row in
    // chunk order. Used for the few rows of wk / wv per workgroup next to wq: (rows x chunks) items spread over all 16 waves instead of
    // whole rows that load only `rows` of them (a wave with a k or v row ran 6 steps where the others ran 4).
    template <bool DBG>
    static __device__ __forceinline__ void run_job_split(Regs & ga, Regs & gb, const GemvP & p, const GemvJob & jb, const XLds & xs, float * out,
                                                         int first, int n_items, int r0, int r1, int lane) {
        static_assert(R == 1 && !PAIR && NC == 1, "split jobs: single rows, one column");
        if (first >= n_items) return;
        const int upl = (jb.U + 63) >> 6, cpr = (upl + CH - 1) / CH;
        float acc[R][NM][NC];
        acc[0][0][0] = 0.0f;
        auto step = [&](Regs & g, int id) __attribute__((always_inline)) {
            const int row = r0 + id / cpr, c = id - (id / cpr) * cpr;
            consume<DBG>(g, acc, p, jb, xs, row, r1, c * CH, lane);
            const float o = wave_sum(acc[0][0][0]); acc[0][0][0] = 0.0f;
            if (lane == 0) out[id] = o;
        };
        int id = first;
        issue(ga, p, jb, r0 + id / cpr, r1, (id - (id / cpr) * cpr) * CH, lane);
        for (; id + 2 * PM_GEMV_NW < n_items; id += 2 * PM_GEMV_NW) {
            const int i1 = id + PM_GEMV_NW, i2 = id + 2 * PM_GEMV_NW;
            issue(gb, p, jb, r0 + i1 / cpr, r1, (i1 - (i1 / cpr) * cpr) * CH, lane);
            step(ga, id);
            issue(ga, p, jb, r0 + i2 / cpr, r1, (i2 - (i2 / cpr) * cpr) * CH, lane);
            step(gb, i1);
        }
        if (id + PM_GEMV_NW < n_items) {
            const int i1 = id + PM_GEMV_NW;
            issue(gb, p, jb, r0 + i1 / cpr, r1, (i1 - (i1 / cpr) * cpr) * CH, lane);
            step(ga, id);
            step(gb, i1);
        } else step(ga, id);
    }
};

// chunks (steps) per row of a job, as Item<>::run_job counts them
template <int TYPE> __device__ __forceinline__ int job_cpr(const GemvJob & jb) {
    constexpr int CH = QT<TYPE>::NV == 64 ? PM_CH64 : PM_CH32;
    return (((jb.U + 63) >> 6) + CH - 1) / CH;
}
// result of row `row` (index inside the workgroup's slice) of a job whose results start at outbuf[ob]
__device__ __forceinline__ float row_result(const GemvJob & jb, const float * outbuf, int ob, int row, int cpr) {
    if (!jb.split) return outbuf[ob + row];
    float o = outbuf[ob + row * cpr];
    for (int c = 1; c < cpr; ++c) o += outbuf[ob + row * cpr + c];           // chunk order: deterministic
    return o;
}

// returns this thread's share of sum (double)(out * out) over the values it stored (the producer-side partial of the next rms_norm, GemvP::ss_out)
template <bool COH, int NC, bool SS = false>
__device__ __forceinline__ double write_out(const GemvJob & jb, const float * outbuf, int r0, int r1, int ob, int tid, long y_stride, int cpr = 1, int ncols = NC) {
    double ss = 0.0;
    for (int t = tid; t < (r1 - r0) * NC; t += PM_GEMV_BLOCK) {
        const int c = NC == 1 ? 0 : t / (r1 - r0), row = NC == 1 ? t : t - c * (r1 - r0);     // consecutive threads -> consecutive rows
        if (NC > 1 && c >= ncols) break;                                                        // (column slots past the batch)
        float out = NC == 1 ? row_result(jb, outbuf, ob, row, cpr) : outbuf[(ob + row) * NC + c];
        if (jb.bias)  out += ld_g(jb.bias + r0 + row);
        if (jb.resid) out += ld_act<false>(jb.resid + c * y_stride + r0 + row);
        st_act<COH>(jb.y + c * y_stride + r0 + row, out);
        if constexpr (SS) {
            const float sq = out * out;                // f32-rounded square, then widened: (ggml_float)(x[i] * x[i]), ggml.c:11977
            ss += (double) sq;
        }
    }
    return ss;
}

// wq | wk | wv epilogue: RoPE on adjacent row pairs, q -> F16-rounded f32 in y, k -> F16 K-cache row of this token's cell, v -> F16 V cache
// (ggml_compute_forward_rope_f32 NORM mode ggml.c:14224-14237 with the per-token cos / sin table; llm_build_kv_store's CPY F32 -> F16,
// src/llama.cpp:9688-9716; the F16 rounding of q is the conversion ggml_compute_forward_mul_mat applies to src1 of the K.q product).
// (c, s_): this thread's (cos, sin) of pair `tid`, fetched by qkv_cs() BEFORE the barrier in front of the epilogue (a workgroup's slice
// holds at most PM_MAX_ROWS_PER_WG / 2 pairs < PM_GEMV_BLOCK: one pair per thread) - loaded after the barrier the table's L2 latency
// sat in the tail of every workgroup (store phase 1.6 us instead of 0.5).
template <bool NX = true>      // NX = false: the NEOX pairing is compiled out (NORM-rope launches: gemv_body's FEAT)
__device__ __forceinline__ void qkv_cs(const GemvJob & jb, const QkvEpi & e, int r0, int r1, int tid, float & c, float & s_) {
    c = 1.0f; s_ = 0.0f;
    if (jb.role == 1 || jb.role == 2) {
        if (NX && jb.nx_s) {                                           // NEOX: pair `tid` of the slice = matrix rows (p0, p0 + n_rot / 2), table entry p0 % dh
            const int ic = job_row(jb, r0 + tid) & (e.dh - 1);
            if (2 * tid < r1 - r0) { c = ld_g(e.tab + 2 * ic); s_ = ld_g(e.tab + 2 * ic + 1); }
        } else {
            const int d = (r0 + 2 * tid) % e.dh;
            if (2 * tid < r1 - r0 && d < e.n_rot) { c = ld_g(e.tab + d); s_ = ld_g(e.tab + d + 1); }
        }
    }
}
template <bool COH = false, bool NX = true>
__device__ __forceinline__ void write_out_qkv(const GemvJob & jb, const QkvEpi & e, const float * outbuf, int r0, int r1, int ob, int tid, int cpr,
                                              int slot, long kv_off, float c, float s_) {
    const int np = (r1 - r0) >> 1;
    {
        const int pr = tid;
        if (pr >= np) return;
        if (NX && jb.nx_s) {
            // NEOX (ggml_compute_forward_rope_f32, ggml.c:14238-14253): x0 = row p0, x1 = row p0 + n_rot / 2 of the same head; the slice holds both
            // (logical rows pr and pr + np). n_rot == head_dim here (gemv_fill): every row of a q / k head is rotated.
            const int p0 = job_row(jb, r0 + pr), p1 = p0 + jb.nx_hrot;
            float o0 = row_result(jb, outbuf, ob, pr, cpr), o1 = row_result(jb, outbuf, ob, pr + np, cpr);
            if (jb.bias) { o0 += ld_g(jb.bias + p0); o1 += ld_g(jb.bias + p1); }
            const float x0 = o0, x1 = o1;
            o0 = x0 * c - x1 * s_; o1 = x0 * s_ + x1 * c;
            const uint16_t h0 = f2h(o0), h1 = f2h(o1);
            if (jb.role == 2) { st_any<COH>(e.kc + kv_off + (long) slot * e.kv_dim + p0, h0); st_any<COH>(e.kc + kv_off + (long) slot * e.kv_dim + p1, h1); }
            else { st_any<COH>(jb.y + p0, h2f(h0)); st_any<COH>(jb.y + p1, h2f(h1)); }
            return;
        }
        const int row = r0 + 2 * pr;
        float o0 = row_result(jb, outbuf, ob, 2 * pr, cpr), o1 = row_result(jb, outbuf, ob, 2 * pr + 1, cpr);
        if (jb.bias) { o0 += ld_g(jb.bias + row); o1 += ld_g(jb.bias + row + 1); }
        if (jb.role == 3) {
            const uint16_t h0 = f2h(o0), h1 = f2h(o1);
            if (e.v_rowmajor) st_any<COH>((uint32_t *) (e.vc + kv_off + (long) slot * e.kv_dim + row), (uint32_t) h0 | ((uint32_t) h1 << 16));
            else { st_any<COH>(e.vc + kv_off + (long) row * e.n_ctx + slot, h0); st_any<COH>(e.vc + kv_off + (long) (row + 1) * e.n_ctx + slot, h1); }
            return;
        }
        if (row % e.dh < e.n_rot) {                                // (even row: pair (row % dh) / 2 of its head)
            const float x0 = o0, x1 = o1;
            o0 = x0 * c - x1 * s_; o1 = x0 * s_ + x1 * c;
        }
        const uint16_t h0 = f2h(o0), h1 = f2h(o1);
        if (jb.role == 2) st_any<COH>((uint32_t *) (e.kc + kv_off + (long) slot * e.kv_dim + row), (uint32_t) h0 | ((uint32_t) h1 << 16));
        else { st_any<COH>(jb.y + row, h2f(h0)); st_any<COH>(jb.y + row + 1, h2f(h1)); }
    }
}

// Attention in the tail of the wq | wk | wv launch (QkvEpi::att_out). The workgroups [g * wgs, (g + 1) * wgs) hold every row of KV-head group g - its
// H / Hkv query heads, its K rows, its V rows - so the group's attention depends on THESE workgroups only: the one seam of the layer that is not
// all-to-all. After its stores are out every workgroup takes a ticket of its group; the last H / Hkv arrivers wait until the group is complete (the
// rest of the chip is not waited for) and compute one query head each on four waves; everybody else leaves.
// The counters are monotonic (every launch adds exactly wgs per group, launches of a stream are serial): ticket mod wgs = arrival order in this launch.
template <int DUMMY = 0>
__device__ __forceinline__ void qkv_attention_tail(const GemvP & p, char * smem, int b, int tid, int wave, int lane) {
    const QkvEpi & e = p.epi;
    const int wgs = e.att_wgs, g = b / wgs, nh = e.att_H / (e.kv_dim / e.dh);
    unsigned * lds_u = (unsigned *) smem;                  // [0] ticket, [1] four-wave barrier counter (the attention body starts 64 bytes further up)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores have been acknowledged
    __syncthreads();                                       // ... every wave's (and nobody reads the activation row any more)
    if (tid == 0) {
        lds_u[0] = __hip_atomic_fetch_add((PM_G unsigned *) (e.att_ticket + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_u[1] = 0;
    }
    __syncthreads();
    const unsigned t = lds_u[0], k = t & (unsigned) (wgs - 1);
    if ((int) k < wgs - nh || wave >= 4) return;
    if (tid == 0) {
        const unsigned want = t - k + (unsigned) wgs;      // the counter's value once the whole group has arrived
        int spins = 0;
        while ((int) (__hip_atomic_load((const PM_G unsigned *) (e.att_ticket + g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { if (e.att_err) __hip_atomic_store((PM_G int *) e.att_err, 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    unsigned gen = 0;
    auto bar4 = [&]() __attribute__((always_inline)) {     // barrier of the four surviving waves (the other twelve have left: no s_barrier)
        ++gen;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
            __hip_atomic_fetch_add(lds_u + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(lds_u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * gen) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    bar4();
    const AttnTailP a = {e.att_q, e.kc, e.vc, e.pos_ptr, e.seq_ptr, e.seq_stride, e.att_out, e.att_H, e.kv_dim / e.dh, e.n_ctx, e.att_scale};
    const int h = g * nh + ((int) k - (wgs - nh));
    if (e.dh == 128) attn_tail_head<128>(a, h, smem + 64, bar4);
    else attn_tail_head<64>(a, h, smem + 64, bar4);
}

// The whole mat-vec of one workgroup (body of gemv_q_kernel / gemv_q_cols_kernel).
template <int TA, int TB, bool PAIR, bool DBG, int NC = 1, bool EPI = false, int NPRE = 2, int FEAT = 0>
__device__ __forceinline__ void gemv_body(const GemvP & p, char * smem, double * nred) {
    constexpr bool SS = (FEAT & PM_FEAT_SS) != 0, TAIL = (FEAT & PM_FEAT_TAIL) != 0, NX = (FEAT & (PM_FEAT_NEOX | PM_FEAT_TAIL)) != 0;
    static_assert(!NX || EPI, "row mapping: wq | wk | wv launches only");
    static_assert(!TAIL || (EPI && !PAIR && NC == 1), "the attention tail belongs to the wq | wk | wv launch");
    constexpr bool MEGA = false;      // (outputs are plain stores: kernel boundaries do the cache maintenance)
    constexpr int ABLK = QT<TA>::ABLK;
    typedef Item<TA, PAIR, NC, NX> IA;
    typedef Item<TB, PAIR, NC, NX> IB;
    constexpr int R = IA::R;
    // LDS: NC activation columns, each [q int8 K | gs int32 K/16 | d float K/ABLK] (16-byte aligned pieces), then the results
    const int col_bytes = ((p.K + 15) & ~15) + (p.K / 16) * 4 + ((p.K / ABLK + 3) & ~3) * 4;
    int8_t * xs_q  = (int8_t *) smem;                                   // [K] (K % 32 == 0 -> 16-B aligned pieces)
    int *    xs_gs = (int *) (smem + ((p.K + 15) & ~15));               // [K/16]
    float *  xs_d  = (float *) (xs_gs + p.K / 16);                      // [K/ABLK]
    float *  outbuf = (float *) (smem + (size_t) NC * col_bytes);       // [sum of this workgroup's rows][NC]
    const int b = blockIdx.x, G = gridDim.x, tid = threadIdx.x, lane = tid & 63;
    // wave index as a SCALAR: everything derived from it (row numbers, row base pointers, loop counters) then lives in
    // SGPRs and the weight loads use the saddr + 32-bit-voffset form instead of 64-bit VALU address arithmetic
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // this workgroup's slice [r0, r1) of every job, and the ITEM list = concatenation of the jobs' R-row groups
    // this workgroup's slice [r0, r1) of every job (scalars, not arrays: a runtime-indexed array would live in scratch)
    const int r0_0 = (int) ((long) p.job[0].N * b / G), r1_0 = (int) ((long) p.job[0].N * (b + 1) / G);
    const int r0_1 = (int) ((long) p.job[1].N * b / G), r1_1 = (int) ((long) p.job[1].N * (b + 1) / G);
    const int r0_2 = (int) ((long) p.job[2].N * b / G), r1_2 = (int) ((long) p.job[2].N * (b + 1) / G);
    // split jobs (EPI launches: wk / wv): items = (row, chunk) steps, results = one partial per item
    const int cpr_1 = !EPI || !p.job[1].split ? 1 : (TA != TB && p.job[1].is_b ? job_cpr<TB>(p.job[1]) : job_cpr<TA>(p.job[1]));
    const int cpr_2 = !EPI || !p.job[2].split ? 1 : (TA != TB && p.job[2].is_b ? job_cpr<TB>(p.job[2]) : job_cpr<TA>(p.job[2]));
    const int ni_0 = (r1_0 - r0_0 + R - 1) / R;
    const int ni_1 = cpr_1 > 1 ? (r1_1 - r0_1) * cpr_1 : (r1_1 - r0_1 + R - 1) / R, ni_2 = cpr_2 > 1 ? (r1_2 - r0_2) * cpr_2 : (r1_2 - r0_2 + R - 1) / R;
    const int ob_1 = r1_0 - r0_0, ob_2 = ob_1 + (r1_1 - r0_1) * cpr_1;
    // QKV epilogue: this token's cache cell and slab, fetched through the scalar cache now, used after the rows
    int epi_slot = 0; long epi_off = 0;
    if (EPI) {
        if (p.epi.dyn) epi_slot = uniform_const_ptr(p.epi.dyn)[0];
        else {
            const int seq = p.epi.seq_ptr ? uniform_const_ptr(p.epi.seq_ptr)[0] : 0;
            epi_slot = uniform_const_ptr(p.epi.pos_ptr)[seq];
            epi_off = (long) seq * p.epi.seq_stride;
        }
    }

    // (1) activation row -> LDS (quantized, bit-exact with the reference quantizers).
    //     [Measured and rejected: pre-issuing the first item's 16-byte weight loads across the prologue (spills under the
    //      128-VGPR budget, -20 %); an L2 prefetch of that item with one dword per 128-B line (-7 % even when limited to
    //      the small wq/wo launches: the in-order VMEM return delays the prologue and the lines are fetched twice); and
    //      splitting the workgroup into 8 prologue waves + 8 waves that pre-issue their first step (-3 %, A/B on one box).]
    unsigned long long tsv[6] = {PM_TS_NOW(), 0, 0, 0, 0, 0};
    ActRegsT<SS> areg;
    stage_issue<ABLK, MEGA, SS>(p, areg, wave, lane);                             // activation loads go out first (they return first)
    typename IA::Regs g0, g1;                                   // job 0 is always of type TA (host side orders the jobs)
    {   // the first NPRE steps of this wave in job 0 (same cursor sequence as run_job: chunks of a row, then the wave's next item)
        const int cpr0 = (((p.job[0].U + 63) >> 6) + IA::CH - 1) / IA::CH;
        int prow = r0_0 + wave * R, pc = 0;
        auto adv = [&]() __attribute__((always_inline)) { if (++pc == cpr0) { pc = 0; prow += PM_GEMV_NW * R; } };
        IA::issue(g0, p, p.job[0], prow, r1_0, pc * IA::CH, lane); adv();
        if (NPRE >= 2) { IA::issue(g1, p, p.job[0], prow, r1_0, pc * IA::CH, lane); adv(); }
    }
    stage_finish<ABLK, MEGA, SS>(p, areg, xs_q, xs_gs, xs_d, nred, wave, lane, NC, col_bytes, &tsv[1]);
    __syncthreads();
    tsv[2] = PM_TS_NOW();
    const XLds xs = {xs_q, xs_gs, xs_d, col_bytes};
    // (2) rows. Items of the jobs are dealt to the waves round-robin, continuing across jobs (wave offset rotates) so that
    //     the small k / v slices do not all land on wave 0.
    const int w1 = (wave + PM_GEMV_NW - ni_0 % PM_GEMV_NW) % PM_GEMV_NW;            // first item id of this wave in job 1
    const int w2 = (wave + 2 * PM_GEMV_NW - (ni_0 + ni_1) % PM_GEMV_NW) % PM_GEMV_NW;
    IA::template run_job<DBG, NPRE>(g0, g1, p, p.job[0], xs, outbuf, wave, ni_0, r0_0, r1_0, lane);
    typename IB::Regs gB, gB1;
    if (ni_1 > 0) {
        if constexpr (EPI && !PAIR && NC == 1 && R == 1) {
            if (cpr_1 > 1) {
                if (TA != TB && p.job[1].is_b) IB::template run_job_split<DBG>(gB, gB1, p, p.job[1], xs, outbuf + ob_1, w1, ni_1, r0_1, r1_1, lane);
                else                           IA::template run_job_split<DBG>(g0, g1, p, p.job[1], xs, outbuf + ob_1, w1, ni_1, r0_1, r1_1, lane);
            }
        }
        if (cpr_1 == 1) {
            if (TA != TB && p.job[1].is_b) IB::template run_job<DBG, 0>(gB, gB1, p, p.job[1], xs, outbuf + ob_1 * NC, w1, ni_1, r0_1, r1_1, lane);
            else                           IA::template run_job<DBG, 0>(g0, g1, p, p.job[1], xs, outbuf + ob_1 * NC, w1, ni_1, r0_1, r1_1, lane);
        }
    }
    if (ni_2 > 0) {
        if constexpr (EPI && !PAIR && NC == 1 && R == 1) {
            if (cpr_2 > 1) {
                if (TA != TB && p.job[2].is_b) IB::template run_job_split<DBG>(gB, gB1, p, p.job[2], xs, outbuf + ob_2, w2, ni_2, r0_2, r1_2, lane);
                else                           IA::template run_job_split<DBG>(g0, g1, p, p.job[2], xs, outbuf + ob_2, w2, ni_2, r0_2, r1_2, lane);
            }
        }
        if (cpr_2 == 1) {
            if (TA != TB && p.job[2].is_b) IB::template run_job<DBG, 0>(gB, gB1, p, p.job[2], xs, outbuf + ob_2 * NC, w2, ni_2, r0_2, r1_2, lane);
            else                           IA::template run_job<DBG, 0>(g0, g1, p, p.job[2], xs, outbuf + ob_2 * NC, w2, ni_2, r0_2, r1_2, lane);
        }
    }
    tsv[3] = PM_TS_NOW();                              // (wave 0 has finished its rows)
    float ec0 = 1.0f, es0 = 0.0f, ec1 = 1.0f, es1 = 0.0f, ec2 = 1.0f, es2 = 0.0f;
    if (EPI && p.epi.tab) {
        qkv_cs<NX>(p.job[0], p.epi, r0_0, r1_0, tid, ec0, es0);
        qkv_cs<NX>(p.job[1], p.epi, r0_1, r1_1, tid, ec1, es1);
        qkv_cs<NX>(p.job[2], p.epi, r0_2, r1_2, tid, ec2, es2);
    }
    __syncthreads();
    tsv[4] = PM_TS_NOW();                              // (all 16 waves have)
    // (4) coalesced write-out (+bias, +residual)
    if (EPI && p.epi.tab) {
        if constexpr (TAIL) {
            // (this instantiation is launched only with att_out set) write-through stores: other compute units read q and this token's cell in this launch
            write_out_qkv<true>(p.job[0], p.epi, outbuf, r0_0, r1_0, 0, tid, 1, epi_slot, epi_off, ec0, es0);
            write_out_qkv<true>(p.job[1], p.epi, outbuf, r0_1, r1_1, ob_1, tid, cpr_1, epi_slot, epi_off, ec1, es1);
            write_out_qkv<true>(p.job[2], p.epi, outbuf, r0_2, r1_2, ob_2, tid, cpr_2, epi_slot, epi_off, ec2, es2);
            qkv_attention_tail(p, smem, b, tid, wave, lane);
        } else {
            write_out_qkv<false, NX>(p.job[0], p.epi, outbuf, r0_0, r1_0, 0, tid, 1, epi_slot, epi_off, ec0, es0);
            write_out_qkv<false, NX>(p.job[1], p.epi, outbuf, r0_1, r1_1, ob_1, tid, cpr_1, epi_slot, epi_off, ec1, es1);
            write_out_qkv<false, NX>(p.job[2], p.epi, outbuf, r0_2, r1_2, ob_2, tid, cpr_2, epi_slot, epi_off, ec2, es2);
        }
    } else {
        const int ncw = NC > 1 && p.ncols > 0 ? p.ncols : NC;
        const double ss = write_out<MEGA, NC, SS>(p.job[0], outbuf, r0_0, r1_0, 0, tid, p.y_stride, 1, ncw);
        write_out<MEGA, NC>(p.job[1], outbuf, r0_1, r1_1, ob_1, tid, p.y_stride, cpr_1, ncw);
        write_out<MEGA, NC>(p.job[2], outbuf, r0_2, r1_2, ob_2, tid, p.y_stride, cpr_2, ncw);
        if constexpr (SS && NC == 1 && !PAIR && !EPI) if (p.ss_out) {
            // this workgroup's partial of the consumer's rms_norm (single-job launches: gemv_fill): rows sit in threads 0 .. r1 - r0 - 1
            const double ws = wave_sum_f64(ss);
            if (r1_0 - r0_0 <= 64) {                   // (every 70B / 8B / 72B shape: 16-32 rows per workgroup -> wave 0 alone, no barrier)
                if (tid == 0) st_g(p.ss_out + b, ws);
            } else {
                __syncthreads();                       // (nred: the norm prologue's reads are long done, but not fenced by a barrier of THIS phase)
                if (lane == 0) nred[wave] = ws;
                __syncthreads();
                if (tid == 0) {
                    double tot = 0.0;
#pragma unroll
                    for (int k = 0; k < PM_GEMV_NW; ++k) tot += nred[k];
                    st_g(p.ss_out + b, tot);
                }
            }
        }
    }
    tsv[5] = PM_TS_NOW();
    pm_ts_store(p.ts, 1 | (PAIR ? 16 : 0) | (p.xmode << 8) | (p.K << 12), tsv);
}


// host side (mmvq.hip): validate a fused job list and fill the kernel argument block
int gemv_fill(const pm_gemv_fused & a, int grid_fixed, GemvP & p, int & ta, int & tb, bool & pair, size_t & lds, int & grid);

} // namespace pmv

// multi-column launch (mmvq_cols.hip): nc = 2 / 4 / 8 activation columns per pass over the weights
int pm_launch_gemv_cols(int type, const pmv::GemvP & p, int nc, bool pair, int grid, size_t lds, hipStream_t st);
