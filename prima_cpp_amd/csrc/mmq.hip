// mmq.hip — batched (prefill) quantized GEMM on the MFMA matrix cores of gfx950.
//
// Replaces ggml_compute_forward_mul_mat (ggml/src/ggml.c:12377) for n_tokens >= 16 (what the reference's CUDA plug-in
// serves with mul_mat_q / dequantize + cuBLAS, ggml-cuda/mmq.cuh:2583, convert.cu:190-279):
//     Y[t][n] = sum_k W[n][k] * X[t][k]           W: Q4_K / Q5_K / Q6_K / Q8_0 rows in the HBM layout of repack.hip
// Design: weights are read from HBM exactly once per 256-token tile, dequantized on the fly into an F16 LDS tile (packed-F16
// math for Q4_K / Q6_K, see convert_w_h; f32 math = the exact d*sc*q - dmin*m of dequantize_row_* rounded to F16 for the
// other types); the activations are converted F32 -> F16
// once per call and copied into a second LDS tile; 8 waves run v_mfma_f32_32x32x16_f16 with f32 accumulation (each wave a
// 64x64 output block = 2x2 MFMA tiles). Tile 128 (weight rows) x 256 (tokens) x 64 (k), two LDS buffers (software
// pipeline over k), LDS rows padded to 72 halfs against bank conflicts.
// This is the compute-bound regime (arithmetic intensity ~ n_tokens FLOP/B >> ridge): the roofline is the dense F16
// MFMA peak, not HBM. Numerics: like the reference's own GPU large-batch path the activations are NOT re-quantized to
// Q8_K here; result vs the CPU reference is within the reference's own backend tolerance (NMSE <= 5e-4,
// tests/test-backend-ops.cpp:1660) - measured ~1e-6, tests/test_gpu_ops.py.
#include "pm355_device.h"
#include "pm355_kernels.h"
#include <stdlib.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 256, BK = 64, LDS_STRIDE = BK + 8;       // weight rows x tokens x k per tile; halfs per LDS row

// 16 consecutive weights of one row (part = 0 / 1: first / second half of the 32-weight sub-block starting at k0, k0 % 32
// == 0) in two steps, so that the global loads of the NEXT k-step can be in flight while the matrix cores work on the
// current one: fetch() = the raw 16-byte pieces, convert() = the exact dequantize_row_* arithmetic in f32 on them.
struct RawW { u32x4 r[3]; uint32_t s; };

template <int TYPE>
__device__ __forceinline__ void fetch_w(const uint8_t * row, int K, int k0, int part, RawW & w) {
    if (TYPE == PM_Q8_0) {                                             // row-SoA: qa[nb][16] | qb[nb][16] | half d[nb]
        w.s = ((const uint16_t *) (row + K))[k0 >> 5];
        w.r[0] = *(const u32x4 *) (row + part * (K / 2) + (k0 >> 1));
        return;
    }
    const int b = k0 >> 8, s = (k0 & 255) >> 5;                         // super-block, 32-value sub-block
    if (TYPE == PM_Q4_K || TYPE == PM_Q5_K) {
        // Q4_K row-SoA: qa[U][16] | qb[U][16] | hdr[nb][16] (unit = block*4 + s/2);  Q5_K native 176-byte blocks
        const long nb4 = K / 256;
        const uint8_t * blk = row + (long) b * PM_BS_Q5_K;
        w.r[1] = *(const u32x4 *) (TYPE == PM_Q4_K ? row + nb4 * 128 + (long) b * 16 : blk);
        w.r[0] = *(const u32x4 *) (TYPE == PM_Q4_K ? row + part * nb4 * 64 + 16 * (4 * (long) b + (s >> 1)) : blk + 48 + 32 * (s >> 1) + 16 * part);
        if (TYPE == PM_Q5_K) w.r[2] = *(const u32x4 *) (blk + 16 + 16 * part);
        return;
    }
    // Q6_K row-SoA: la[U][16] | lb[U][16] | qh[U][16] | sc[nb][16] | d[nb], unit = 4 b + 2 hh + v;  sub-block s -> half
    // hh = s/4, quarter kq = s%4: its 32 ql bytes are the pieces v = 0, 1 (= part) of stream (kq & 1)
    const long nb = K / 256;
    const int hh = s >> 2, kq = s & 3;
    w.r[0] = *(const u32x4 *) (row + (kq & 1) * nb * 64 + 16 * (4 * (long) b + 2 * hh + part));
    w.r[1] = *(const u32x4 *) (row + nb * 128 + (long) b * 64 + 32 * hh + 16 * part);
    w.s = (uint32_t) ((const uint8_t *) (row + pm_q6k_sc_off((uint32_t) nb, (uint32_t) b) + 8 * hh + 2 * kq))[part] |
          ((uint32_t) *(const uint16_t *) (row + pm_q6k_d_off((uint32_t) nb, (uint32_t) b)) << 16);
}

template <int TYPE>
__device__ __forceinline__ void convert_w(const RawW & w, int k0, float (&o)[16]) {
    if (TYPE == PM_Q8_0) {
        const float d = h2f((uint16_t) w.s);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[4 * i + j] = (float) (int8_t) (w.r[0][i] >> (8 * j)) * d;
        return;
    }
    const int s = (k0 & 255) >> 5;
    if (TYPE == PM_Q4_K || TYPE == PM_Q5_K) {
        const u32x4 h = w.r[1];
        int sc, mn;
        k4_scale_min(h[1], h[2], h[3], s, sc, mn);
        const float ds = h2f((uint16_t) (h[0] & 0xFFFF)) * (float) sc, ms = h2f((uint16_t) (h[0] >> 16)) * (float) mn;
        const int sh = (s & 1) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int v = (w.r[0][i] >> (8 * j + sh)) & 0xF;
                if (TYPE == PM_Q5_K) v += ((w.r[2][i] >> (8 * j + s)) & 1) << 4;
                o[4 * i + j] = ds * (float) v - ms;
            }
        return;
    }
    const int kq = s & 3;
    const float dd = h2f((uint16_t) (w.s >> 16)) * (float) (int8_t) (w.s & 0xFF);
    const int lsh = (kq >> 1) * 4, hsh = 2 * kq;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v = (int) (((w.r[0][i] >> (8 * j + lsh)) & 0xF) | (((w.r[1][i] >> (8 * j + hsh)) & 3) << 4)) - 32;
            o[4 * i + j] = dd * (float) v;
        }
}

// Packed-F16 dequantization of the same 16 weights straight into the two half8 vectors the LDS tile wants (PM_GEMM_F16_DEQUANT,
// Q4_K and Q6_K): v_perm_b32 spreads four 4/6-bit values into two dwords of (0x6400 | q) = the F16 numbers 1024 + q, one
// packed subtract makes them q exactly, one packed fma applies d*sc (and -dmin*m) - 2 VALU instructions per weight instead of
// ~7 (the dequantization was as expensive as the MFMAs it feeds). d*sc and dmin*m are formed in f32 and rounded to F16, so a
// weight carries up to 2 F16 roundings instead of 1: far inside the F16-activation noise (NMSE vs the oracle stays ~1e-6).
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
#ifndef PM_GEMM_F16_DEQUANT
#define PM_GEMM_F16_DEQUANT 1
#endif
template <int TYPE>
__device__ __forceinline__ void convert_w_h(const RawW & w, int k0, half8 (&o)[2]) {
    const int s = (k0 & 255) >> 5;
    half2v mul, add;
    float sub;                                                           // 1024 (+32 for Q6_K)
    int sh = 0, lsh = 0, hsh = 0;
    if (TYPE == PM_Q4_K) {
        const u32x4 h = w.r[1];
        int sc, mn;
        k4_scale_min(h[1], h[2], h[3], s, sc, mn);
        const _Float16 ds = (_Float16) (h2f((uint16_t) (h[0] & 0xFFFF)) * (float) sc), ms = (_Float16) (h2f((uint16_t) (h[0] >> 16)) * (float) mn);
        mul = half2v{ds, ds}; add = half2v{(_Float16) -ms, (_Float16) -ms};
        sub = 1024.0f; sh = (s & 1) * 4;
    } else {
        const int kq = s & 3;
        const _Float16 dd = (_Float16) (h2f((uint16_t) (w.s >> 16)) * (float) (int8_t) (w.s & 0xFF));
        mul = half2v{dd, dd}; add = half2v{(_Float16) 0.0f, (_Float16) 0.0f};
        sub = 1056.0f; lsh = (kq >> 1) * 4; hsh = 2 * kq;
    }
    const half2v bias = half2v{(_Float16) -sub, (_Float16) -sub};
    uint32_t out[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v;
        if (TYPE == PM_Q4_K) v = (w.r[0][i] >> sh) & 0x0F0F0F0Fu;
        else                 v = ((w.r[0][i] >> lsh) & 0x0F0F0F0Fu) | (((w.r[1][i] >> hsh) & 0x03030303u) << 4);
        const uint32_t lo = __builtin_amdgcn_perm(0x64646464u, v, 0x05010400u);   // bytes [v0, 0x64, v1, 0x64] = F16 (1024 + v0, 1024 + v1)
        const uint32_t hi = __builtin_amdgcn_perm(0x64646464u, v, 0x07030602u);
        half2v a = __builtin_bit_cast(half2v, lo), b = __builtin_bit_cast(half2v, hi);
        a = (a + bias) * mul + add;                                      // (1024 + q) - 1024 is exact; then one packed multiply-add
        b = (b + bias) * mul + add;
        out[2 * i] = __builtin_bit_cast(uint32_t, a); out[2 * i + 1] = __builtin_bit_cast(uint32_t, b);
    }
    o[0] = __builtin_bit_cast(half8, u32x4{out[0], out[1], out[2], out[3]});
    o[1] = __builtin_bit_cast(half8, u32x4{out[4], out[5], out[6], out[7]});
}

// Q4_K: both 16-weight parts of one 32-weight sub-block share its 6-bit scale / min - decode them once (gemm_q_f16_kernel2 stages
// the two parts from the same thread)
__device__ __forceinline__ void convert_q4k_pair_h(const RawW & w0, const RawW & w1, int k0, half8 (&o)[4]) {
    const int s = (k0 & 255) >> 5;
    const u32x4 h = w0.r[1];
    int sc, mn;
    k4_scale_min(h[1], h[2], h[3], s, sc, mn);
    const _Float16 ds = (_Float16) (h2f((uint16_t) (h[0] & 0xFFFF)) * (float) sc), ms = (_Float16) (h2f((uint16_t) (h[0] >> 16)) * (float) mn);
    const half2v mul = half2v{ds, ds}, add = half2v{(_Float16) -ms, (_Float16) -ms};
    const half2v bias = half2v{(_Float16) -1024.0f, (_Float16) -1024.0f};
    const int sh = (s & 1) * 4;
    uint32_t out[16];
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t v = ((part ? w1.r[0][i] : w0.r[0][i]) >> sh) & 0x0F0F0F0Fu;
            const uint32_t lo = __builtin_amdgcn_perm(0x64646464u, v, 0x05010400u);
            const uint32_t hi = __builtin_amdgcn_perm(0x64646464u, v, 0x07030602u);
            half2v a = __builtin_bit_cast(half2v, lo), b = __builtin_bit_cast(half2v, hi);
            a = (a + bias) * mul + add;
            b = (b + bias) * mul + add;
            out[8 * part + 2 * i] = __builtin_bit_cast(uint32_t, a); out[8 * part + 2 * i + 1] = __builtin_bit_cast(uint32_t, b);
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = __builtin_bit_cast(half8, u32x4{out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]});
}

struct GemmP {
    const uint8_t * W; const _Float16 * Xh; float * Y; const float * bias; const float * resid;
    const float * silu_gate;                 // optional [T][N]: Y = silu(gate) * (W.x)   (the SiLU.mul of the FFN fused into the up projection)
    long row_stride; int K, N, T;             // N = weight rows of THIS launch (a launch may cover a row range of the matrix)
    long ldy;                                // elements between consecutive tokens of Y / Yh / resid / silu_gate
    _Float16 * Yh;                           // optional: the result goes here as F16 [T][N] (the next GEMM's activations) instead of Y
    int exp;                                 // ablation switches (measurements only, compiled in with -DPM_GEMM_ABLATE=1; results are wrong when set)
};
// Ablations of gemm_q_f16_kernel2 (PM355_EXTRA_FLAGS=-DPM_GEMM_ABLATE=1 build, PM355_GEMM_EXP=<bits>, tools/gemm_probe.py; numbers in
// DESIGN.md section 6 / profiles/r02_gemm_ablation.txt): 1 no dequantization arithmetic, 2 no global loads in the loop, 4 no barrier,
// 8 half of the fragment reads, 16 no staging writes, 32 no weight loads, 64 no activation loads.
#ifndef PM_GEMM_ABLATE
#define PM_GEMM_ABLATE 0
#endif

// f32 -> f16 (RNE, == GGML_FP32_TO_FP16) of the activation matrix, once per GEMM call instead of once per weight tile
__global__ __launch_bounds__(256) void cvt_f16_kernel(const float * __restrict__ x, _Float16 * __restrict__ y, long n8) {
    const long i = (long) blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = ((const float4 *) x)[2 * i], b = ((const float4 *) x)[2 * i + 1];
    half8 v = {(_Float16) a.x, (_Float16) a.y, (_Float16) a.z, (_Float16) a.w, (_Float16) b.x, (_Float16) b.y, (_Float16) b.z, (_Float16) b.w};
    ((half8 *) y)[i] = v;
}

// grid (ceil(N/128), ceil(T/256)), 512 threads = 8 waves (2 x 4), each a 64x64 output block: a dequantized weight tile
// serves 256 tokens, so the dequantization VALU work per MFMA is half of what a 128-token tile costs. Software pipeline
// over k-steps with two LDS buffers: the raw weight pieces and the f16 activations of step s+1 are loaded into registers
// before the MFMAs of step s and are dequantized / stored into the other LDS buffer after them; one barrier per step.
// (Measured and rejected: loading three steps ahead - no gain: per step the CU spends ~1000 cycles each on MFMA, on the
//  dequantization VALU work and on the 128 KB of LDS fragment reads, the global latency is not what limits it.)
template <int TYPE>
__global__ __launch_bounds__(512) void gemm_q_f16_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];      // [2][(BM + BN) * LDS_STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BM, t0 = blockIdx.y * BN;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 64;             // this wave's 64x64 block inside the tile
    float16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging: weights  thread -> (tile row = tid / 4, 32-k half = (tid / 2) % 2, 16-weight part = tid % 2)
    //          tokens   thread -> (tile row = tid / 2, 32-k half = tid % 2)
    const int arow = tid >> 2, ahalf = (tid >> 1) & 1, apart = tid & 1;
    const int brow = tid >> 1, bhalf = tid & 1;
    const uint8_t * wptr = p.W + (long) min(n0 + arow, p.N - 1) * p.row_stride;
    const _Float16 * xptr = p.Xh + (long) min(t0 + brow, p.T - 1) * p.K + 32 * bhalf;
    constexpr int BUF = (BM + BN) * LDS_STRIDE;
    // Two register sets A / B: while the MFMAs of step s run, the SAME straight-line block dequantizes step s+1 (loaded one
    // iteration earlier) into the other LDS buffer - VALU and matrix pipe of one wave overlap - and issues the loads of step
    // s+3. No conditionals in the steady state: steps past the end are clamped loads / a redundant store into the idle buffer.
    struct Regs { RawW w; half8 x[4]; };
    Regs RA, RB;
    const int nst = p.K / BK;
    auto fetch = [&](Regs & R, int st) __attribute__((always_inline)) {
        const int k0 = min(st, nst - 1) * BK;
        fetch_w<TYPE>(wptr, p.K, k0 + 32 * ahalf, apart, R.w);
#pragma unroll
        for (int i = 0; i < 4; ++i) R.x[i] = *(const half8 *) (xptr + k0 + 8 * i);
    };
    auto stage = [&](const Regs & R, int st) __attribute__((always_inline)) {
        _Float16 * buf = lds + (st & 1) * BUF;
        _Float16 * da = buf + arow * LDS_STRIDE + 32 * ahalf + 16 * apart;
        _Float16 * db = buf + (BM + brow) * LDS_STRIDE + 32 * bhalf;
        if (PM_GEMM_F16_DEQUANT && (TYPE == PM_Q4_K || TYPE == PM_Q6_K)) {
            half8 o[2];
            convert_w_h<TYPE>(R.w, min(st, nst - 1) * BK + 32 * ahalf, o);
            *(half8 *) da = o[0]; *(half8 *) (da + 8) = o[1];
        } else {
            float o[16];
            convert_w<TYPE>(R.w, min(st, nst - 1) * BK + 32 * ahalf, o);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                half8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (_Float16) o[8 * i + j];
                *(half8 *) (da + 8 * i) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *(half8 *) (db + 8 * i) = R.x[i];
    };
    // MFMA: A fragment lane l = A[row = l&31][k = 8*(l>>5) .. +8], B fragment = B[k][col = l&31]
    auto mma = [&](int st) __attribute__((always_inline)) {
        const _Float16 * As = lds + (st & 1) * BUF, * Bs = As + BM * LDS_STRIDE;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            half8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *(const half8 *) (As + (wm + 32 * i + (lane & 31)) * LDS_STRIDE + kk + 8 * (lane >> 5));
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *(const half8 *) (Bs + (wn + 32 * j + (lane & 31)) * LDS_STRIDE + kk + 8 * (lane >> 5));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    fetch(RA, 0);
    stage(RA, 0);
    fetch(RA, 1); fetch(RB, 2);
    __syncthreads();
    // scheduling hint for the block: the 16 MFMAs of a step are issued with ~8 VALU instructions of the dequantization
    // between them (left alone the compiler emits the MFMAs back to back - the wave then sits in the matrix pipe's issue
    // queue for 512 cycles - and only afterwards the VALU work)
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, PM_GEMM_F16_DEQUANT ? 4 : 8, 0);   // VALU
        }
    };
    for (int st = 0; st < nst; st += 2) {
        mma(st); stage(RA, st + 1); fetch(RA, st + 3);
        interleave();
        __syncthreads();
        if (st + 1 < nst) {
            mma(st + 1); stage(RB, st + 2); fetch(RB, st + 4);
            interleave();
            __syncthreads();
        }
    }
    // ---- epilogue: C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]; Y[t][n]: 4 consecutive n per float4
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + wn + 32 * j + (lane & 31);
            if (t >= p.T) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wm + 32 * i + 8 * g + 4 * (lane >> 5);
                if (n + 3 < p.N) {
                    float4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (p.bias)  { const float4 bb = *(const float4 *) (p.bias + n); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
                    if (p.resid) { const float4 rr = *(const float4 *) (p.resid + (long) t * p.ldy + n); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
                    if (p.silu_gate) {
                        const float4 g = *(const float4 *) (p.silu_gate + (long) t * p.ldy + n);
                        v.x *= g.x / (1.0f + expf(-g.x)); v.y *= g.y / (1.0f + expf(-g.y)); v.z *= g.z / (1.0f + expf(-g.z)); v.w *= g.w / (1.0f + expf(-g.w));
                    }
                    if (p.Yh) *(half4v *) (p.Yh + (long) t * p.ldy + n) = half4v{(_Float16) v.x, (_Float16) v.y, (_Float16) v.z, (_Float16) v.w};
                    else *(float4 *) (p.Y + (long) t * p.ldy + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.N) {
                        float v = acc[i][j][4 * g + e];
                        if (p.bias) v += p.bias[n + e];
                        if (p.resid) v += p.resid[(long) t * p.ldy + n + e];
                        if (p.silu_gate) { const float g = p.silu_gate[(long) t * p.ldy + n + e]; v *= g / (1.0f + expf(-g)); }
                        if (p.Yh) p.Yh[(long) t * p.ldy + n + e] = (_Float16) v; else p.Y[(long) t * p.ldy + n + e] = v;
                    }
                }
            }
        }
}


// ---- second generation: 256 x 256 x 64 tile, 8 waves (4 x 2) of 64 (weight rows) x 128 (tokens) each ---------------------------
// What the first kernel above loses (rocprofv3 --pmc on the ffn_gate shape, profiles/r02_prefill_pmc.txt: MfmaUtil 41 %, VALUBusy 28 %,
// LdsUtil 36 %, no bank conflicts, memory unit never stalled): nothing is saturated - the waves WAIT. Its ISA shows why: the compiler
// kept three fragment registers and emitted ds_read_b128 -> s_waitcnt lgkmcnt(0) -> v_mfma chains, i.e. one exposed LDS latency
// (~130 cycles) in front of most 32-cycle MFMAs. Here
//   * the A / B fragments of k-slice kk+1 are loaded into a SECOND register set while the MFMAs of slice kk run (explicit double
//     buffering; counted lgkmcnt waits), 
//   * a wave owns 2 x 4 MFMA tiles: 6 fragment reads feed 8 MFMAs (was 4 for 4), and a k-step is 32 MFMAs (1024 matrix-pipe
//     cycles) per barrier instead of 16,
//   * the weight tile is 256 rows: every dequantized weight still serves 256 tokens, the activation tile is read by twice as many rows.
// LDS: 2 buffers x (256 + 256) rows x 72 halfs = 147 KB (one workgroup per CU, 2 waves per SIMD); staging registers: one set, the
// loads of step s+2 are issued right after step s+1 was written to LDS (a step is ~1 us: enough for an HBM round trip).
constexpr int BM2 = 256, BN2 = 256;
#ifndef PM_GEMM2_VALU_PER_MFMA
#define PM_GEMM2_VALU_PER_MFMA 3
#endif

template <int TYPE>
__global__ __launch_bounds__(512) void gemm_q_f16_kernel2(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];      // [2][(BM2 + BN2) * LDS_STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BM2, t0 = blockIdx.y * BN2;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 128;             // this wave's 64 x 128 block inside the tile
    float16v acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // staging: weights  thread -> (tile row = tid / 2, 32-k half = tid % 2), both 16-weight parts of that half
    //          tokens   thread -> (tile row = tid / 2, 32-k half = tid % 2)
    const int arow = tid >> 1, ahalf = tid & 1;
    const uint8_t * wptr = p.W + (long) min(n0 + arow, p.N - 1) * p.row_stride;
    const _Float16 * xptr = p.Xh + (long) min(t0 + arow, p.T - 1) * p.K + 32 * ahalf;
    constexpr int BUF = (BM2 + BN2) * LDS_STRIDE;
    struct Regs { RawW w[2]; half8 x[4]; };
    Regs R;
    const int nst = p.K / BK;
    const int exp = PM_GEMM_ABLATE ? p.exp : 0;
    auto fetch = [&](int st) __attribute__((always_inline)) {
        const int k0 = min(st, nst - 1) * BK;
        if (exp & 2) return;
        if (!(exp & 32)) {
            if (TYPE == PM_Q4_K) {
                // Q4_K: the two threads of a row (lanes 2r, 2r+1 = 32-k halves = sub-blocks 2u, 2u+1) need the SAME two 16-byte pieces
                // (stream a / stream b of unit u, low / high nibbles) and the same block header: each lane loads ONE of them (its own
                // stream; lane 2r also the header) and the pair swaps through DPP at staging time - 1.5 weight-load instructions per
                // thread and k-step instead of 4
                const int b = k0 >> 8, u = (k0 & 255) >> 6;
                const long nb4 = p.K / 256;
                R.w[0].r[0] = *(const u32x4 *) (wptr + (long) ahalf * nb4 * 64 + 16 * (4 * (long) b + u));
                if (ahalf == 0) R.w[0].r[1] = *(const u32x4 *) (wptr + nb4 * 128 + (long) b * 16);
            } else if (TYPE == PM_Q6_K) {
                // Q6_K: the two threads of a row (32-k halves = sub-blocks 2 j, 2 j + 1) work inside ONE 128-weight half (b, hh) of the super-block for
                // two consecutive k-steps: what they need of it - 32 bytes of their own ql stream (la: sub-blocks 0, 2; lb: 1, 3: both nibbles),
                // the 32 qh bytes, 8 scales and d - is loaded ONCE per two steps, the shared pieces by one lane each (qh piece `ahalf`, the scale
                // word by lane 0, d by lane 1) and exchanged through DPP at staging time: 2 load instructions per thread and k-step instead of
                // the 8 of two fetch_w calls (two 16-byte pieces + a 1-byte scale + a 2-byte d each) - the loads, not the arithmetic, were what
                // held this instantiation at half the Q4_K rate (round 3: 433 vs 881 TFLOP/s).
                if ((k0 & 127) == 0) {
                    const int b = k0 >> 8, hh = (k0 & 255) >> 7;
                    const long nb = p.K / 256;
                    R.w[0].r[0] = *(const u32x4 *) (wptr + (long) ahalf * nb * 64 + 16 * (4 * (long) b + 2 * hh));
                    R.w[0].r[1] = *(const u32x4 *) (wptr + (long) ahalf * nb * 64 + 16 * (4 * (long) b + 2 * hh + 1));
                    R.w[0].r[2] = *(const u32x4 *) (wptr + nb * 128 + (long) b * 64 + 32 * hh + 16 * ahalf);
                    if (ahalf == 0) { const u32x2 s8 = *(const u32x2 *) (wptr + pm_q6k_sc_off((uint32_t) nb, (uint32_t) b) + 8 * hh); R.w[1].r[0][0] = s8[0]; R.w[1].r[0][1] = s8[1]; }
                    else R.w[1].r[0][0] = (uint32_t) *(const uint16_t *) (wptr + pm_q6k_d_off((uint32_t) nb, (uint32_t) b));
                }
            } else {
                fetch_w<TYPE>(wptr, p.K, k0 + 32 * ahalf, 0, R.w[0]);
                fetch_w<TYPE>(wptr, p.K, k0 + 32 * ahalf, 1, R.w[1]);
            }
        }
        if (!(exp & 64)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) R.x[i] = *(const half8 *) (xptr + k0 + 8 * i);
        }
    };
    auto stage = [&](int st) __attribute__((always_inline)) {
        _Float16 * buf = lds + (st & 1) * BUF;
        _Float16 * da = buf + arow * LDS_STRIDE + 32 * ahalf;
        _Float16 * db = buf + (BM2 + arow) * LDS_STRIDE + 32 * ahalf;
        const int k0 = min(st, nst - 1) * BK + 32 * ahalf;
        if (exp & 16) return;
        if (exp & 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *(half8 *) (da + 8 * q) = __builtin_bit_cast(half8, R.w[0].r[q % 3]);
        } else
        if (PM_GEMM_F16_DEQUANT && TYPE == PM_Q4_K) {
            half8 o[4];
            // pair exchange (quad_perm [1,0,3,2]): the neighbour's stream piece and, for the odd lane, the header
            RawW wa, wb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t own = R.w[0].r[0][q];
                const uint32_t oth = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) own, 0xB1, 0xF, 0xF, true);
                wa.r[0][q] = ahalf ? oth : own;              // stream a piece (16-weight part 0 of the sub-block)
                wb.r[0][q] = ahalf ? own : oth;              // stream b piece (part 1)
                const uint32_t hown = R.w[0].r[1][q];
                const uint32_t hoth = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) hown, 0xB1, 0xF, 0xF, true);
                wa.r[1][q] = ahalf ? hoth : hown;
            }
            convert_q4k_pair_h(wa, wb, k0, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) *(half8 *) (da + 8 * q) = o[q];
        } else
        if (PM_GEMM_F16_DEQUANT && TYPE == PM_Q6_K) {
            // pair exchange (quad_perm [1,0,3,2]): the neighbour's qh piece, the scale word (held by the even lane) and d (held by the odd lane)
            const int kq = (k0 >> 5) & 3;
            const uint32_t o0 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) R.w[1].r[0][0], 0xB1, 0xF, 0xF, true);
            const uint32_t o1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) R.w[1].r[0][1], 0xB1, 0xF, 0xF, true);
            const uint32_t sc_lo = ahalf ? o0 : R.w[1].r[0][0], sc_hi = ahalf ? o1 : R.w[1].r[0][1];
            const uint32_t dbits = ahalf ? R.w[1].r[0][0] : o0;
            const uint64_t sc8 = ((uint64_t) sc_hi << 32) | sc_lo;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                RawW w;
                w.r[0] = part ? R.w[0].r[1] : R.w[0].r[0];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t own = R.w[0].r[2][q];
                    const uint32_t oth = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) own, 0xB1, 0xF, 0xF, true);
                    w.r[1][q] = part == ahalf ? own : oth;
                }
                w.s = (uint32_t) ((sc8 >> (8 * (2 * kq + part))) & 0xFF) | (dbits << 16);
                half8 o[2];
                convert_w_h<TYPE>(w, k0, o);
                *(half8 *) (da + 16 * part) = o[0]; *(half8 *) (da + 16 * part + 8) = o[1];
            }
        } else
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            if (PM_GEMM_F16_DEQUANT && (TYPE == PM_Q4_K || TYPE == PM_Q6_K)) {
                half8 o[2];
                convert_w_h<TYPE>(R.w[part], k0, o);
                *(half8 *) (da + 16 * part) = o[0]; *(half8 *) (da + 16 * part + 8) = o[1];
            } else {
                float o[16];
                convert_w<TYPE>(R.w[part], k0, o);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    half8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (_Float16) o[8 * i + j];
                    *(half8 *) (da + 16 * part + 8 * i) = v;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *(half8 *) (db + 8 * i) = R.x[i];
    };
    struct Frag { half8 a[2], b[4]; };
    auto load_frag = [&](Frag & f, const _Float16 * As, const _Float16 * Bs, int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) f.a[i] = *(const half8 *) (As + (wm + 32 * i + (lane & 31)) * LDS_STRIDE + kk + 8 * (lane >> 5));
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = *(const half8 *) (Bs + (wn + 32 * j + (lane & 31)) * LDS_STRIDE + kk + 8 * (lane >> 5));
    };
    auto mma8 = [&](const Frag & f) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
    };
    fetch(0);
    stage(0);
    fetch(1);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const _Float16 * As = lds + (st & 1) * BUF, * Bs = As + BM2 * LDS_STRIDE;
        Frag f0, f1;
        load_frag(f0, As, Bs, 0);
        load_frag(f1, As, Bs, 16);
        mma8(f0);
        if (!(exp & 8)) load_frag(f0, As, Bs, 32);
        mma8(f1);
        if (!(exp & 8)) load_frag(f1, As, Bs, 48);
        mma8(f0);
        stage(st + 1);                                    // dequantize step st+1 (registers) into the other buffer ...
        mma8(f1);
        fetch(st + 2);                                    // ... and put the loads of step st+2 in flight
        // schedule: fragment reads of the next slice first, then MFMAs with a few VALU instructions of the dequantization between them
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);    // 6 DS reads
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, PM_GEMM2_VALU_PER_MFMA, 0);    // VALU in its shadow
            }
        }
        if (!(exp & 4)) __syncthreads();
    }
    // ---- epilogue: C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]; Y[t][n]: 4 consecutive n per float4
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + wn + 32 * j + (lane & 31);
            if (t >= p.T) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wm + 32 * i + 8 * g + 4 * (lane >> 5);
                if (n + 3 < p.N) {
                    float4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (p.bias)  { const float4 bb = *(const float4 *) (p.bias + n); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
                    if (p.resid) { const float4 rr = *(const float4 *) (p.resid + (long) t * p.ldy + n); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
                    if (p.silu_gate) {
                        const float4 gg = *(const float4 *) (p.silu_gate + (long) t * p.ldy + n);
                        v.x *= gg.x / (1.0f + expf(-gg.x)); v.y *= gg.y / (1.0f + expf(-gg.y)); v.z *= gg.z / (1.0f + expf(-gg.z)); v.w *= gg.w / (1.0f + expf(-gg.w));
                    }
                    if (p.Yh) *(half4v *) (p.Yh + (long) t * p.ldy + n) = half4v{(_Float16) v.x, (_Float16) v.y, (_Float16) v.z, (_Float16) v.w};
                    else *(float4 *) (p.Y + (long) t * p.ldy + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.N) {
                        float v = acc[i][j][4 * g + e];
                        if (p.bias) v += p.bias[n + e];
                        if (p.resid) v += p.resid[(long) t * p.ldy + n + e];
                        if (p.silu_gate) { const float gg = p.silu_gate[(long) t * p.ldy + n + e]; v *= gg / (1.0f + expf(-gg)); }
                        if (p.Yh) p.Yh[(long) t * p.ldy + n + e] = (_Float16) v; else p.Y[(long) t * p.ldy + n + e] = v;
                    }
                }
            }
        }
}

} // namespace

// f16 copy of the activations: one scratch buffer per DEVICE (a process may drive several GPUs through the plug-in), grown
// on demand; GEMM calls of one device are issued from one host thread (ggml-backend stream semantics)
static _Float16 * g_xh[16] = {};
static size_t g_xh_elems[16] = {};

int pm_launch_gemm_q(int type, const void * W, const float * X, float * Y, int K, int N, int T, const float * bias,
                     const float * resid, hipStream_t st) {
    return pm_launch_gemm_q_ex(type, W, X, Y, K, N, T, bias, resid, nullptr, 0, st);
}

// silu_gate: Y = silu(gate[t][n]) * (W.x + bias) (gate may alias Y). reuse_x != 0: the per-device f16 scratch already holds
// the f16 copy of THIS X (previous call on this stream with the same X, T, K): skip the conversion.
int pm_launch_gemm_q_ex(int type, const void * W, const float * X, float * Y, int K, int N, int T, const float * bias,
                        const float * resid, const float * silu_gate, int reuse_x, hipStream_t st) {
    return pm_launch_gemm_q_h(type, W, X, nullptr, Y, nullptr, K, N, T, bias, resid, silu_gate, reuse_x, st);
}

// F16 plumbing between the prefill kernels: x_f16 != null = the activations are already F16 [T][K] (written by the producing kernel:
// rms-norm, attention, the previous GEMM's epilogue) - no conversion pass, no scratch; y_f16 != null = the result is stored as F16 [T][N]
// (the rounding the next GEMM's conversion pass would apply) and Y is not written.
int pm_launch_gemm_q_h(int type, const void * W, const float * X, const void * x_f16, float * Y, void * y_f16, int K, int N, int T,
                       const float * bias, const float * resid, const float * silu_gate, int reuse_x, hipStream_t st) {
    if (K % 64 || (type != PM_Q8_0 && K % 256) || N % 4) return -2;
    if (type != PM_Q4_K && type != PM_Q5_K && type != PM_Q6_K && type != PM_Q8_0) return -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    const size_t need = (size_t) T * K;
    if (!x_f16 && need > g_xh_elems[dev]) {
        if (g_xh[dev]) { (void) hipDeviceSynchronize(); (void) hipFree(g_xh[dev]); }
        if (hipMalloc((void **) &g_xh[dev], need * 2) != hipSuccess) { g_xh[dev] = nullptr; g_xh_elems[dev] = 0; return -3; }
        g_xh_elems[dev] = need;
    }
    _Float16 * xh = x_f16 ? (_Float16 *) x_f16 : g_xh[dev];
    if (!reuse_x && !x_f16) hipLaunchKernelGGL(cvt_f16_kernel, dim3((unsigned) ((need / 8 + 255) / 256)), dim3(256), 0, st, X, xh, (long) (need / 8));
    static const int exp_sw = [] { const char * e = getenv("PM355_GEMM_EXP"); return e ? atoi(e) : 0; }();
    static const int force = [] { const char * e = getenv("PM355_GEMM_KERNEL"); return e ? atoi(e) : 0; }();
    // third generation (mmq_pf.hip: LDS-DMA for every byte, weights dequantized in registers) wherever it serves the shape; PM355_GEMM_KERNEL=1 / 2 keep the older kernels (A/B)
    if ((force == 0 || force == 3) && pm_gemm_pf_check(type, K, N, T) == 0) {
        const pm_gemm_pf_job job = {type, N, W, Y, y_f16, bias, resid, silu_gate, (long) N};
        return pm_launch_gemm_pf(&job, 1, xh, K, T, st);
    }
    const long rs = (long) pm_weight_row_stride(type, K);
    // a launch over weight rows [r0, r0 + n): every per-column pointer is advanced, the token stride stays N
    auto params = [&](int r0, int n) {
        GemmP p = {(const uint8_t *) W + (long) r0 * rs, xh, Y ? Y + r0 : nullptr, bias ? bias + r0 : nullptr, resid ? resid + r0 : nullptr,
                   silu_gate ? silu_gate + r0 : nullptr, rs, K, n, T, (long) N, y_f16 ? (_Float16 *) y_f16 + r0 : nullptr, exp_sw};
        return p;
    };
    auto launch2 = [&](int r0, int n) {               // 256 x 256 tiles
        const GemmP p = params(r0, n);
        const dim3 grid2((n + BM2 - 1) / BM2, (T + BN2 - 1) / BN2);
        const size_t lds2 = (size_t) 2 * (BM2 + BN2) * LDS_STRIDE * sizeof(_Float16);
        auto go2 = [&](auto kern) {
            pm_allow_big_lds((const void *) kern, lds2 > 48 * 1024 ? lds2 : 48 * 1024 + 1);
            hipLaunchKernelGGL(kern, grid2, dim3(512), lds2, st, p);
        };
        switch (type) {
            case PM_Q4_K: go2(gemm_q_f16_kernel2<PM_Q4_K>); break;
            case PM_Q5_K: go2(gemm_q_f16_kernel2<PM_Q5_K>); break;
            case PM_Q6_K: go2(gemm_q_f16_kernel2<PM_Q6_K>); break;
            default:      go2(gemm_q_f16_kernel2<PM_Q8_0>); break;
        }
    };
    auto launch1 = [&](int r0, int n) {               // 128 x 256 tiles
        const GemmP p = params(r0, n);
        const dim3 grid((n + BM - 1) / BM, (T + BN - 1) / BN);
        const size_t lds = (size_t) 2 * (BM + BN) * LDS_STRIDE * sizeof(_Float16);
        auto go = [&](auto kern) {
            pm_allow_big_lds((const void *) kern, lds > 48 * 1024 ? lds : 48 * 1024 + 1);
            hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p);
        };
        switch (type) {
            case PM_Q4_K: go(gemm_q_f16_kernel<PM_Q4_K>); break;
            case PM_Q5_K: go(gemm_q_f16_kernel<PM_Q5_K>); break;
            case PM_Q6_K: go(gemm_q_f16_kernel<PM_Q6_K>); break;
            default:      go(gemm_q_f16_kernel<PM_Q8_0>); break;
        }
    };
    // 256 x 256 tiles (gemm_q_f16_kernel2) when they still fill the chip; else the 128 x 256 kernel (more workgroups for small N)
    static const bool no_tail_split = [] { const char * e = getenv("PM355_GEMM_TAIL_SPLIT"); return e && e[0] == '0'; }();
    const int cus = pm_device_cus();
    const int tn = (N + BM2 - 1) / BM2, tt = (T + BN2 - 1) / BN2;
    const long wg2 = (long) tn * tt;
    const bool use2 = force == 2 || (force != 1 && 2 * wg2 >= cus);     // at least half the CUs get a 256 x 256 tile
    if (!use2) { launch1(0, N); return 0; }
    // Tail: one 256 x 256 workgroup occupies a CU, so tn * tt tiles take ceil(tiles / CUs) rounds and the last one may be mostly empty
    // (ffn_gate of the 70B shape at 2048 tokens: 896 tiles = 3.5 rounds, 12.5 % of the launch idle). The weight rows of the last partial
    // round go to the 128 x 256 kernel instead (twice the workgroups, ~0.62 of a big tile's time each): [0, N1) = whole rounds of big
    // tiles, [N1, N) = small tiles - when that is estimated to be shorter.
    const long full = wg2 / cus, rem = wg2 % cus;
    if (!no_tail_split && force != 2 && full >= 1 && rem != 0) {
        const int tn_full = (int) (full * cus / tt);                   // big-tile rows that fill `full` rounds (or slightly less)
        const int N1 = tn_full * BM2;
        if (tn_full >= 1 && N1 < N) {
            const long small = (long) ((N - N1 + BM - 1) / BM) * ((T + BN - 1) / BN);
            const double t_old = (double) (full + 1);
            const double t_new = (double) ((long) tn_full * tt + cus - 1) / cus + 0.62 * (double) ((small + cus - 1) / cus);
            if (t_new < t_old - 0.05) { launch2(0, N1); launch1(N1, N - N1); return 0; }
        }
    }
    launch2(0, N);
    return 0;
}

// Several matrices over the same activations: f32 X is converted to F16 once (x_f16 != null: already F16, e.g. written by the producing kernel);
// one launch of the third-generation kernel when it serves every job, else one launch per job.
int pm_launch_gemm_q_multi(const pm_gemm_pf_job * jobs, int njobs, const float * X, const void * x_f16, int K, int T, hipStream_t st) {
    if (njobs < 1 || njobs > 4 || !jobs || (!X && !x_f16)) return -2;
    static const int force = [] { const char * e = getenv("PM355_GEMM_KERNEL"); return e ? atoi(e) : 0; }();
    bool all_pf = force == 0 || force == 3;
    int nty = 0, ty[4];
    for (int j = 0; j < njobs; ++j) {
        all_pf = all_pf && pm_gemm_pf_check(jobs[j].type, K, jobs[j].N, T) == 0;
        bool seen = false;
        for (int q = 0; q < nty; ++q) seen = seen || ty[q] == jobs[j].type;
        if (!seen) ty[nty++] = jobs[j].type;
    }
    all_pf = all_pf && nty <= 2;
    if (nty == 2 && (ty[0] == PM_Q8_0 || ty[1] == PM_Q8_0)) all_pf = false;      // (Q8_0 goes as a launch of its own)
    int rc = 0;
    for (int j = 0; j < njobs; ++j) {
        if (all_pf && j > 0) break;
        const pm_gemm_pf_job & b = jobs[j];
        if (all_pf) {
            if (x_f16) return pm_launch_gemm_pf(jobs, njobs, x_f16, K, T, st);
            // (the single-job entry converts X into the per-device scratch and launches job 0; the others follow on the same scratch)
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
            const size_t need = (size_t) T * K;
            if (need > g_xh_elems[dev]) {
                if (g_xh[dev]) { (void) hipDeviceSynchronize(); (void) hipFree(g_xh[dev]); }
                if (hipMalloc((void **) &g_xh[dev], need * 2) != hipSuccess) { g_xh[dev] = nullptr; g_xh_elems[dev] = 0; return -3; }
                g_xh_elems[dev] = need;
            }
            hipLaunchKernelGGL(cvt_f16_kernel, dim3((unsigned) ((need / 8 + 255) / 256)), dim3(256), 0, st, X, g_xh[dev], (long) (need / 8));
            return pm_launch_gemm_pf(jobs, njobs, g_xh[dev], K, T, st);
        }
        rc |= pm_launch_gemm_q_h(b.type, b.W, X, x_f16, b.Y, b.Yh, K, b.N, T, b.bias, b.resid, b.silu_gate, j > 0 ? 1 : 0, st);
    }
    return rc;
}

int pm_launch_gemm_q_pair(const pm_gemm_pf_job * gate_up, const float * X, int K, int T, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
    const size_t need = (size_t) T * K;
    if (need > g_xh_elems[dev]) {
        if (g_xh[dev]) { (void) hipDeviceSynchronize(); (void) hipFree(g_xh[dev]); }
        if (hipMalloc((void **) &g_xh[dev], need * 2) != hipSuccess) { g_xh[dev] = nullptr; g_xh_elems[dev] = 0; return -3; }
        g_xh_elems[dev] = need;
    }
    hipLaunchKernelGGL(cvt_f16_kernel, dim3((unsigned) ((need / 8 + 255) / 256)), dim3(256), 0, st, X, g_xh[dev], (long) (need / 8));
    return pm_launch_gemm_pf_ex(gate_up, 2, g_xh[dev], K, T, 1, st);
}
