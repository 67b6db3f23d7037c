// attn_prefill.hip — causal multi-token (prefill) attention on the MFMA matrix cores of gfx950.
//
// Replaces, for n_tokens >= 16, the three nodes of llm_build_kqv without flash-attn (src/llama.cpp:10032-10110):
//   kq  = MUL_MAT(K F16, q)            ggml_compute_forward_mul_mat with vec_dot_type F16: q is converted to F16, f32 accumulate
//   p   = SOFT_MAX_EXT(kq, mask, scale) ggml_compute_forward_soft_max_f32 (ggml.c:13783): max, expf, sum, scale by 1/sum
//   kqv = MUL_MAT(V^T F16, p)          p converted to F16, f32 accumulate
// with the SAME rounding points (q and p rounded to F16, K / V F16, f32 accumulation) - only the f32 summation order
// differs. The per-token kernel (attn_decode_kernel) needs O(n_tokens^2) scalar work: 9.2 ms per layer for a 2048-token
// prompt on the 70B shape, more than all GEMMs of the layer together.
//
// Design. One workgroup = 128 consecutive query tokens of one head, 4 waves x 32 queries. Everything is computed
// TRANSPOSED so that a query is a COLUMN of the MFMA result: S^T = K Q^T (keys x queries) and O^T = V^T P^T. In the
// 32x32 accumulator layout a lane then holds 16 keys of ONE query (column = lane & 31), so the softmax row statistics are
// in-lane reductions plus one exchange with the partner lane (lane ^ 32), and P^T comes out of the accumulator in exactly
// the order the next MFMA wants its B operand (again with one partner exchange) - no LDS round trip, no LDS at all.
// K rows ([key][dh] F16) and V^T rows ([dh][key] F16, the reference's transposed V cache) are both "k-contiguous" MFMA A
// operands and are read straight from global memory (the 4 waves of a workgroup and the 8 query heads of a KV group share
// them through L1 / L2). Two passes over the keys, because p must be rounded to F16 AFTER the division by the full row
// sum, exactly like the reference: pass 1 = row max and sum (online), pass 2 = recompute S, p = exp(s - max) / sum -> F16,
// accumulate O. FLOPs: 3 x (n_q x n_kv x dh x 2) / 2 per head.
#include "pm355_device.h"
#include "pm355_layer_ops.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

struct PfP {
    const float * q; const uint16_t * kc; const uint16_t * vc; const int32_t * pos0_ptr; const int32_t * seq_ptr; long seq_stride;
    float * out; int T, H, Hkv, n_ctx; float scale;
    // ggml-graph mode (mask != nullptr; pm355_attn_prefill_masked): every query attends cells [0, n_kv) with the additive F32 KQ_mask
    // row of its token (0 / -inf; llama_set_inputs src/llama.cpp:17379-17420) instead of the causal rule "cell <= position"
    const float * mask; long mask_stride; int n_kv;
    // flash-attention graphs (FLASH_ATTN_EXT, llm_build_kqv src/llama.cpp:10075-10095): v_rowmajor = the V cache is [n_ctx][Hkv*DH] like K
    // (the V^T operand is then gathered with 2-byte loads: a lane needs 8 consecutive keys of ONE channel); mask_f16 = the mask rows are
    // F16. P is rounded to F16 here as well (it is an MFMA operand; the reference keeps it f32 and rounds its F16 accumulator instead).
    int v_rowmajor, mask_f16;
    _Float16 * out_h;                            // optional: the result as F16 [T][H*DH] (activations of the wo GEMM) instead of `out`
};

// key index (inside a 32-key tile) of accumulator register r in lane-half h: C[row = (r&3) + 8 (r>>2) + 4 h][col]
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// K and V tiles (32 keys) travel global -> registers -> LDS once per WORKGROUP and are double-buffered: the loads of tile j+1 are in
// flight while the four waves multiply tile j out of LDS (round 2; before, every wave loaded its own copy straight into MFMA operands,
// 213 + 80 registers = one wave per SIMD and nothing to hide the L2 latency behind: 8.3 us per key tile, 1.14 ms per layer for a
// 2048-token prompt of the 70B shape). Row strides are padded so that the fragment reads (lane = key / channel, 16 bytes) are
// conflict-free: K rows DH + 8 halves (68 dwords: lane i -> bank 4 i), V^T rows 40 halves (20 dwords: 16 lanes cover the 64 banks once).
// A row-major V cache (flash-attention graphs) is transposed on its way into LDS.
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(PfP p) {
    constexpr int KK = DH / 16;                  // k-steps of the S^T MFMAs
    constexpr int DT = DH / 32;                  // 32-row tiles of O^T
    constexpr int KS = DH + 8, VS = 40;          // LDS row strides (halves)
    constexpr int UPR = DH / 8;                  // 16-byte units per K row
    constexpr int NU = DH / 64;                  // units per thread and tile (K: 32 * UPR / 256; V^T: 4 * DH / 256)
    __shared__ __attribute__((aligned(16))) _Float16 ks[2][32 * KS];
    __shared__ __attribute__((aligned(16))) _Float16 vs[2][DH * VS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hf = lane >> 5;
    const int h = blockIdx.y, hk = h / (p.H / p.Hkv);
    const int seq = p.seq_ptr ? *p.seq_ptr : 0;
    const int pos0 = p.mask ? 0 : p.pos0_ptr[seq];
    const long krow = (long) p.Hkv * DH;
    const uint16_t * kc = p.kc + (long) seq * p.seq_stride + (long) hk * DH;          // K[key][Hkv*DH]
    const uint16_t * vc = p.vc + (long) seq * p.seq_stride + (p.v_rowmajor ? (long) hk * DH : (long) hk * DH * p.n_ctx); // V^T[hk*DH + e][n_ctx] / V[key][Hkv*DH]
    // causal mode: the LAST query blocks attend the most keys - they are dispatched first (longest-processing-time order) so that the
    // short ones fill the tail
    const int qb = p.mask ? (int) blockIdx.x : (int) (gridDim.x - 1 - blockIdx.x);
    const int tq0 = qb * 128 + wave * 32;                                              // first query token of this wave
    const bool wave_live = tq0 < p.T;
    const int tq = min(tq0 + col, p.T - 1);                                            // this lane's query (clamped)
    const int qpos = pos0 + tq;                                                        // attends keys 0 .. qpos
    const int last = pos0 + min(tq0 + 31, p.T - 1);                                    // wave-uniform causal limit
    const int n_tiles = !wave_live ? 0 : (p.mask ? (p.n_kv + 31) / 32 : last / 32 + 1);
    // the workgroup walks the tiles its LAST live wave needs (every wave takes part in the staging and the barriers)
    const int n_wg = p.mask ? (p.n_kv + 31) / 32 : (pos0 + min(qb * 128 + 127, p.T - 1)) / 32 + 1;
    const float * mrow = (p.mask && !p.mask_f16) ? p.mask + (long) tq * p.mask_stride : nullptr;
    const uint16_t * mrow16 = (p.mask && p.mask_f16) ? (const uint16_t *) p.mask + (long) tq * p.mask_stride : nullptr;

    // ---- staging: global -> registers (fetch) -> LDS (put)
    u32x4 kg[NU], vg[NU];
    auto fetch_k = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + 256 * i, key = min(j * 32 + u / UPR, p.n_ctx - 1), piece = u % UPR;
            kg[i] = *(const u32x4 *) (kc + (long) key * krow + 8 * piece);
        }
    };
    auto fetch_v = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + 256 * i;
            if (!p.v_rowmajor) { const int row = u >> 2, unit = u & 3; vg[i] = *(const u32x4 *) (vc + (long) row * p.n_ctx + j * 32 + 8 * unit); }
            else { const int key = min(j * 32 + u / UPR, p.n_ctx - 1), piece = u % UPR; vg[i] = *(const u32x4 *) (vc + (long) key * krow + 8 * piece); }
        }
    };
    auto put_k = [&](int b) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NU; ++i) { const int u = tid + 256 * i; *(u32x4 *) (ks[b] + (u / UPR) * KS + 8 * (u % UPR)) = kg[i]; }
    };
    auto put_v = [&](int b) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + 256 * i;
            if (!p.v_rowmajor) *(u32x4 *) (vs[b] + (u >> 2) * VS + 8 * (u & 3)) = vg[i];
            else {                                                       // V[key][8 piece .. +8] -> V^T rows 8 piece .. +8, column key
                const int key = u / UPR, piece = u % UPR;
                uint16_t * d = (uint16_t *) vs[b] + (8 * piece) * VS + key;
#pragma unroll
                for (int e = 0; e < 4; ++e) { d[(2 * e) * VS] = (uint16_t) (vg[i][e] & 0xFFFF); d[(2 * e + 1) * VS] = (uint16_t) (vg[i][e] >> 16); }
            }
        }
    };

    // Q^T as B operand: lane (col = query, hf) holds q[query][16 kk + 8 hf .. +8], rounded to F16 like the reference
    half8 qf[KK];
    {
        const float * qr = p.q + ((long) tq * p.H + h) * DH + 8 * hf;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const float4 a = *(const float4 *) (qr + 16 * kk), b = *(const float4 *) (qr + 16 * kk + 4);
            qf[kk] = half8{(_Float16) a.x, (_Float16) a.y, (_Float16) a.z, (_Float16) a.w, (_Float16) b.x, (_Float16) b.y, (_Float16) b.z, (_Float16) b.w};
        }
    }
    // S^T tile (32 keys x 32 queries) of key tile j out of LDS buffer b: A = K rows (lane: key = col, dh slice 16 kk + 8 hf)
    auto scores = [&](int j, int b, float (&s)[16]) __attribute__((always_inline)) {
        float16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const _Float16 * kr = ks[b] + col * KS + 8 * hf;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8 *) (kr + 16 * kk), qf[kk], acc, 0, 0, 0);
        if (mrow || mrow16) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {                            // registers 4g .. 4g+3 = keys j*32 + 8g + 4hf .. +4: one float4 of the mask row
                const int k0 = j * 32 + 8 * g + 4 * hf;
                float4 mv = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                if (k0 + 3 < p.n_kv) {
                    if (mrow) mv = *(const float4 *) (mrow + k0);
                    else { const u32x2 mh = *(const u32x2 *) (mrow16 + k0); mv = float4{h2f((uint16_t) (mh[0] & 0xFFFF)), h2f((uint16_t) (mh[0] >> 16)), h2f((uint16_t) (mh[1] & 0xFFFF)), h2f((uint16_t) (mh[1] >> 16))}; }
                }
                s[4 * g + 0] = acc[4 * g + 0] * p.scale + mv.x; s[4 * g + 1] = acc[4 * g + 1] * p.scale + mv.y;
                s[4 * g + 2] = acc[4 * g + 2] * p.scale + mv.z; s[4 * g + 3] = acc[4 * g + 3] * p.scale + mv.w;
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = j * 32 + acc_row(r, hf);
            s[r] = k <= qpos ? acc[r] * p.scale : -INFINITY;        // soft_max_ext: x * scale (+ causal mask = -inf)
        }
    };

    // ---- pass 1: row max and sum of exp (online within the lane, merged with the partner lane at the end)
    // exp = v_exp_f32 (__expf, ~1 ulp): the probabilities are rounded to F16 right after, and the reference's own SIMD builds use a
    // polynomial expf (ggml_v_expf, ggml.c:2370); libm's expf was ~30 VALU instructions x 33 per lane and key tile - most of the kernel
    float m = -INFINITY, l = 0.0f;
    fetch_k(0);
    put_k(0);
    __syncthreads();
    for (int j = 0; j < n_wg; ++j) {
        const bool more = j + 1 < n_wg;
        if (more) fetch_k(j + 1);
        if (j < n_tiles) {
            float s[16];
            scores(j, j & 1, s);
            float mt = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
            const float mn = fmaxf(m, mt);
            if (mn != -INFINITY) {
                float acc = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc += __expf(s[r] - mn);
                l = l * __expf(m - mn) + acc;     // m == -inf: l == 0 and expf(-inf) == 0
                m = mn;
            }
        }
        if (more) put_k((j + 1) & 1);
        __syncthreads();
    }
    {
        const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
        const float mn = fmaxf(m, mo);           // key 0 is always visible: mn is finite
        l = (m == -INFINITY ? 0.0f : l * expf(m - mn)) + (mo == -INFINITY ? 0.0f : lo * expf(mo - mn));
        m = mn;
    }
    const float inv = 1.0f / l;

    // ---- pass 2: p = exp(s - m) * inv -> F16;  O^T (dh x queries) += V^T tile (dh x keys) . P^T tile (keys x queries)
    float16v o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.0f;
    fetch_k(0); fetch_v(0);
    put_k(0); put_v(0);
    __syncthreads();
    for (int j = 0; j < n_wg; ++j) {
        const bool more = j + 1 < n_wg;
        if (more) { fetch_k(j + 1); fetch_v(j + 1); }
        if (j < n_tiles) {
            float s[16];
            scores(j, j & 1, s);
            half4 pk[4];                         // pk[g] = the 4 keys 8 g + 4 hf .. +4 of this query
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) pk[g][i] = (_Float16) (__expf(s[4 * g + i] - m) * inv);
            // B operand of k-step kk (keys 16 kk .. +16): lane half hf wants keys 16 kk + 8 hf .. +8 = two groups of 4, one of
            // which sits in the partner lane (lane ^ 32)
            half8 pf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half4 mine = hf ? pk[2 * kk + 1] : pk[2 * kk];         // keys this lane keeps
                const half4 send = hf ? pk[2 * kk] : pk[2 * kk + 1];         // keys the partner needs
                uint32_t s0 = ((const uint32_t *) &send)[0], s1 = ((const uint32_t *) &send)[1];
                s0 = (uint32_t) __shfl_xor((int) s0, 32); s1 = (uint32_t) __shfl_xor((int) s1, 32);
                half4 got;
                ((uint32_t *) &got)[0] = s0; ((uint32_t *) &got)[1] = s1;
                const half4 lo = hf ? got : mine, hi = hf ? mine : got;      // ascending key order
                pf[kk] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            const _Float16 * vb = vs[j & 1] + col * VS + 8 * hf;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8 *) (vb + (32 * d) * VS + 16 * kk), pf[kk], o[d], 0, 0, 0);
        }
        if (more) { put_k((j + 1) & 1); put_v((j + 1) & 1); }
        __syncthreads();
    }
    // ---- O^T[dh = 32 d + acc_row(r, hf)][query = col] -> out[t][h*DH + dh]
    if (wave_live && tq0 + col < p.T) {
        if (p.out_h) {
            _Float16 * orow = p.out_h + ((long) tq * p.H + h) * DH;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(half4 *) (orow + 32 * d + 8 * g + 4 * hf) = half4{(_Float16) o[d][4 * g], (_Float16) o[d][4 * g + 1], (_Float16) o[d][4 * g + 2], (_Float16) o[d][4 * g + 3]};
        } else {
        float * orow = p.out + ((long) tq * p.H + h) * DH;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4 *) (orow + 32 * d + 8 * g + 4 * hf) = float4{o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]};
        }
    }
}

} // namespace

// n_ctx % 32 == 0 and head_dim 64 / 128 only (callers fall back to the per-token kernel otherwise)
int pm_launch_attn_prefill(const float * q, const void * kc, const void * vc, const int32_t * pos0, const int32_t * seq,
                           long seq_stride, float * out, int n_tok, int H, int Hkv, int dh, int n_ctx, float scale, hipStream_t st,
                           const float * mask, long mask_stride, int n_kv, int v_rowmajor, int mask_f16, void * out_f16) {
    if ((dh != 64 && dh != 128) || n_ctx % 32 || n_tok < 1) return -1;
    if (mask && (n_kv < 1 || n_kv > n_ctx || n_kv % 4 || mask_stride % 4)) return -1;
    if ((v_rowmajor || mask_f16) && !mask) return -1;                 // the flash-attention form exists in ggml-graph mode only
    PfP p = {q, (const uint16_t *) kc, (const uint16_t *) vc, pos0, seq, seq_stride, out, n_tok, H, Hkv, n_ctx, scale, mask, mask_stride, n_kv,
             v_rowmajor ? 1 : 0, mask_f16 ? 1 : 0, (_Float16 *) out_f16};
    const dim3 grid((n_tok + 127) / 128, H);
    if (dh == 128) hipLaunchKernelGGL(attn_prefill_kernel<128>, grid, dim3(256), 0, st, p);
    else           hipLaunchKernelGGL(attn_prefill_kernel<64>, grid, dim3(256), 0, st, p);
    return 0;
}
