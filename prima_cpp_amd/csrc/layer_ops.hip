// layer_ops.hip — the non-GEMV ops of one transformer layer at decode (n_tokens small), gfx950.
//
// Reference semantics restated per kernel (paths relative to the reference repo):
//   embed_rows      ggml_compute_forward_get_rows_q / _f32   ggml/src/ggml.c:13288, :13414 (+ dequantize_row_*)
//   rope_kv_store   ggml_compute_forward_rope_f32 ggml.c:14143 (NORM + NEOX, freq_factors, YaRN) followed by
//                   llm_build_kv_store src/llama.cpp:9673 (K -> F16 rows, V -> F16 TRANSPOSED [channel][n_ctx])
//   attn_decode     llm_build_kqv src/llama.cpp:10062-10148: MUL_MAT(K f16, q->f16) ; SOFT_MAX_EXT(scale, causal mask) ;
//                   MUL_MAT(V^T f16, p->f16) with the reference's rounding points (q and p are rounded to F16
//                   before the dot: ggml.c:12445-12473 with vec_dot_type F16)
//   argmax          greedy sampler llama_sampler_greedy (src/llama-sampling.cpp:390-397): first maximum wins
//   add / silu_mul / scale / cpy  the elementwise nodes of build_llama (ggml.c:9002, :11581, :10077)
// All are HBM/latency-bound byte movers; positions (pos) are read from DEVICE memory so that a captured
// hipGraph can be replayed for every token without re-recording.
#include "pm355_device.h"
#include "pm355_kernels.h"
#include "pm355_layer_ops.h"

// ------------------------------------------------------------------------------------------------
// generic element dequantization from the HBM layouts (slow path: embedding row lookup only)
// ------------------------------------------------------------------------------------------------
__device__ float pm_dequant_elem(int type, const uint8_t * row, int K, int i) {
    switch (type) {
    case PM_F32: return ((const float *) row)[i];
    case PM_F16: return h2f(((const uint16_t *) row)[i]);
    case PM_Q8_0: {                       // row-SoA: qa[nb][16] | qb[nb][16] | half d[nb]   (repack.hip)
        const float d = h2f(((const uint16_t *) (row + K))[i >> 5]);
        const int r = i & 31;
        return (float) ((const int8_t *) row)[(r >> 4) * (K / 2) + 16 * (i >> 5) + (r & 15)] * d;
    }
    case PM_Q4_K:                         // row-SoA: qa[U][16] | qb[U][16] | hdr[nb][16], unit = (block, j)   (repack.hip)
    case PM_Q5_K: {                       // native 176-byte blocks
        const long nb = K / 256;
        const int b = i >> 8, e = i & 255, s = e >> 5, l = e & 31;     // sub-block s, element l
        const uint8_t * blk = row + (long) b * PM_BS_Q5_K;
        const uint32_t * h = (const uint32_t *) (type == PM_Q4_K ? row + nb * 128 + (long) b * 16 : blk);
        int sc, mn;
        k4_scale_min(h[1], h[2], h[3], s, sc, mn);
        const float d = h2f((uint16_t) (h[0] & 0xFFFF)), dmin = h2f((uint16_t) (h[0] >> 16));
        int q = type == PM_Q4_K ? row[(l >> 4) * nb * 64 + 16 * (4 * (long) b + (s >> 1)) + (l & 15)] : blk[48 + 32 * (s >> 1) + l];
        q = (s & 1) ? (q >> 4) : (q & 0xF);
        if (type == PM_Q5_K) q += ((blk[16 + l] >> s) & 1) << 4;
        const float ds = d * (float) sc, ms = dmin * (float) mn;
        return ds * (float) q - ms;
    }
    case PM_Q6_K: {                       // row-SoA: la[U][16] | lb[U][16] | qh[U][16] | sc[nb][16] | d[nb], unit = (block, half, 16-col slice)
        const long nb = K / 256;
        const int b = i >> 8, e = i & 255, hh = e >> 7, r = e & 127, k = r >> 5, l = r & 31;
        const uint8_t lq = row[(k & 1) * nb * 64 + 16 * (4 * (long) b + 2 * hh + (l >> 4)) + (l & 15)];
        const uint8_t hq = row[nb * 128 + (long) b * 64 + 32 * hh + l];
        const int q = (int) (((k & 2) ? (lq >> 4) : (lq & 0xF)) | (((hq >> (2 * k)) & 3) << 4)) - 32;
        const int sc = (int) (int8_t) row[pm_q6k_sc_off((uint32_t) nb, (uint32_t) b) + 8 * hh + 2 * k + (l >> 4)];
        const float d = h2f(*(const uint16_t *) (row + pm_q6k_d_off((uint32_t) nb, (uint32_t) b)));
        return d * (float) sc * (float) q;
    }
    }
    return 0.0f;
}

__global__ __launch_bounds__(256) void embed_rows_kernel(int type, const uint8_t * table, long row_bytes, int K,
                                                         const int32_t * tokens, int n_tok, float * out) {
    const int t = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tok || i >= K) return;
    out[(long) t * K + i] = pm_dequant_elem(type, table + (long) tokens[t] * row_bytes, K, i);
}

// ------------------------------------------------------------------------------------------------
// RoPE (+ KV store). One 64-thread workgroup per (head, token): lane = rotation pair index.
// theta is built by the reference's running product (theta *= theta_scale per pair) so that the f32
// rounding sequence is identical; cosf/sinf are the device libm (<= 2 ulp from glibc).
// ------------------------------------------------------------------------------------------------
#include "attn_device.h"

// q: [n_tok][H*dh] f32 (rotated in place into q_out), k: [n_tok][Hkv*dh], v: [n_tok][Hkv*dh]
// K cache: f16 [n_ctx][Hkv*dh]; V cache: f16 [Hkv*dh][n_ctx] (transposed, as in the reference without flash-attn)
__global__ __launch_bounds__(64) void rope_kv_store_kernel(const float * q, const float * k, const float * v,
                                                           float * q_out, float * k_out_f32,
                                                           uint16_t * kc, uint16_t * vc,
                                                           const int32_t * pos0_ptr, const int32_t * seq_ptr, long seq_stride,
                                                           const float * freq_factors,
                                                           int H, int Hkv, int dh, int n_ctx, RopeP r, int round_q) {
    const int head = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    const int seq = seq_ptr ? *seq_ptr : 0;              // multi-sequence: one KV slab and one position per sequence
    const int pos = pos0_ptr[seq] + t;
    kc += (long) seq * seq_stride; vc += (long) seq * seq_stride;
    const bool is_q = head < H;
    const int hh = is_q ? head : head - H;
    const float * src = is_q ? q + ((long) t * H + hh) * dh : k + ((long) t * Hkv + hh) * dh;
    const bool neox = r.mode & 2;
    const int half = r.n_dims / 2;
    for (int pair = lane; pair < dh / 2; pair += 64) {
        float o0, o1; int a, b;
        if (pair < half) {
            a = neox ? pair : 2 * pair;
            b = neox ? pair + half : 2 * pair + 1;
            float c, s;
            rope_cs(r, (float) pos, pair, freq_factors, c, s);
            const float x0 = src[a], x1 = src[b];
            o0 = x0 * c - x1 * s;
            o1 = x0 * s + x1 * c;
        } else {                         // dims beyond n_dims pass through
            a = r.n_dims + 2 * (pair - half); b = a + 1;
            o0 = src[a]; o1 = src[b];
        }
        if (is_q) {
            float * d = q_out + ((long) t * H + hh) * dh;
            if (round_q) { o0 = h2f(f2h(o0)); o1 = h2f(f2h(o1)); }
            d[a] = o0; d[b] = o1;
        } else {
            if (k_out_f32) { float * d = k_out_f32 + ((long) t * Hkv + hh) * dh; d[a] = o0; d[b] = o1; }
            uint16_t * d = kc + (long) pos * Hkv * dh + (long) hh * dh;
            d[a] = f2h(o0); d[b] = f2h(o1);
        }
    }
    if (!is_q) {                         // V: plain F32 -> F16, transposed store
        for (int e = lane; e < dh; e += 64) {
            const int c = hh * dh + e;
            vc[(long) c * n_ctx + pos] = f2h(v[((long) t * Hkv + hh) * dh + e]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// decode attention: one 256-thread workgroup per (query head, token).
//   LDS: q as f16-rounded floats [dh] | scores/probabilities [n_ctx]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_decode_kernel(const float * q, const uint16_t * kc, const uint16_t * vc,
                                                          const int32_t * pos0_ptr, const int32_t * seq_ptr, long seq_stride,
                                                          float * out, int H, int Hkv, int dh, int n_ctx, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float redf[8];
    __shared__ double redd[4];
    float * qs = (float *) smem;                 // [dh]
    float * sc = qs + dh;                        // [n_ctx]
    const int h = blockIdx.x, t = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = h / (H / Hkv);
    const int seq = seq_ptr ? *seq_ptr : 0;
    kc += (long) seq * seq_stride; vc += (long) seq * seq_stride;
    const int n_kv = pos0_ptr[seq] + t + 1;      // causal: keys 0..pos
    for (int e = tid; e < dh; e += 256) qs[e] = h2f(f2h(q[((long) t * H + h) * dh + e]));
    __syncthreads();
    // ---- scores
    float lmax = -INFINITY;
    for (int i = tid; i < n_kv; i += 256) {
        const uint16_t * kr = kc + (long) i * Hkv * dh + (long) hk * dh;
        float acc = 0.0f;
        for (int e0 = 0; e0 < dh; e0 += 32) {    // 4 loads in flight per step (dh % 32 == 0 for every head size served: 64, 128, 256; else the tail below)
            if (e0 + 32 <= dh) {
                u32x4 kk[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) kk[c] = *(const u32x4 *) (kr + e0 + 8 * c);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc += h2f((uint16_t) (kk[c][j] & 0xFFFF)) * qs[e0 + 8 * c + 2 * j];
                        acc += h2f((uint16_t) (kk[c][j] >> 16)) * qs[e0 + 8 * c + 2 * j + 1];
                    }
            } else {
                for (int e = e0; e < dh; e += 8) {
                    const u32x4 kk = *(const u32x4 *) (kr + e);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc += h2f((uint16_t) (kk[j] & 0xFFFF)) * qs[e + 2 * j];
                        acc += h2f((uint16_t) (kk[j] >> 16)) * qs[e + 2 * j + 1];
                    }
                }
            }
        }
        const float s = acc * scale;             // mask is 0 for visible keys
        sc[i] = s;
        lmax = fmaxf(lmax, s);
    }
    // block max
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if (lane == 0) redf[wave] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    // ---- exp and sum (f64 accumulate like the reference)
    double lsum = 0.0;
    for (int i = tid; i < n_kv; i += 256) {
        const float e = expf(sc[i] - mx);
        sc[i] = e;
        lsum += (double) e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
    if (lane == 0) redd[wave] = lsum;
    __syncthreads();
    const double tot = (redd[0] + redd[1]) + (redd[2] + redd[3]);
    const float inv = (float) (1.0 / tot);
    const int n_pad = (n_kv + 7) & ~7;
    for (int i = tid; i < n_pad; i += 256) sc[i] = i < n_kv ? h2f(f2h(sc[i] * inv)) : 0.0f;   // p rounded to F16
    __syncthreads();
    // ---- out[e] = sum_i V^T[hk*dh+e][i] * p[i]; wave w owns e = w, w+4, ...; 8 keys per lane per step
    // (8 output elements per pass: their V loads are in flight together - one dependent load per element made this phase a chain of
    //  dh / 4 L2 round trips, 22 us per layer for an 8-token batch of the 70B shape; same per-element summation order)
    for (int e0 = wave; e0 < dh; e0 += 32) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = lane * 8; i < n_pad; i += 512) {
            u32x4 vv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) vv[k] = *(const u32x4 *) (vc + (long) (hk * dh + min(e0 + 4 * k, dh - 1)) * n_ctx + i);
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = sc[i + j];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[k] += h2f((uint16_t) (vv[k][j] & 0xFFFF)) * p[2 * j];
                    acc[k] += h2f((uint16_t) (vv[k][j] >> 16)) * p[2 * j + 1];
                }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float a = wave_sum(acc[k]);
            if (lane == 0 && e0 + 4 * k < dh) out[((long) t * H + h) * dh + e0 + 4 * k] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// single-token decode attention with RoPE and the KV store fused in (one launch per layer instead of two,
// and no dependent load chains: every phase issues independent loads).
//   grid = H query heads, 256 threads. Each workgroup rotates its own q and (redundantly, 64 pairs) the k of its
//   KV head for the current position; the first query head of each KV group also writes the F16 K row / V column.
//   The current position's key/value are taken from LDS, never re-read from the cache (no inter-workgroup hazard).
//   PV: thread = (channel e, part); each thread streams its own V^T row with 16-B loads - no cross-lane reduction.
// Same rounding points as attn_decode_kernel / the reference (q, p -> F16; K, V F16; f32 accumulate).
// ------------------------------------------------------------------------------------------------
template <int DH, int VM = 0>
__global__ __launch_bounds__(256) void attn_rope_fused_kernel(AttnP a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float redf[8];
    __shared__ double redd[4];
    attn_rope_body<DH, false, VM>(a, blockIdx.x, smem, redf, redd);
}

// ------------------------------------------------------------------------------------------------
// argmax over n floats, first maximum wins. Single workgroup of 1024 threads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void argmax_kernel(const float * x, int n, int32_t * out_idx, float * out_val) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float best = -INFINITY; int idx = 0x7fffffff;
    auto take = [&](float v, int i) __attribute__((always_inline)) { if (v > best || (v == best && i < idx)) { best = v; idx = i; } };
    // latency-bound (one workgroup, ~0.5 MB): 4 independent 16-byte loads in flight per thread
    const bool vec = ((uintptr_t) x & 15) == 0;
    const int n4 = vec ? n / 4 : 0;
    int i4 = tid;
    for (; i4 + 3 * 1024 < n4; i4 += 4 * 1024) {
        float4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = ((const float4 *) x)[i4 + k * 1024];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = 4 * (i4 + k * 1024); take(t[k].x, i); take(t[k].y, i + 1); take(t[k].z, i + 2); take(t[k].w, i + 3); }
    }
    for (; i4 < n4; i4 += 1024) { const float4 t = ((const float4 *) x)[i4]; const int i = 4 * i4; take(t.x, i); take(t.y, i + 1); take(t.z, i + 2); take(t.w, i + 3); }
    for (int i = 4 * n4 + tid; i < n; i += 1024) take(x[i], i);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(idx, off);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        // nothing compared greater than -inf (every logit NaN or -inf): index 0, like llama_sampler_greedy's loop that starts from entry 0
        // (src/llama-sampling.cpp:390-397) - never an out-of-range id that the next step would use as an embedding row
        *out_idx = (idx < 0 || idx >= n) ? 0 : idx;
        if (out_val) *out_val = best;
    }
}

// ------------------------------------------------------------------------------------------------
// small elementwise kernels (node-equivalent fallbacks used by the ggml-backend plug-in)
// ------------------------------------------------------------------------------------------------
__global__ void add_kernel(const float * a, const float * b, float * y, long n, long nb) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i % nb];
}
__global__ void mul_kernel(const float * a, const float * b, float * y, long n, long nb) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] * b[i % nb];
}
__global__ void silu_mul_kernel(const float * g, const float * u, float * y, long n) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float s = g[i] / (1.0f + expf(-g[i])); y[i] = u ? s * u[i] : s; }
}
__global__ void scale_kernel(const float * a, float * y, float s, long n) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] * s;
}
// ctl = {seq, n_seq}; pos[seq] += by; then seq = (seq + rotate) % n_seq
__global__ void advance_kernel(int32_t * pos, int32_t * ctl, int by, int rotate) {
    const int s = ctl[0];
    pos[s] += by;
    if (rotate) ctl[0] = (s + rotate) % ctl[1];
}
__global__ void set_i32_kernel(int32_t * p, int v) { *p = v; }
__global__ void set_i32x2_kernel(int32_t * p, int a, int b) { p[0] = a; p[1] = b; }

// ---- launchers -------------------------------------------------------------------------------------
void pm_launch_embed(int type, const void * table, int K, const int32_t * tokens, int n_tok, float * out, hipStream_t st) {
    hipLaunchKernelGGL(embed_rows_kernel, dim3((K + 255) / 256, n_tok), dim3(256), 0, st,
                       type, (const uint8_t *) table, (long) pm_weight_row_stride(type, K), K, tokens, n_tok, out);
}

void pm_rope_params(pm_rope_cfg & c) {
    // host-side constants exactly as ggml_compute_forward_rope_f32 computes them (ggml.c:14196-14199)
    c.theta_scale = powf(c.freq_base, -2.0f / c.n_dims);
    auto corr_dim = [&](float n_rot) {
        return c.n_dims * logf(c.n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(c.freq_base));
    };
    c.corr0 = fmaxf(0, floorf(corr_dim(c.beta_fast)));
    c.corr1 = fminf((float) (c.n_dims - 1), ceilf(corr_dim(c.beta_slow)));
}

void pm_launch_rope_kv_store(const float * q, const float * k, const float * v, float * q_out, float * k_out_f32,
                             void * kc, void * vc, const int32_t * pos0, const int32_t * seq, long seq_stride,
                             const float * freq_factors,
                             int n_tok, int H, int Hkv, int dh, int n_ctx, const pm_rope_cfg & c, hipStream_t st, int round_q) {
    RopeP r;
    r.n_dims = c.n_dims; r.mode = c.mode; r.n_ctx_orig = c.n_ctx_orig; r.theta_scale = c.theta_scale;
    r.freq_scale = c.freq_scale; r.ext_factor = c.ext_factor; r.attn_factor = c.attn_factor; r.corr0 = c.corr0; r.corr1 = c.corr1;
    hipLaunchKernelGGL(rope_kv_store_kernel, dim3(H + Hkv, n_tok), dim3(64), 0, st,
                       q, k, v, q_out, k_out_f32, (uint16_t *) kc, (uint16_t *) vc, pos0, seq, seq_stride, freq_factors, H, Hkv, dh, n_ctx, r, round_q);
}

int pm_launch_attn_decode(const float * q, const void * kc, const void * vc, const int32_t * pos0,
                          const int32_t * seq, long seq_stride, float * out,
                          int n_tok, int H, int Hkv, int dh, int n_ctx, float scale, hipStream_t st) {
    const size_t lds = (size_t) (dh + n_ctx) * 4;
    if (lds > 150 * 1024 || dh % 8 || n_ctx % 8) return -1;
    if (lds > 48 * 1024) {
        static bool set[16] = {};
        const int dv = pm_cur_dev();
        if (!set[dv]) { (void) hipFuncSetAttribute((const void *) attn_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); set[dv] = true; }
    }
    hipLaunchKernelGGL(attn_decode_kernel, dim3(H, n_tok), dim3(256), lds, st,
                       q, (const uint16_t *) kc, (const uint16_t *) vc, pos0, seq, seq_stride, out, H, Hkv, dh, n_ctx, scale);
    return 0;
}

int pm_launch_attn_rope_fused(const float * q, const float * k, const float * v, void * kc, void * vc,
                               const int32_t * pos0, const int32_t * seq, long seq_stride, const float * freq_factors,
                               float * out, int H, int Hkv, int dh, int n_ctx, float scale, const pm_rope_cfg & c, hipStream_t st,
                               const int32_t * dyn, const void * mask, int max_keys, int v_rowmajor, int mask_f16) {
    if ((dh != 64 && dh != 128 && dh != 256) || n_ctx % 8) return -1;
    const size_t lds = (size_t) (4 * dh + (v_rowmajor ? 2048 : 256) + (max_keys > 0 ? ((max_keys + 7) & ~7) : n_ctx) + 8) * 4;
    if (lds > 150 * 1024) return -1;
    RopeP r;
    r.n_dims = c.n_dims; r.mode = c.mode; r.n_ctx_orig = c.n_ctx_orig; r.theta_scale = c.theta_scale;
    r.freq_scale = c.freq_scale; r.ext_factor = c.ext_factor; r.attn_factor = c.attn_factor; r.corr0 = c.corr0; r.corr1 = c.corr1;
    auto launch = [&](auto kern) {
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        AttnP a = {q, k, v, (uint16_t *) kc, (uint16_t *) vc, pos0, seq, seq_stride, freq_factors, out, H, Hkv, n_ctx, scale, r, dyn,
                   (const float *) mask, mask_f16, pm_ts_next_slot()};
        hipLaunchKernelGGL(kern, dim3(H), dim3(256), lds, st, a);
    };
    if (v_rowmajor) {
        if (dh == 64) launch(attn_rope_fused_kernel<64, 1>);
        else if (dh == 128) launch(attn_rope_fused_kernel<128, 1>);
        else launch(attn_rope_fused_kernel<256, 1>);
    } else if (dh == 64) launch(attn_rope_fused_kernel<64>);
    else if (dh == 128) launch(attn_rope_fused_kernel<128>);
    else launch(attn_rope_fused_kernel<256>);
    return 0;
}

void pm_launch_argmax(const float * x, int n, int32_t * idx, float * val, hipStream_t st) {
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, x, n, idx, val);
}
void pm_launch_add(const float * a, const float * b, float * y, long n, long nb, hipStream_t st) {
    hipLaunchKernelGGL(add_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, a, b, y, n, nb);
}
void pm_launch_mul(const float * a, const float * b, float * y, long n, long nb, hipStream_t st) {
    hipLaunchKernelGGL(mul_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, a, b, y, n, nb);
}
void pm_launch_silu_mul(const float * g, const float * u, float * y, long n, hipStream_t st) {
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, g, u, y, n);
}
void pm_launch_scale(const float * a, float * y, float s, long n, hipStream_t st) {
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, a, y, s, n);
}
void pm_launch_set_i32(int32_t * p, int v, hipStream_t st) {
    hipLaunchKernelGGL(set_i32_kernel, dim3(1), dim3(1), 0, st, p, v);
}
void pm_launch_set_i32x2(int32_t * p, int a, int b, hipStream_t st) {
    hipLaunchKernelGGL(set_i32x2_kernel, dim3(1), dim3(1), 0, st, p, a, b);
}
void pm_launch_advance(int32_t * pos, int32_t * ctl, int by, int rotate, hipStream_t st) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, pos, ctl, by, rotate);
}
