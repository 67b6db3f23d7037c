// attn_q8.hip — quantized (Q8_0) KV cache behind the plug-in: `-ctk q8_0 [-ctv q8_0] -fa` (llama_kv_cache_init src/llama.cpp:3531-3560;
// a quantized V cache requires flash attention, :19925-19928).
//
// The cache tensors are 1-D Q8_0 tensors in the NATIVE ggml block order (34-byte blocks: half d + 32 int8, ggml-common.h:187-191) -
// the row-SoA re-ordering of repack.hip applies to weight MATRICES only - so every byte offset the reference computes for its views
// (row_size(type, n_embd_k_gqa) * kv_head, ...) and for state save / restore stays valid. Blocks are 2-byte aligned: fields are
// read with 16-bit loads. Three pieces, each with the reference CPU path's arithmetic:
//   1. KV store   CPY(f32 -> Q8_0 view): quantize_row_q8_0_ref (ggml-quants.c:848-873): d = amax / 127 (stored F16), q = roundf(x / d)
//   2. flash-attention node (multi-token batches), K and / or V quantized: ggml_compute_forward_flash_attn_ext_f16 (ggml.c:15870-16050):
//      K Q8_0 -> the query row is quantized to Q8_0 as well (q_to_vec_dot = vec_dot_type of K) and s = sum over blocks of
//      (int8 . int8) * (d_k * d_q) (ggml_vec_dot_q8_0_q8_0, ggml-quants.c:5518); V Q8_0 -> rows are dequantized (q * d) and
//      accumulated in f32 (the reference's VKQ32 path). The softmax is two-pass here, online there.
//   3. the same for ONE token fused with RoPE and the KV store in one launch per layer (what the plug-in's graph lowering uses at
//      decode), one workgroup per query head.
#include "../../include/prima_mi355.h"
#include "attn_device.h"
#include "pm355_kernels.h"
#include "pm355_layer_ops.h"

namespace {

constexpr int QB = 34;                           // bytes per block_q8_0

__device__ __forceinline__ float q8_d(const uint8_t * blk) { return h2f(*(const PM_G uint16_t *) blk); }
// int8 values 2j, 2j+1 of a block as floats
__device__ __forceinline__ void q8_pair(const uint8_t * blk, int j, float & a, float & b) {
    const uint16_t h = *(const PM_G uint16_t *) (blk + 2 + 2 * j);
    a = (float) (int8_t) (h & 0xFF); b = (float) (int8_t) (h >> 8);
}
// one value per lane, 32 consecutive lanes = one block: returns the int8 value; d16 = the block scale as stored (F16 bits)
__device__ __forceinline__ int q8_quant32(float x, uint16_t & d16) {
    float a = fabsf(x);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) a = fmaxf(a, __shfl_xor(a, off));
    const float d = a / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    d16 = f2h(d);
    return (int) roundf(x * id);
}

struct TDq { char * data; long ne[4]; long nb[4]; };
TDq to_tdq(const pm355_tensor * t) {
    TDq d; d.data = (char *) t->data;
    for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = (long) t->nb[i]; }
    return d;
}

// ---- 1. CPY f32 -> Q8_0 ------------------------------------------------------------------------------------------------------
// src: f32, any strides with nb[0] == 4 and ne[0] % 32 == 0; dst: contiguous Q8_0 blocks in the source's logical row order
// (ggml_compute_forward_dup_f32 with a quantized destination, ggml.c:8560-8590)
__global__ __launch_bounds__(256) void cpy_f32_q8_0_kernel(TDq s, uint8_t * dst, long n_blocks) {
    const long t = (long) blockIdx.x * 256 + threadIdx.x;
    const long blk = t >> 5; const int l = (int) (t & 31);
    const bool live = blk < n_blocks;
    const long bpr = s.ne[0] / 32;               // blocks per source row
    const long row = live ? blk / bpr : 0, b0 = live ? blk % bpr : 0;
    long r = row;
    const long i1 = r % s.ne[1]; r /= s.ne[1]; const long i2 = r % s.ne[2]; const long i3 = r / s.ne[2];
    const float x = live ? *(const float *) (s.data + (b0 * 32 + l) * 4 + i1 * s.nb[1] + i2 * s.nb[2] + i3 * s.nb[3]) : 0.0f;
    uint16_t d16;
    const int q = q8_quant32(x, d16);
    if (!live) return;
    uint8_t * o = dst + blk * QB;
    if (l == 0) *(uint16_t *) o = d16;
    o[2 + l] = (uint8_t) (int8_t) q;
}

// ---- 2. flash-attention node with a quantized K and / or V ------------------------------------------------------------------------
// Same structure as flash_attn_ext_kernel (ggml_ops.hip): one 256-thread workgroup per (query, head, batch), scores in LDS.
template <bool KQ8, bool VQ8>
__global__ __launch_bounds__(256) void flash_attn_ext_q8_kernel(TDq q, TDq k, TDq v, const char * mask, long mask_nb1, TDq d, float scale,
                                                              float softcap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float redf[8];
    __shared__ float dq[8];                       // Q8_0 scales of the query row (D <= 256)
    const int D = (int) q.ne[0];
    const int n_kv = (int) k.ne[1];
    float * qs = (float *) smem;                  // [D] query: F16-rounded values, or its int8 values (as floats) when K is Q8_0
    float * sc = qs + D;                          // [n_kv]
    float * part = sc + ((n_kv + 3) & ~3);        // [nslot][D]
    const int iq1 = blockIdx.x, h = blockIdx.y, b3 = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = h / (int) (q.ne[2] / k.ne[2]), hv = h / (int) (q.ne[2] / v.ne[2]);
    const int bk = b3 / (int) (q.ne[3] / k.ne[3]), bv = b3 / (int) (q.ne[3] / v.ne[3]);
    const char * qp = q.data + (long) iq1 * q.nb[1] + (long) h * q.nb[2] + (long) b3 * q.nb[3];
    {   // D is a multiple of 64 here (launcher): whole waves
        const float x = tid < D ? *(const float *) (qp + (long) tid * 4) : 0.0f;
        if (KQ8) {
            uint16_t d16;
            const int qi = q8_quant32(x, d16);
            if (tid < D) { qs[tid] = (float) qi; if ((tid & 31) == 0) dq[tid >> 5] = h2f(d16); }
        } else if (tid < D) qs[tid] = h2f(f2h(x));
    }
    const uint16_t * mp = mask ? (const uint16_t *) (mask + (long) iq1 * mask_nb1) : nullptr;
    __syncthreads();
    float lmax = -INFINITY;
    for (int i = tid; i < n_kv; i += 256) {
        const float mv = mp ? h2f(mp[i]) : 0.0f;
        float s = -INFINITY;
        if (mv != -INFINITY) {
            const uint8_t * kr = (const uint8_t *) (k.data + (long) i * k.nb[1] + (long) hk * k.nb[2] + (long) bk * k.nb[3]);
            float acc = 0.0f;
            if (KQ8) {
                for (int b = 0; b < D / 32; ++b) {
                    const uint8_t * blk = kr + b * QB;
                    float sumi = 0.0f;                                   // exact in f32: |sum| <= 32 * 127 * 127
#pragma unroll
                    for (int j = 0; j < 16; ++j) { float x0, x1; q8_pair(blk, j, x0, x1); sumi += x0 * qs[32 * b + 2 * j] + x1 * qs[32 * b + 2 * j + 1]; }
                    acc += sumi * (q8_d(blk) * dq[b]);
                }
            } else {
                const uint16_t * kh = (const uint16_t *) kr;
                for (int e = 0; e < D; ++e) acc += h2f(kh[e]) * qs[e];
            }
            s = acc * scale;
            if (softcap != 0.0f) s = softcap * tanhf(s);
            s += mv;
        }
        sc[i] = s;
        lmax = fmaxf(lmax, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if (lane == 0) redf[wave] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    float lsum = 0.0f;
    for (int i = tid; i < n_kv; i += 256) {
        const float p = (sc[i] == -INFINITY || mx == -INFINITY) ? 0.0f : expf(sc[i] - mx);
        sc[i] = p;
        lsum += p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
    __syncthreads();
    if (lane == 0) redf[4 + wave] = lsum;
    __syncthreads();
    const float S = (redf[4] + redf[5]) + (redf[6] + redf[7]);
    const int nchunk = D / 8, nslot = 256 / nchunk;
    const int c = tid % nchunk, slot = tid / nchunk;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = slot; i < n_kv; i += nslot) {
        const float p = sc[i];
        if (p == 0.0f) continue;
        const uint8_t * vr = (const uint8_t *) (v.data + (long) i * v.nb[1] + (long) hv * v.nb[2] + (long) bv * v.nb[3]);
        if (VQ8) {
            const uint8_t * blk = vr + (c / 4) * QB;
            const float dd = q8_d(blk);
#pragma unroll
            for (int j = 0; j < 4; ++j) { float x0, x1; q8_pair(blk, (c % 4) * 4 + j, x0, x1); acc[2 * j] += (x0 * dd) * p; acc[2 * j + 1] += (x1 * dd) * p; }
        } else {
            const uint16_t * vh = (const uint16_t *) vr + 8 * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += h2f(vh[j]) * p;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[slot * D + 8 * c + j] = acc[j];
    __syncthreads();
    char * dp = d.data + (long) h * d.nb[1] + (long) iq1 * d.nb[2] + (long) b3 * d.nb[3];
    for (int e = tid; e < D; e += 256) {
        float o = 0.0f;
        for (int sl = 0; sl < nslot; ++sl) o += part[sl * D + e];
        *(float *) (dp + (long) e * 4) = S > 0.0f ? o / S : 0.0f;
    }
}

// ---- 3. one token: RoPE + quantized KV store + attention ---------------------------------------------------------------------------
struct Q8TokP {
    const float * q, * k, * v; uint8_t * kc, * vc; const int32_t * pos, * dyn; const void * mask; int mask_f16;
    const float * ff; float * out; int H, Hkv, n_ctx; float scale; RopeP r;
};

template <int DH, bool KQ8, bool VQ8>
__global__ __launch_bounds__(256) void attn_q8_token_kernel(Q8TokP a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float redf[8];
    __shared__ float dq[8], dk[8];
    constexpr int NB = DH / 32, C8 = DH / 8, NSL = 256 / C8;
    float * qf = (float *) smem;                  // [DH] rotated q: F16-rounded, or its int8 values
    float * kf = qf + DH;                         // [DH] rotated k of this token, likewise
    float * vf = kf + DH;                         // [DH] v of this token as the cache holds it (F16-rounded / dequantized)
    float * part = vf + DH;                       // [NSL][DH]
    float * sc = part + NSL * DH;                 // [n_kv]
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, Hkv = a.Hkv;
    const int hk = h / (H / Hkv);
    const bool writer = h % (H / Hkv) == 0;       // one query head per KV head stores the token's K / V
    const int pos = a.pos[0], slot = a.dyn[0], n_kv = a.dyn[1];
    const RopeP r = a.r;
    const bool neox = r.mode & 2;
    const int half = r.n_dims / 2;
    const long k_row = KQ8 ? (long) Hkv * NB * QB : (long) Hkv * DH * 2;       // bytes per cache row
    const long v_row = VQ8 ? (long) Hkv * NB * QB : (long) Hkv * DH * 2;
    const long k_head = KQ8 ? (long) hk * NB * QB : (long) hk * DH * 2;
    const long v_head = VQ8 ? (long) hk * NB * QB : (long) hk * DH * 2;
    // rotate q (threads 0 .. DH/2-1) and k (DH/2 .. DH-1): same arithmetic as attn_rope_body, unrounded f32 into LDS
    if (tid < DH) {
        const bool is_k = tid >= DH / 2;
        const int pair = is_k ? tid - DH / 2 : tid;
        const float * src = is_k ? a.k + (long) hk * DH : a.q + (long) h * DH;
        int ia, ib;
        if (pair < half) { ia = neox ? pair : 2 * pair; ib = neox ? pair + half : 2 * pair + 1; }
        else             { ia = r.n_dims + 2 * (pair - half); ib = ia + 1; }
        float o0 = src[ia], o1 = src[ib];
        if (pair < half) {
            float c, s_;
            rope_cs(r, (float) pos, pair, a.ff, c, s_);
            const float x0 = o0, x1 = o1;
            o0 = x0 * c - x1 * s_; o1 = x0 * s_ + x1 * c;
        }
        float * dst = is_k ? kf : qf;
        dst[ia] = o0; dst[ib] = o1;
        vf[tid] = a.v[(long) hk * DH + tid];
    }
    __syncthreads();
    // quantize / round the three rows in place and store the token's K and V in the cache
    if (tid < DH) {
        const float xq = qf[tid], xk = kf[tid], xv = vf[tid];
        uint8_t * kdst = a.kc + (long) slot * k_row + k_head, * vdst = a.vc + (long) slot * v_row + v_head;
        if (KQ8) {
            uint16_t d16;
            const int qi = q8_quant32(xq, d16);
            qf[tid] = (float) qi; if ((tid & 31) == 0) dq[tid >> 5] = h2f(d16);
            const int ki = q8_quant32(xk, d16);
            kf[tid] = (float) ki; if ((tid & 31) == 0) dk[tid >> 5] = h2f(d16);
            if (writer) { uint8_t * o = kdst + (tid >> 5) * QB; if ((tid & 31) == 0) *(uint16_t *) o = d16; o[2 + (tid & 31)] = (uint8_t) (int8_t) ki; }
        } else {
            qf[tid] = h2f(f2h(xq));
            const uint16_t hk16 = f2h(xk);
            kf[tid] = h2f(hk16);
            if (writer) ((uint16_t *) kdst)[tid] = hk16;
        }
        if (VQ8) {
            uint16_t d16;
            const int vi = q8_quant32(xv, d16);
            vf[tid] = (float) vi * h2f(d16);
            if (writer) { uint8_t * o = vdst + (tid >> 5) * QB; if ((tid & 31) == 0) *(uint16_t *) o = d16; o[2 + (tid & 31)] = (uint8_t) (int8_t) vi; }
        } else {
            const uint16_t hv16 = f2h(xv);
            vf[tid] = h2f(hv16);
            if (writer) ((uint16_t *) vdst)[tid] = hv16;
        }
    }
    __syncthreads();
    // scores
    float lmax = -INFINITY;
    for (int i = tid; i < n_kv; i += 256) {
        float acc = 0.0f;
        if (i == slot) {                          // the token's own key: from LDS (its cache row is being written by this launch)
            if (KQ8) {
                for (int b = 0; b < NB; ++b) { float sumi = 0.0f; for (int j = 0; j < 32; ++j) sumi += kf[32 * b + j] * qf[32 * b + j]; acc += sumi * (dk[b] * dq[b]); }
            } else for (int e = 0; e < DH; ++e) acc += kf[e] * qf[e];
        } else {
            const uint8_t * kr = a.kc + (long) i * k_row + k_head;
            if (KQ8) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const uint8_t * blk = kr + b * QB;
                    float sumi = 0.0f;
#pragma unroll
                    for (int j = 0; j < 16; ++j) { float x0, x1; q8_pair(blk, j, x0, x1); sumi += x0 * qf[32 * b + 2 * j] + x1 * qf[32 * b + 2 * j + 1]; }
                    acc += sumi * (q8_d(blk) * dq[b]);
                }
            } else {
                const PM_G u32x4 * k4 = (const PM_G u32x4 *) kr;
#pragma unroll
                for (int jj = 0; jj < DH / 8; ++jj) {
                    const u32x4 kk = k4[jj];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc += h2f((uint16_t) (kk[j] & 0xFFFF)) * qf[8 * jj + 2 * j];
                        acc += h2f((uint16_t) (kk[j] >> 16)) * qf[8 * jj + 2 * j + 1];
                    }
                }
            }
        }
        const float s_ = acc * a.scale + attn_mask_at(a.mask, a.mask_f16, i);
        sc[i] = s_;
        lmax = fmaxf(lmax, s_);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if (lane == 0) redf[wave] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    float lsum = 0.0f;
    for (int i = tid; i < n_kv; i += 256) {
        const float e = (sc[i] == -INFINITY || mx == -INFINITY) ? 0.0f : expf(sc[i] - mx);
        sc[i] = e;
        lsum += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
    __syncthreads();
    if (lane == 0) redf[4 + wave] = lsum;
    __syncthreads();
    const float S = (redf[4] + redf[5]) + (redf[6] + redf[7]);
    const float inv = S > 0.0f ? 1.0f / S : 0.0f;
    // P.V over the cached rows (thread = 8-channel chunk c8, key slot ks), the token's own row from LDS
    const int c8 = tid % C8, ks = tid / C8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = ks; i < n_kv; i += NSL) {
        if (i == slot) continue;
        const float p = sc[i];
        if (p == 0.0f) continue;
        const uint8_t * vr = a.vc + (long) i * v_row + v_head;
        if (VQ8) {
            const uint8_t * blk = vr + (c8 / 4) * QB;
            const float dd = q8_d(blk);
#pragma unroll
            for (int j = 0; j < 4; ++j) { float x0, x1; q8_pair(blk, (c8 % 4) * 4 + j, x0, x1); acc[2 * j] += (x0 * dd) * p; acc[2 * j + 1] += (x1 * dd) * p; }
        } else {
            const u32x4 vv = *(const PM_G u32x4 *) (vr + 16 * c8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[2 * j] += h2f((uint16_t) (vv[j] & 0xFFFF)) * p; acc[2 * j + 1] += h2f((uint16_t) (vv[j] >> 16)) * p; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[ks * DH + 8 * c8 + j] = acc[j];
    __syncthreads();
    if (tid < DH) {
        float o = 0.0f;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) o += part[sl * DH + tid];
        o += vf[tid] * sc[slot];
        a.out[(long) h * DH + tid] = o * inv;
    }
}

// ---- 3b. long contexts: the same rotation + store alone (one launch), so that the attention can run on the matrix cores over cells that are ALL
// in the cache (attn_flash_mfma.hip with Q8_0 K / V). q_out[h * DH + e] = what the K.q product multiplies the cache with: the F16-rounded rotated
// query, or for a Q8_0 K cache its Q8_0 quantization (the reference quantizes the query row too: q_to_vec_dot, ggml.c:15905-15912) written as
// the values q_i * d rounded to F16 (exactly representable operands for the F16 matrix instruction).
template <int DH, bool KQ8, bool VQ8>
__global__ __launch_bounds__(DH) void q8_token_prep_kernel(Q8TokP a, float * q_out) {
    __shared__ float qf[DH], kf[DH];
    constexpr int NB = DH / 32;
    const int h = blockIdx.x, tid = threadIdx.x;
    const int H = a.H, Hkv = a.Hkv;
    const int hk = h / (H / Hkv);
    const bool writer = h % (H / Hkv) == 0;
    const int pos = a.pos[0], slot = a.dyn[0];
    const RopeP r = a.r;
    const bool neox = r.mode & 2;
    const int half = r.n_dims / 2;
    const long k_row = KQ8 ? (long) Hkv * NB * QB : (long) Hkv * DH * 2;
    const long v_row = VQ8 ? (long) Hkv * NB * QB : (long) Hkv * DH * 2;
    const long k_head = KQ8 ? (long) hk * NB * QB : (long) hk * DH * 2;
    const long v_head = VQ8 ? (long) hk * NB * QB : (long) hk * DH * 2;
    {   // rotate q (threads 0 .. DH/2-1) and k (DH/2 .. DH-1): attn_q8_token_kernel's arithmetic
        const bool is_k = tid >= DH / 2;
        const int pair = is_k ? tid - DH / 2 : tid;
        const float * src = is_k ? a.k + (long) hk * DH : a.q + (long) h * DH;
        int ia, ib;
        if (pair < half) { ia = neox ? pair : 2 * pair; ib = neox ? pair + half : 2 * pair + 1; }
        else             { ia = r.n_dims + 2 * (pair - half); ib = ia + 1; }
        float o0 = src[ia], o1 = src[ib];
        if (pair < half) {
            float c, s_;
            rope_cs(r, (float) pos, pair, a.ff, c, s_);
            const float x0 = o0, x1 = o1;
            o0 = x0 * c - x1 * s_; o1 = x0 * s_ + x1 * c;
        }
        float * dst = is_k ? kf : qf;
        dst[ia] = o0; dst[ib] = o1;
    }
    __syncthreads();
    const float xq = qf[tid], xk = kf[tid], xv = a.v[(long) hk * DH + tid];
    uint8_t * kdst = a.kc + (long) slot * k_row + k_head, * vdst = a.vc + (long) slot * v_row + v_head;
    if (KQ8) {
        uint16_t d16;
        const int qi = q8_quant32(xq, d16);
        q_out[(long) h * DH + tid] = h2f(f2h((float) qi * h2f(d16)));
        const int ki = q8_quant32(xk, d16);
        if (writer) { uint8_t * o = kdst + (tid >> 5) * QB; if ((tid & 31) == 0) *(uint16_t *) o = d16; o[2 + (tid & 31)] = (uint8_t) (int8_t) ki; }
    } else {
        q_out[(long) h * DH + tid] = h2f(f2h(xq));
        if (writer) ((uint16_t *) kdst)[tid] = f2h(xk);
    }
    if (VQ8) {
        uint16_t d16;
        const int vi = q8_quant32(xv, d16);
        if (writer) { uint8_t * o = vdst + (tid >> 5) * QB; if ((tid & 31) == 0) *(uint16_t *) o = d16; o[2 + (tid & 31)] = (uint8_t) (int8_t) vi; }
    } else if (writer) ((uint16_t *) vdst)[tid] = f2h(xv);
}

// > 48 KiB of dynamic LDS needs the function attribute once per (kernel, device); the kernels here share their pointer TYPE, so the
// bookkeeping is keyed on the pointer value
void set_lds(const void * kern, size_t lds) {
    if (lds <= 48 * 1024) return;
    static const void * done[64][2] = {};
    static int n_done = 0;
    const void * dev = (const void *) (uintptr_t) (pm_cur_dev() + 1);
    for (int i = 0; i < n_done; ++i) if (done[i][0] == kern && done[i][1] == dev) return;
    (void) hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (n_done < 64) { done[n_done][0] = kern; done[n_done][1] = dev; ++n_done; }
}

} // namespace

// CPY f32 -> Q8_0 (dst = n/32 contiguous native blocks); -1: unsupported
int pm_launch_cpy_f32_q8_0(const pm355_tensor * src, void * dst, hipStream_t st) {
    if (src->type != PM_F32 || src->nb[0] != 4 || src->ne[0] % 32) return -1;
    const long n_blocks = (long) (src->ne[0] / 32) * src->ne[1] * src->ne[2] * src->ne[3];
    if (n_blocks <= 0) return 0;
    hipLaunchKernelGGL(cpy_f32_q8_0_kernel, dim3((unsigned) ((n_blocks * 32 + 255) / 256)), dim3(256), 0, st, to_tdq(src), (uint8_t *) dst, n_blocks);
    return 0;
}

// FLASH_ATTN_EXT with K and / or V of type Q8_0 (the other one F16); -1: unsupported shape, -2: scores do not fit LDS
int pm_launch_flash_attn_ext_q8(const pm355_tensor * q, const pm355_tensor * k, const pm355_tensor * v, const pm355_tensor * mask,
                                const pm355_tensor * dst, float scale, float softcap, hipStream_t st) {
    const bool kq8 = k->type == PM_Q8_0, vq8 = v->type == PM_Q8_0;
    const int D = (int) q->ne[0];
    if ((D != 64 && D != 128 && D != 256) || (!kq8 && !vq8)) return -1;
    if ((kq8 ? k->nb[0] != QB : k->nb[0] != 2) || (vq8 ? v->nb[0] != QB : v->nb[0] != 2)) return -1;
    const int nslot = 256 / (D / 8);
    const size_t lds = ((size_t) D + (((size_t) k->ne[1] + 3) & ~(size_t) 3) + (size_t) nslot * D) * 4;
    if (lds > 150 * 1024) return -2;
    if (softcap != 0.0f) scale /= softcap;
    const dim3 grid((unsigned) q->ne[1], (unsigned) q->ne[2], (unsigned) q->ne[3]);
    auto go = [&](auto kern) {
        set_lds((const void *) kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, to_tdq(q), to_tdq(k), to_tdq(v), mask ? (const char *) mask->data : nullptr,
                           mask ? (long) mask->nb[1] : 0, to_tdq(dst), scale, softcap);
    };
    if (kq8 && vq8) go(flash_attn_ext_q8_kernel<true, true>);
    else if (kq8)   go(flash_attn_ext_q8_kernel<true, false>);
    else            go(flash_attn_ext_q8_kernel<false, true>);
    return 0;
}

// one token, long context: RoPE + (quantizing) KV store in one small launch (q_rot receives the operand rows of the K.q product), then the caller
// runs pm_launch_attn_flash_cached over the cached cells. Row-major caches. -1: unsupported
int pm_launch_q8_token_prep(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * pos, const int32_t * dyn, const float * ff,
                            float * q_rot, int H, int Hkv, int dh, int n_ctx, const pm_rope_cfg & c, int k_q8, int v_q8, hipStream_t st) {
    if ((dh != 64 && dh != 128) || H % Hkv || (!k_q8 && !v_q8) || !dyn || !q_rot) return -1;
    Q8TokP a = {};
    a.q = q; a.k = k; a.v = v; a.kc = (uint8_t *) kc; a.vc = (uint8_t *) vc; a.pos = pos; a.dyn = dyn; a.ff = ff; a.H = H; a.Hkv = Hkv; a.n_ctx = n_ctx;
    a.r.n_dims = c.n_dims; a.r.mode = c.mode; a.r.n_ctx_orig = c.n_ctx_orig; a.r.theta_scale = c.theta_scale; a.r.freq_scale = c.freq_scale;
    a.r.ext_factor = c.ext_factor; a.r.attn_factor = c.attn_factor; a.r.corr0 = c.corr0; a.r.corr1 = c.corr1;
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(H), dim3(dh), 0, st, a, q_rot); };
    if (dh == 128) {
        if (k_q8 && v_q8) go(q8_token_prep_kernel<128, true, true>); else if (k_q8) go(q8_token_prep_kernel<128, true, false>); else go(q8_token_prep_kernel<128, false, true>);
    } else {
        if (k_q8 && v_q8) go(q8_token_prep_kernel<64, true, true>); else if (k_q8) go(q8_token_prep_kernel<64, true, false>); else go(q8_token_prep_kernel<64, false, true>);
    }
    return 0;
}

// one token, one launch: RoPE + KV store (quantizing) + attention over cells [0, dyn[1]); row-major caches. -1: unsupported
int pm_launch_attn_q8_token(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * pos, const int32_t * dyn,
                            const void * mask, int mask_f16, const float * ff, float * out, int H, int Hkv, int dh, int n_ctx, float scale,
                            const pm_rope_cfg & c, int k_q8, int v_q8, int max_keys, hipStream_t st) {
    if ((dh != 64 && dh != 128) || H % Hkv || (!k_q8 && !v_q8) || !dyn) return -1;
    const int keys = max_keys > 0 ? max_keys : n_ctx;
    const size_t lds = ((size_t) 3 * dh + 2048 + (size_t) ((keys + 7) & ~7) + 8) * 4;
    if (lds > 150 * 1024) return -1;
    Q8TokP a;
    a.q = q; a.k = k; a.v = v; a.kc = (uint8_t *) kc; a.vc = (uint8_t *) vc; a.pos = pos; a.dyn = dyn; a.mask = mask; a.mask_f16 = mask_f16;
    a.ff = ff; a.out = out; a.H = H; a.Hkv = Hkv; a.n_ctx = n_ctx; a.scale = scale;
    a.r.n_dims = c.n_dims; a.r.mode = c.mode; a.r.n_ctx_orig = c.n_ctx_orig; a.r.theta_scale = c.theta_scale; a.r.freq_scale = c.freq_scale;
    a.r.ext_factor = c.ext_factor; a.r.attn_factor = c.attn_factor; a.r.corr0 = c.corr0; a.r.corr1 = c.corr1;
    auto go = [&](auto kern) { set_lds((const void *) kern, lds); hipLaunchKernelGGL(kern, dim3(H), dim3(256), lds, st, a); };
    if (dh == 128) {
        if (k_q8 && v_q8) go(attn_q8_token_kernel<128, true, true>); else if (k_q8) go(attn_q8_token_kernel<128, true, false>); else go(attn_q8_token_kernel<128, false, true>);
    } else {
        if (k_q8 && v_q8) go(attn_q8_token_kernel<64, true, true>); else if (k_q8) go(attn_q8_token_kernel<64, true, false>); else go(attn_q8_token_kernel<64, false, true>);
    }
    return 0;
}
