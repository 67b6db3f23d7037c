// attn_device.h — device code of the fused single-token attention (RoPE + KV store + K.q + softmax + V.p) launched by
// layer_ops.hip. Design notes: layer_ops.hip.
#pragma once
#include "pm355_device.h"

struct RopeP {
    int n_dims, mode, n_ctx_orig;
    float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1;
};

__device__ __forceinline__ void rope_cs(const RopeP & r, float pos, int pair, const float * ff, float & c, float & s) {
    float theta = pos;
    for (int j = 0; j < pair; ++j) theta *= r.theta_scale;
    const float f = ff ? ld_g(ff + pair) : 1.0f;
    const float te = theta / f;
    float ti = r.freq_scale * te, th = ti, ms = r.attn_factor;
    if (r.ext_factor != 0.0f) {
        const float y = ((float) pair - r.corr0) / fmaxf(0.001f, r.corr1 - r.corr0);   // i0/2 == pair
        const float ramp = 1 - fminf(1, fmaxf(0, y));
        const float mix = ramp * r.ext_factor;
        th = ti * (1 - mix) + te * mix;
        ms *= 1.0f + 0.1f * logf(1.0f / r.freq_scale);
    }
    c = cosf(th) * ms;
    s = sinf(th) * ms;
}

struct AttnP {
    const float * q, * k, * v; uint16_t * kc, * vc; const int32_t * pos0_ptr, * seq_ptr; long seq_stride;
    const float * freq_factors; float * out; int H, Hkv, n_ctx; float scale; RopeP r;
    // ggml-graph mode (plug-in, pm355_attn_token): dyn != nullptr -> the RoPE position is pos0_ptr[0] (the graph's inp_pos), the
    // token's K row / V column go to cache cell dyn[0] (llama_kv_cache head, src/llama.cpp:9688) and cells [0, dyn[1]) are attended
    // with the additive f32 KQ mask row `mask` (0 / -inf, llama_set_inputs src/llama.cpp:17379-17420). Engine mode (dyn == nullptr):
    // cell == position, cells [0, pos] attended, no mask.
    const int32_t * dyn; const float * mask;
    // mask_f16 != 0: `mask` points to F16 values (the flash-attention graphs cast the KQ mask, build_inp_KQ_mask src/llama.cpp:10466)
    int mask_f16;
    unsigned long long * ts;                 // measurement builds (-DPM_TS)
    int tok;                                 // engine mode, cached form: token index of a multi-token launch (position = pos0 + tok; q / out rows are advanced by the kernel)
};
__device__ __forceinline__ float attn_mask_at(const void * mask, int f16, int i) {
    if (!mask) return 0.0f;
    return f16 ? h2f(((const PM_G uint16_t *) mask)[i]) : ((const PM_G float *) mask)[i];
}

// Body of one query head `h`. Written for 256 ACTIVE threads; a larger workgroup passes its extra threads through: they only take
// part in the barriers.
// VM = layout of the V cache: 0 = transposed [n_embd_v_gqa][n_ctx] (llm_build_kv_store without flash attention, src/llama.cpp:9712),
// 1 = row-major [n_ctx][n_embd_v_gqa] like K (flash-attention graphs, :9705) - with the reference's flash-attention rounding points
// (ggml_compute_forward_flash_attn_ext_f16, ggml.c:15870-16050): q -> F16, probabilities stay f32 (no F16 rounding of P).
// CACHED: the wq | wk | wv launch already rotated q (F16-rounded values in a.q) and stored this token's K row / V column (QkvEpi,
// mmvq_device.h): no rope, no store, the token's own key is a cached key like any other.
template <int DH, bool COH, int VM = 0, bool CACHED = false>
__device__ __forceinline__ void attn_rope_body(const AttnP & a, int h, char * smem, float * redf /*[8]*/, double * redd /*[4]*/) {
    // all of these are device-global memory: the explicit address space keeps the accesses global_* (not FLAT) when the
    // pointers may come out of a descriptor in memory
    const PM_G float * q = (const PM_G float *) a.q, * k = (const PM_G float *) a.k, * v = (const PM_G float *) a.v;
    PM_G uint16_t * kc = (PM_G uint16_t *) a.kc, * vc = (PM_G uint16_t *) a.vc;
    const PM_G int32_t * pos0_ptr = (const PM_G int32_t *) a.pos0_ptr, * seq_ptr = (const PM_G int32_t *) a.seq_ptr; const long seq_stride = a.seq_stride;
    const float * freq_factors = a.freq_factors; float * out = a.out;
    const int H = a.H, Hkv = a.Hkv, n_ctx = a.n_ctx; const float scale = a.scale; const RopeP r = a.r;
    const bool active = threadIdx.x < 256;
    unsigned long long tsv[6] = {PM_TS_NOW(), 0, 0, 0, 0, 0};
    constexpr int PARTS = 256 / DH;
    constexpr int KQ = DH / 8;                   // 16-byte pieces per K row
    float * qs   = (float *) smem;               // [DH]  f16-rounded rotated q
    float * kcur = qs + DH;                      // [DH]  f16-rounded rotated k of this position
    float * vcur = kcur + DH;                    // [DH]  f16-rounded v of this position
    float * cs   = vcur + DH;                    // [DH]  cos/sin per pair
    constexpr int C8 = DH / 8, NSL = 256 / C8;   // VM 1: 16-byte chunks per V row, key slots per workgroup
    constexpr int PART_FLOATS = VM == 0 ? 256 : 2048;
    float * part = cs + DH;                      // PV partials: [256] (VM 0) / [NSL][DH] (VM 1)
    float * sc   = part + PART_FLOATS;           // [n_ctx] scores / probabilities
    const int tid = active ? (int) threadIdx.x : (1 << 28), lane = tid & 63, wave = tid >> 6;   // passengers: every range test fails
    const int hk = h / (H / Hkv);
    const bool neox = r.mode & 2;
    const int half = r.n_dims / 2;
    // ---- (0) loads that depend on nothing but the head go out first: this token's q / k pair and v element
    const bool is_k = tid >= DH / 2;
    const int pair = is_k ? tid - DH / 2 : tid;
    int ia = 0, ib = 0; float x0 = 0.0f, x1 = 0.0f, vnew = 0.0f;
    if (CACHED) {
        if (tid < DH) x0 = q[(long) h * DH + tid];
    } else {
    if (tid < DH) {
        const PM_G float * src = is_k ? k + (long) hk * DH : q + (long) h * DH;
        if (pair < half) { ia = neox ? pair : 2 * pair; ib = neox ? pair + half : 2 * pair + 1; }
        else             { ia = r.n_dims + 2 * (pair - half); ib = ia + 1; }
        x0 = src[ia]; x1 = src[ib];
    }
    if (DH > 128 || tid >= 128) { const int e = DH > 128 ? tid : tid - 128; if (e < DH) vnew = v[(long) hk * DH + e]; }
    }
    // ---- (1) + (2) position and the first K / V^T loads. The ADDRESSES of this thread's K row of the first sweep (cell = tid) and of its
    //      first two V^T chunks do not depend on the position, only on the sequence's slab - so with a single slab (seq_stride == 0)
    //      they are issued together with the position / q / k / v loads: ONE exposed memory latency instead of two (which cells are
    //      valid is decided later; rows beyond n_kv are loaded and ignored). With several slabs the sequence id has to arrive first.
    u32x4 kreg[KQ];
    const int ve = tid % DH, vpt = tid / DH;
    u32x4 vreg0 = {0, 0, 0, 0}, vreg1 = {0, 0, 0, 0};
    auto first_loads = [&](const PM_G uint16_t * kcb, const PM_G uint16_t * vcb) __attribute__((always_inline)) {
        const PM_G uint16_t * kr = kcb + (long) (tid < n_ctx ? tid : 0) * Hkv * DH + (long) hk * DH;
#pragma unroll
        for (int j = 0; j < KQ; ++j) kreg[j] = *(const PM_G u32x4 *) (kr + 8 * j);
        if (VM == 0) {
            const PM_G uint16_t * vr = vcb + (long) (hk * DH + ve) * n_ctx;
            const int i0 = vpt * 8, i1 = i0 + PARTS * 8;
            vreg0 = *(const PM_G u32x4 *) (vr + (i0 + 8 <= n_ctx ? i0 : 0));
            vreg1 = *(const PM_G u32x4 *) (vr + (i1 + 8 <= n_ctx ? i1 : 0));
        } else {                                 // rows (tid / C8) and (tid / C8 + NSL), 16-byte chunk tid % C8 of this KV head
            const int c8 = tid % C8, ks = tid / C8;
            const PM_G uint16_t * vr = vcb + (long) hk * DH + 8 * c8;
            vreg0 = *(const PM_G u32x4 *) (vr + (long) (ks < n_ctx ? ks : 0) * Hkv * DH);
            vreg1 = *(const PM_G u32x4 *) (vr + (long) (ks + NSL < n_ctx ? ks + NSL : 0) * Hkv * DH);
        }
    };
    int seq = 0, pos, slot, n_kv;                // rope position, cache cell of this token, cells attended = [0, n_kv)
    int pv = 0, sv = 0;
    if (seq_ptr) { pv = pos0_ptr[lane]; sv = *seq_ptr; } else pv = pos0_ptr[0];     // (the engine's position table has 64 entries)
    int dyn0 = 0, dyn1 = 0;
    if (a.dyn) { const PM_G int32_t * dyn = (const PM_G int32_t *) a.dyn; dyn0 = dyn[0]; dyn1 = dyn[1]; }
    if (seq_stride == 0 || !seq_ptr) first_loads(kc, vc);
    if (seq_ptr) {
        seq = __builtin_amdgcn_readfirstlane(sv);
        pos = __builtin_amdgcn_readlane(pv, seq) + a.tok;
        slot = pos; n_kv = pos + 1;
    } else {
        pos = __builtin_amdgcn_readfirstlane(pv) + a.tok;
        slot = pos; n_kv = pos + 1;
        if (a.dyn) { slot = __builtin_amdgcn_readfirstlane(dyn0); n_kv = __builtin_amdgcn_readfirstlane(dyn1); }
    }
    if (CACHED) slot = -1;                        // nothing to exclude: every attended cell is read from the cache
    const void * mask = a.mask; const int mf16 = a.mask_f16;
    kc += (long) seq * seq_stride; vc += (long) seq * seq_stride;
    if (seq_stride != 0 && seq_ptr) first_loads(kc, vc);
    const int n_pad = (n_kv + 7) & ~7;           // cached cells [0, n_kv) without `slot`, padded to the 16-byte load width
    const bool have_k = tid < n_kv && tid != slot;
    const PM_G uint16_t * vrow = vc + (long) (hk * DH + ve) * n_ctx;
    // rotate q (threads 0..DH/2-1) and k (threads DH/2..DH-1): each thread builds its own cos/sin
    if (CACHED) {
        if (tid < DH) qs[tid] = x0;
    } else {
    if (tid < DH) {
        float o0 = x0, o1 = x1;
        if (pair < half) {
            float c, s_;
            rope_cs(r, (float) pos, pair, freq_factors, c, s_);
            o0 = x0 * c - x1 * s_; o1 = x0 * s_ + x1 * c;
        }
        const uint16_t h0 = f2h(o0), h1 = f2h(o1);
        float * dst = is_k ? kcur : qs;
        dst[ia] = h2f(h0); dst[ib] = h2f(h1);
        if (is_k && h % (H / Hkv) == 0) {
            PM_G uint16_t * d = kc + (long) slot * Hkv * DH + (long) hk * DH;
            d[ia] = h0; d[ib] = h1;
        }
    }
    if (DH > 128 || tid >= 128) {
        const int e = DH > 128 ? tid : tid - 128;
        if (e < DH) {
            const uint16_t hv = f2h(vnew);
            vcur[e] = h2f(hv);
            if (h % (H / Hkv) == 0) {
                if (VM == 0) vc[(long) (hk * DH + e) * n_ctx + slot] = hv;
                else         vc[(long) slot * Hkv * DH + hk * DH + e] = hv;
            }
        }
    }
    }
    __syncthreads();
    tsv[1] = PM_TS_NOW();                        // loads back, rope done
    // ---- scores: cached cells (thread per key; first sweep from the pre-loaded registers), current key from LDS
    float lmax = -INFINITY;
    auto dot_row = [&](const u32x4 (&kk)[KQ]) __attribute__((always_inline)) {
        float acc = 0.0f;
#pragma unroll
        for (int jj = 0; jj < KQ; ++jj)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc += h2f((uint16_t) (kk[jj][j] & 0xFFFF)) * qs[8 * jj + 2 * j];
                acc += h2f((uint16_t) (kk[jj][j] >> 16)) * qs[8 * jj + 2 * j + 1];
            }
        return acc * scale;
    };
    if (have_k) { const float s_ = dot_row(kreg) + attn_mask_at(mask, mf16, tid); sc[tid] = s_; lmax = s_; }
    for (int i = tid + 256; i < n_kv; i += 256) {
        if (i == slot) continue;
        const PM_G uint16_t * kr = kc + (long) i * Hkv * DH + (long) hk * DH;
        u32x4 kk[KQ];
#pragma unroll
        for (int j = 0; j < KQ; ++j) kk[j] = *(const PM_G u32x4 *) (kr + 8 * j);
        const float s_ = dot_row(kk) + attn_mask_at(mask, mf16, i);
        sc[i] = s_;
        lmax = fmaxf(lmax, s_);
    }
    if (!CACHED && wave == 3) {                  // the current key (from LDS): one wave, lanes over the head dimension
        float acc = 0.0f;
#pragma unroll
        for (int e = lane; e < DH; e += 64) acc += kcur[e] * qs[e];
        const float s_ = wave_sum(acc) * scale + attn_mask_at(mask, mf16, slot);
        if (lane == 0) sc[slot] = s_;
        lmax = fmaxf(lmax, s_);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if (lane == 0 && active) redf[wave] = lmax;
    __syncthreads();
    tsv[2] = PM_TS_NOW();                        // scores
    const float mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double lsum = 0.0;
    for (int i = tid; i < n_kv; i += 256) {
        const float e = expf(sc[i] - mx);
        sc[i] = e;
        lsum += (double) e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
    if (lane == 0 && active) redd[wave] = lsum;
    __syncthreads();
    const double tot = (redd[0] + redd[1]) + (redd[2] + redd[3]);
    const float inv = (float) (1.0 / tot);
    const float p_cur = CACHED ? 0.0f : (VM == 0 ? h2f(f2h(sc[slot] * inv)) : sc[slot] * inv);   // every thread reads exp() of the current key
    __syncthreads();
    for (int i = tid; i < n_pad; i += 256)                             // VM 0: p rounded to F16; pad and `slot` = 0
        sc[i] = (i < n_kv && i != slot) ? (VM == 0 ? h2f(f2h(sc[i] * inv)) : sc[i] * inv) : 0.0f;
    __syncthreads();
    tsv[3] = PM_TS_NOW();                        // softmax
    if (VM == 1) {
        // ---- PV, row-major V: thread (16-byte chunk c8, key slot ks) takes rows ks, ks + NSL, ... (first two pre-loaded)
        const int c8 = tid % C8, ks = tid / C8;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto fma_row = [&](const u32x4 & vv, float pr) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j]     += h2f((uint16_t) (vv[j] & 0xFFFF)) * pr;
                acc[2 * j + 1] += h2f((uint16_t) (vv[j] >> 16)) * pr;
            }
        };
        const PM_G uint16_t * vr = vc + (long) hk * DH + 8 * c8;
        int i = ks;
        if (i < n_kv && i != slot) fma_row(vreg0, sc[i]);
        i += NSL;
        if (i < n_kv && i != slot) fma_row(vreg1, sc[i]);
        for (i += NSL; i < n_kv; i += NSL) {
            if (i == slot) continue;                                    // (that row is being written by this launch)
            const u32x4 vv = *(const PM_G u32x4 *) (vr + (long) i * Hkv * DH);
            fma_row(vv, sc[i]);
        }
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) part[ks * DH + 8 * c8 + j] = acc[j];
        }
        __syncthreads();
        if (tid < DH) {
            float o = 0.0f;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) o += part[sl * DH + tid];
            if (!CACHED) o += vcur[tid] * p_cur;
            st_act<COH>(out + (long) h * DH + tid, o);
        }
        return;
    }
    // ---- PV: thread (e, part) streams V^T[hk*DH+e][8*chunk ..] for chunk = part, part+PARTS, ... (first two pre-loaded)
    {
        float acc = 0.0f;
        auto fma8 = [&](const u32x4 & vv, int i) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc += h2f((uint16_t) (vv[j] & 0xFFFF)) * sc[i + 2 * j];
                acc += h2f((uint16_t) (vv[j] >> 16)) * sc[i + 2 * j + 1];
            }
        };
        int i = vpt * 8;
        if (i < n_pad) fma8(vreg0, i);
        i += PARTS * 8;
        if (i < n_pad) fma8(vreg1, i);
        for (i += PARTS * 8; i < n_pad; i += PARTS * 8) { const u32x4 vv = *(const PM_G u32x4 *) (vrow + i); fma8(vv, i); }
        if (active) part[tid] = acc;
    }
    __syncthreads();
    tsv[4] = PM_TS_NOW();                        // P.V partials
    if (tid < DH) {
        float acc = 0.0f;
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) acc += part[pt * DH + tid];
        if (!CACHED) acc += vcur[tid] * p_cur;
        st_act<COH>(out + (long) h * DH + tid, acc);
    }
    tsv[5] = PM_TS_NOW();
    pm_ts_store(a.ts, 2, tsv);
}
