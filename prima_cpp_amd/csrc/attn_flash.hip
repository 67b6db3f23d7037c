// attn_flash.hip — single-token decode attention for long contexts in ONE launch (flash-decoding), replacing the three launches of
// attn_split.hip (scores -> probabilities + partial P.V -> combine) wherever a context is long enough to leave the one-workgroup-per-head
// kernel (attn_device.h).
//
// Why (tools/long_ctx_probe.py, Llama-3-70B): the split path costs ~25 us per layer "whatever the context" in launches and dependent
// latency chains - 41 us per layer at 2k cells, 60 us at 8k, 132 us at 32k, against 1.2 / 5 / 19 us of KV bytes at HBM speed; decode
// dropped from 116 tok/s (short prompt) to 84 / 75 / 52 tok/s.
//
// One workgroup = (KV head g, span of SPAN keys) serving the R <= 8 query heads of the group (K / V bytes are read once per group):
//   1. RoPE of the group's queries (every workgroup); the workgroup whose span holds the token's cell also rotates k, keeps it in LDS and
//      stores the K row / V column (row) of the token - as attn_split.hip does
//   2. the span is walked in chunks of 128 keys with an ONLINE softmax: s = scale K.q (+ mask), running max m and sum l per head,
//      O <- O exp(m_old - m_new) + P~ V with the unnormalised P~ = exp(s - m) kept in f32
//   3. the partial (m, l, O) of the span is published write-through; a ticket per KV head is taken; the LAST workgroup of the KV head
//      (no waiting, no residency requirement) merges the partials: out = sum_c exp(m_c - m) O_c / sum_c exp(m_c - m) l_c, and resets the
//      ticket for the next launch.
// Rounding points: q, K, V are F16 values as in every other attention kernel here; the probabilities are NOT rounded to F16 after
// normalisation (the reference's non-flash graph rounds them when it converts P for the V product; its flash-attention op keeps them
// f32 like this kernel) - long contexts therefore carry the flash-attention rounding points whichever graph asked. Parity: against the
// split kernels and the f64 restatement within 2e-6 relative (tests/test_gpu_ops.py), through the engine / plug-in at the long-context
// tolerances of tests/test_gpu_engine.py and tests/test_gpu_llama_decode.py.
#include "attn_device.h"
#include "pm355_layer_ops.h"

namespace {

constexpr int CK = 128;                          // keys per chunk (one pass of the workgroup)
constexpr int RMAX = 8;

struct FlashP {
    const float * q, * k, * v; uint16_t * kc, * vc; const int32_t * pos0_ptr, * seq_ptr; long seq_stride;
    const float * ff; float * out;
    float * M, * L, * P; unsigned * ticket;       // scratch: M / L [H][nspan], P [nspan][H][dh], ticket [Hkv] (zero between launches)
    int H, Hkv, n_ctx, nspan, span; float scale; RopeP r;
    const int32_t * dyn; const void * mask; int mask_f16, vm;
};

template <int LPK> __device__ __forceinline__ float group_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v);
    if (LPK == 16) v += dpp_f<0x140>(v);
    return v;
}

template <int DH, int VM>
__global__ __launch_bounds__(256) void attn_flash_kernel(FlashP p) {
    constexpr int LPK = DH / 8;                  // lanes per key (16 bytes of the K row each)
    constexpr int KPP = 256 / LPK;               // keys per pass
    constexpr int NPS = CK / KPP;
    constexpr int PARTS = 256 / DH;              // VM 0: key sub-ranges per chunk (thread = (channel, part))
    constexpr int KP = CK / PARTS;
    constexpr int C8 = DH / 8, NSL = 256 / C8, NR = CK / NSL;   // VM 1: thread = (16-byte chunk c8, key slot ks)
    __shared__ float qs[RMAX][DH];
    __shared__ float sc[RMAX][CK];               // scores, then unnormalised probabilities of the current chunk
    __shared__ float kcur[DH];
    __shared__ float cs2[DH / 2][2];
    __shared__ float mrun[RMAX], lrun[RMAX], resc[RMAX];
    __shared__ float red[(VM == 0 ? PARTS : 4) * RMAX * DH];
    __shared__ int last_flag;
    const int c = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = p.H / p.Hkv;
    const int seq = p.seq_ptr ? *p.seq_ptr : 0;
    int pos = p.pos0_ptr[seq], slot = pos, n_kv = pos + 1;
    if (p.dyn) { slot = p.dyn[0]; n_kv = p.dyn[1]; }
    const int k_begin = c * p.span;
    if (k_begin >= n_kv) return;                 // (not counted: the ticket target is the number of ACTIVE spans)
    const int nact = (n_kv + p.span - 1) / p.span;
    const int k_end = min(k_begin + p.span, n_kv);
    const long krow = (long) p.Hkv * DH;
    uint16_t * kcw = p.kc + (long) seq * p.seq_stride, * vcw = p.vc + (long) seq * p.seq_stride;
    const uint16_t * kc = kcw + (long) g * DH;
    const int piece = tid % LPK, kslot = tid / LPK;
    // memory pipeline: the V pieces of a chunk are requested at its top (they are not needed before the probabilities exist), the K pieces
    // of the NEXT chunk as soon as this chunk's scores have consumed the registers: one exposed round trip per span instead of two per chunk
    constexpr int NVR = VM == 0 ? KP / 8 : NR;
    u32x4 kall[NPS], vall[NVR];
    auto load_k = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) kall[ps] = *(const u32x4 *) (kc + (long) min(k0 + ps * KPP + kslot, p.n_ctx - 1) * krow + 8 * piece);
    };
    auto load_v = [&](int k0) __attribute__((always_inline)) {
        if (VM == 0) {
            const uint16_t * vr = vcw + (long) (g * DH + tid % DH) * p.n_ctx;
#pragma unroll
            for (int j = 0; j < NVR; ++j) vall[j] = *(const u32x4 *) (vr + min(k0 + (tid / DH) * KP + 8 * j, p.n_ctx - 8));
        } else {
            const uint16_t * vr = vcw + (long) g * DH + 8 * (tid % C8);
#pragma unroll
            for (int j = 0; j < NVR; ++j) vall[j] = *(const u32x4 *) (vr + (long) min(k0 + tid / C8 + j * NSL, p.n_ctx - 1) * krow);
        }
    };
    // the first chunk's K and V pieces do not depend on the queries: they are requested before the RoPE (one exposed latency less)
    load_k(k_begin);
    load_v(k_begin);
    // ---- RoPE of the group's queries (+ the token's key / value store in the span that owns its cell): as attn_split_scores_kernel
    const bool neox = p.r.mode & 2;
    const int half = p.r.n_dims / 2;
    if (tid < DH / 2) {
        float cs_ = 1.0f, sn_ = 0.0f;
        if (tid < half) rope_cs(p.r, (float) pos, tid, p.ff, cs_, sn_);
        cs2[tid][0] = cs_; cs2[tid][1] = sn_;
    }
    if (tid < RMAX) { mrun[tid] = -INFINITY; lrun[tid] = 0.0f; }
    __syncthreads();
    const bool mine = slot >= k_begin && slot < k_begin + p.span;
    for (int i = tid; i < (R + (mine ? 1 : 0)) * (DH / 2); i += 256) {
        const int h = i / (DH / 2), pair = i - h * (DH / 2);
        const bool is_k = h == R;
        const float * src = is_k ? p.k + (long) g * DH : p.q + (long) (g * R + h) * DH;
        int ia, ib;
        if (pair < half) { ia = neox ? pair : 2 * pair; ib = neox ? pair + half : 2 * pair + 1; }
        else             { ia = p.r.n_dims + 2 * (pair - half); ib = ia + 1; }
        float o0 = src[ia], o1 = src[ib];
        if (pair < half) {
            const float cs_ = cs2[pair][0], sn_ = cs2[pair][1];
            const float x0 = o0, x1 = o1;
            o0 = x0 * cs_ - x1 * sn_; o1 = x0 * sn_ + x1 * cs_;
        }
        const uint16_t h0 = f2h(o0), h1 = f2h(o1);
        float * dst = is_k ? kcur : qs[h];
        dst[ia] = h2f(h0); dst[ib] = h2f(h1);
        if (is_k) {
            uint16_t * d = kcw + (long) slot * krow + (long) g * DH;
            d[ia] = h0; d[ib] = h1;
        }
    }
    float vcur = 0.0f;                           // the token's own value (channel tid % DH), F16-rounded: used from registers
    if (mine) {
        const int e = tid % DH;
        const uint16_t hv = f2h(p.v[(long) g * DH + e]);
        vcur = h2f(hv);
        if (tid < DH) {
            if (VM == 0) vcw[(long) (g * DH + e) * p.n_ctx + slot] = hv;
            else         vcw[(long) slot * krow + (long) g * DH + e] = hv;
        }
    }
    __syncthreads();
    float qr[RMAX][8];
#pragma unroll
    for (int h = 0; h < RMAX; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) qr[h][i] = h < R ? qs[h][8 * piece + i] : 0.0f;
    // running output: VM 0: thread (channel e, part pt) ; VM 1: thread (8 channels of chunk c8, key slot ks)
    float oacc[RMAX][VM == 0 ? 1 : 8];
#pragma unroll
    for (int h = 0; h < RMAX; ++h)
#pragma unroll
        for (int i = 0; i < (VM == 0 ? 1 : 8); ++i) oacc[h][i] = 0.0f;

    for (int k0 = k_begin; k0 < k_end; k0 += CK) {
        if (k0 != k_begin) load_v(k0);
        // ---- scores of this chunk
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int kin = ps * KPP + kslot, key = k0 + kin;
            const u32x4 kk = kall[ps];
            float kf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { kf[2 * j] = h2f((uint16_t) (kk[j] & 0xFFFF)); kf[2 * j + 1] = h2f((uint16_t) (kk[j] >> 16)); }
            if (key == slot) {
#pragma unroll
                for (int i = 0; i < 8; ++i) kf[i] = kcur[8 * piece + i];
            }
            float acc[RMAX];
#pragma unroll
            for (int h = 0; h < RMAX; ++h) {
                acc[h] = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[h] = fmaf(kf[i], qr[h][i], acc[h]);
            }
#pragma unroll
            for (int h = 0; h < RMAX; ++h) acc[h] = group_sum<LPK>(acc[h]);
            const float mk = (p.mask && key < n_kv) ? attn_mask_at(p.mask, p.mask_f16, key) : 0.0f;
#pragma unroll
            for (int h = 0; h < RMAX; ++h) if (h < R && piece == (h % LPK)) sc[h][kin] = key < n_kv ? acc[h] * p.scale + mk : -INFINITY;
        }
        if (k0 + CK < k_end) load_k(k0 + CK);
        __syncthreads();
        // ---- online softmax statistics: wave w handles heads w, w + 4; sc <- exp(s - m_new)
        for (int h = wave; h < R; h += 4) {
            float m = -INFINITY;
            for (int i = lane; i < CK; i += 64) m = fmaxf(m, sc[h][i]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            const float mo = mrun[h], mn = fmaxf(mo, m);
            float l = 0.0f;
            for (int i = lane; i < CK; i += 64) {
                const float s = sc[h][i];
                const float e = (s == -INFINITY || mn == -INFINITY) ? 0.0f : __expf(s - mn);
                sc[h][i] = e; l += e;
            }
            l = wave_sum(l);
            if (lane == 0) {
                const float f = (mo == -INFINITY || mn == -INFINITY) ? 0.0f : __expf(mo - mn);
                resc[h] = f; lrun[h] = lrun[h] * f + l; mrun[h] = mn;
            }
        }
        __syncthreads();
        // ---- O <- O * resc + P~ . V over this chunk
        if (VM == 0) {
            const int pt = tid / DH;
            constexpr int NV = KP / 8;
#pragma unroll
            for (int h = 0; h < RMAX; ++h) if (h < R) oacc[h][0] *= resc[h];
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int kin = pt * KP + 8 * j;
                if (k0 + kin >= n_kv) continue;
                const u32x4 vv = vall[j];
                float vf[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { vf[2 * i] = h2f((uint16_t) (vv[i] & 0xFFFF)); vf[2 * i + 1] = h2f((uint16_t) (vv[i] >> 16)); }
#pragma unroll
                for (int i = 0; i < 8; ++i) if (k0 + kin + i == slot) vf[i] = vcur;             // the token's own column is being written by this launch
#pragma unroll
                for (int h = 0; h < RMAX; ++h) if (h < R) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) oacc[h][0] = fmaf(vf[i], sc[h][kin + i], oacc[h][0]);
                }
            }
        } else {
            const int c8 = tid % C8, ks = tid / C8;
#pragma unroll
            for (int h = 0; h < RMAX; ++h) if (h < R) {
                const float f = resc[h];
#pragma unroll
                for (int i = 0; i < 8; ++i) oacc[h][i] *= f;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int kin = ks + j * NSL, key = k0 + kin;
                if (key >= n_kv) continue;
                const u32x4 vv = vall[j];
                float vf[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { vf[2 * i] = h2f((uint16_t) (vv[i] & 0xFFFF)); vf[2 * i + 1] = h2f((uint16_t) (vv[i] >> 16)); }
                if (key == slot) {               // the token's own row is being written by this launch: values from the projection
#pragma unroll
                    for (int i = 0; i < 8; ++i) vf[i] = h2f(f2h(p.v[(long) g * DH + 8 * c8 + i]));
                }
#pragma unroll
                for (int h = 0; h < RMAX; ++h) if (h < R) {
                    const float pr = sc[h][kin];
#pragma unroll
                    for (int i = 0; i < 8; ++i) oacc[h][i] = fmaf(vf[i], pr, oacc[h][i]);
                }
            }
        }
        __syncthreads();                         // sc / resc are rewritten by the next chunk
    }
    // ---- fold the per-thread partial outputs of the workgroup -> red[.][h][e], then publish (m, l, O) of this span
    if (VM == 0) {
        const int e = tid % DH, pt = tid / DH;
#pragma unroll
        for (int h = 0; h < RMAX; ++h) if (h < R) red[(pt * RMAX + h) * DH + e] = oacc[h][0];
    } else {
        const int c8 = tid % C8;
#pragma unroll
        for (int h = 0; h < RMAX; ++h) if (h < R) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v_ = oacc[h][i];
#pragma unroll
                for (int off = C8; off < 64; off <<= 1) v_ += __shfl_xor(v_, off);
                if (lane < C8) red[(wave * RMAX + h) * DH + 8 * c8 + i] = v_;
            }
        }
    }
    __syncthreads();
    constexpr int NRED = VM == 0 ? PARTS : 4;
    for (int i = tid; i < R * DH; i += 256) {
        const int h = i / DH, e = i - h * DH;
        float o = 0.0f;
#pragma unroll
        for (int q = 0; q < NRED; ++q) o += red[(q * RMAX + h) * DH + e];
        st_act<true>(p.P + ((long) c * p.H + g * R + h) * DH + e, o);
    }
    if (tid < R) { st_act<true>(p.M + (g * R + tid) * p.nspan + c, mrun[tid]); st_act<true>(p.L + (g * R + tid) * p.nspan + c, lrun[tid]); }
    // ---- ticket: the last span of this KV head merges
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's write-through stores are acknowledged
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add((PM_G unsigned *) (p.ticket + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t + 1 == (unsigned) nact;
        if (last) __hip_atomic_store((PM_G unsigned *) (p.ticket + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everybody has arrived: ready for the next launch
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    // merge. The partials were written write-through by other workgroups (other XCDs) in THIS launch: one agent-scope acquire per wave
    // drops whatever stale lines of the scratch this CU / XCD still holds from earlier layers, then plain vector loads (device-coherent
    // dword loads serialise: 256 dependent round trips per thread at 64 spans)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int i = tid; i < R * nact; i += 256) {
        const int h = i / nact, cc = i - h * nact;
        sc[h][cc] = ld_g(p.M + (g * R + h) * p.nspan + cc);
        red[h * CK + cc] = ld_g(p.L + (g * R + h) * p.nspan + cc);
    }
    __syncthreads();
    if (tid < R) {
        float m = -INFINITY;
        for (int cc = 0; cc < nact; ++cc) m = fmaxf(m, sc[tid][cc]);
        float den = 0.0f;
        for (int cc = 0; cc < nact; ++cc) {
            const float mc = sc[tid][cc];
            const float f = mc == -INFINITY ? 0.0f : __expf(mc - m);
            den += f * red[tid * CK + cc];
            sc[tid][cc] = f;
        }
        const float inv = den > 0.0f ? 1.0f / den : 0.0f;
        for (int cc = 0; cc < nact; ++cc) sc[tid][cc] *= inv;
    }
    __syncthreads();
    for (int i = tid; i < R * DH / 4; i += 256) {
        const int h = i / (DH / 4), e4 = i - h * (DH / 4);
        const int hh = g * R + h;
        const float4 * src = (const float4 *) (p.P + (long) hh * DH) + e4;
        const long stride4 = (long) p.H * DH / 4;
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 8
        for (int cc = 0; cc < nact; ++cc) {
            const float4 v4 = ld_g(src + cc * stride4);
            const float w = sc[h][cc];
            a.x += w * v4.x; a.y += w * v4.y; a.z += w * v4.z; a.w += w * v4.w;
        }
        *(float4 *) (p.out + (long) hh * DH + 4 * e4) = a;
    }
}

} // namespace

// scratch floats: M, L [H][nspan_max], P [nspan_max][H][dh], tickets [Hkv] (the caller zeroes the scratch once after allocating it)
size_t pm_attn_flash_scratch_floats(int H, int Hkv, int dh, int n_ctx) {
    const size_t nspan = (size_t) (n_ctx + CK - 1) / CK;
    return 2 * (size_t) H * nspan + nspan * H * dh + (size_t) Hkv + 64 + 64 + (size_t) H * dh;      // (+ the Q8_0 path's query rows: pm_attn_flash_qrot_offset)
}
size_t pm_attn_flash_qrot_offset(int H, int Hkv, int dh, int n_ctx) {
    const size_t nspan = (size_t) (n_ctx + CK - 1) / CK;
    return 2 * (size_t) H * nspan + nspan * H * dh + (size_t) Hkv + 64 + 64;      // behind the tickets of both flash kernels (attn_flash_mfma.hip: + Hkv + 16 .. + 48)
}

// q, k, v = the RAW projections of the token; RoPE, KV store and attention over cells [0, n_kv) happen in this one launch.
// max_cells bounds the cells attended (sizes the grid: spans beyond n_kv exit at once); 0 = n_ctx. -1: unsupported shape.
int pm_launch_attn_flash(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * pos0, const int32_t * seq,
                         long seq_stride, const float * freq_factors, float * out, float * scratch, int H, int Hkv, int dh, int n_ctx,
                         float scale, const pm_rope_cfg & c, hipStream_t st, const int32_t * dyn, const void * mask, int v_rowmajor, int mask_f16,
                         int max_cells) {
    if ((dh != 64 && dh != 128) || H % Hkv || H / Hkv > RMAX || n_ctx % 8 || !scratch) return -1;
    const int nspan_max = (n_ctx + CK - 1) / CK;
    const int cells = max_cells > 0 && max_cells < n_ctx ? max_cells : n_ctx;
    // span = keys per workgroup: 128 up to 8k cells, then grown so that a KV head never has more than 64 partials to merge
    int span = CK;
    while ((cells + span - 1) / span > 64) span *= 2;
    FlashP p = {};
    p.q = q; p.k = k; p.v = v; p.kc = (uint16_t *) kc; p.vc = (uint16_t *) vc; p.pos0_ptr = pos0; p.seq_ptr = seq; p.seq_stride = seq_stride;
    p.ff = freq_factors; p.out = out;
    p.M = scratch; p.L = p.M + (size_t) H * nspan_max; p.P = p.L + (size_t) H * nspan_max; p.ticket = (unsigned *) (p.P + (size_t) nspan_max * H * dh);
    p.H = H; p.Hkv = Hkv; p.n_ctx = n_ctx; p.nspan = nspan_max; p.span = span; p.scale = scale;
    p.r.n_dims = c.n_dims; p.r.mode = c.mode; p.r.n_ctx_orig = c.n_ctx_orig; p.r.theta_scale = c.theta_scale; p.r.freq_scale = c.freq_scale;
    p.r.ext_factor = c.ext_factor; p.r.attn_factor = c.attn_factor; p.r.corr0 = c.corr0; p.r.corr1 = c.corr1;
    p.dyn = dyn; p.mask = mask; p.mask_f16 = mask_f16; p.vm = v_rowmajor ? 1 : 0;
    const dim3 grid((cells + span - 1) / span, Hkv);
    if (dh == 128) {
        if (p.vm) hipLaunchKernelGGL((attn_flash_kernel<128, 1>), grid, dim3(256), 0, st, p);
        else      hipLaunchKernelGGL((attn_flash_kernel<128, 0>), grid, dim3(256), 0, st, p);
    } else {
        if (p.vm) hipLaunchKernelGGL((attn_flash_kernel<64, 1>), grid, dim3(256), 0, st, p);
        else      hipLaunchKernelGGL((attn_flash_kernel<64, 0>), grid, dim3(256), 0, st, p);
    }
    return 0;
}
