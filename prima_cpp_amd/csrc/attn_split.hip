// attn_split.hip — single-token decode attention for LONG contexts (n_kv >~ 512): the keys are split over many workgroups.
//
// The fused decode kernel (attn_device.h) gives every query head ONE workgroup: perfect while the kernel is latency-bound
// (7 us at n_kv ~ 50), but a CU pulls ~25 GB/s, so at n_kv = 3800 the 1.9 MB of K/V per head cost 110 us per layer - 8.8 ms
// per Llama-3-70B token, more than streaming all the weights. Here one workgroup = (KV head, chunk of 128 keys) serves the
// whole group of query heads that shares the KV head (K / V bytes are read once per group, not once per head), over
// n_ctx/128 x n_head_kv workgroups:
//   1. scores : s[h][key] = scale * K[key] . q_h  for the chunk -> global score buffer; per (head, chunk) max and sum(exp)
//   2. pv     : global max / sum from the chunk statistics; p = exp(s - max) / sum rounded to F16 (same rounding point as
//               the reference, ggml.c:13783-13879 + the F16 conversion of mul_mat's src1); partial O = V^T[:, chunk] . p
//   3. combine: out[h][e] = sum over chunks of the partial O, fixed order (deterministic)
// RoPE and the KV store of the new token are done by the scores kernel (every workgroup rotates its group's queries; the one
// whose chunk holds the token's position also rotates k, uses it from LDS and writes the K row / V column). 3 launches
// instead of 1, so the engine switches to this path only beyond PM355_ATTN_SPLIT_MIN positions (default 320 since round 5, engine.hip; 640 was the measured
// crossover: the 3 launches cost ~25 us per layer whatever the context, the fused kernel 7 us + 27 ns per position; the host
// mirrors the device position counters to pick the captured graph). Llama-3-70B decode at 3.8k context: 57.6 -> 92 tok/s. Rounding points as in the fused kernel: q and p -> F16, F16 K / V, f32 accumulation; differences to it are
// summation order only (the sum of exponentials is assembled from per-chunk sums).
#include "attn_device.h"
#include "pm355_layer_ops.h"

namespace {

constexpr int CK = 128;                          // keys per workgroup (256: half the workgroups, 1.4x the latency chain per workgroup)
constexpr int RMAX = 8;                          // query heads per KV head (GQA group) handled by one workgroup

struct SplitP {
    const float * q; const uint16_t * kc; const uint16_t * vc; const int32_t * pos0_ptr; const int32_t * seq_ptr; long seq_stride;
    float * out; float * S; float * M; float * L; float * P;     // scratch: S[H][n_ctx], M / L[H][nchunk], P[nchunk][H][dh]
    int H, Hkv, n_ctx, nchunk; float scale;
    // do_rope: q is the RAW projection; the scores kernel rotates it (every workgroup, its own group's heads), and the
    // workgroup whose chunk contains the token's position also rotates k, takes it from LDS and stores the K row / V column
    const float * k; const float * v; const float * ff; uint16_t * kc_w; uint16_t * vc_w; RopeP r; int do_rope;
    // ggml-graph mode (see AttnP in attn_device.h): dyn = {cache cell of the token, cells attended}, mask = additive f32 KQ mask row
    const int32_t * dyn; const void * mask; int mask_f16;
    int vm;                                      // V cache layout: 0 transposed, 1 row-major (flash-attention graphs; P stays f32) - attn_device.h
};

// rope position `pos`, cache cell `slot` the token is stored in, cells attended = [0, n_kv)
__device__ __forceinline__ void cur_pos(const SplitP & p, int & seq, int & pos, int & slot, int & n_kv) {
    seq = p.seq_ptr ? *p.seq_ptr : 0;
    pos = p.pos0_ptr[seq]; slot = pos; n_kv = pos + 1;
    if (p.dyn) { slot = p.dyn[0]; n_kv = p.dyn[1]; }
}

// sum over the LPK (8 or 16) consecutive lanes that share one key; result in every lane of the group
template <int LPK> __device__ __forceinline__ float group_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v);
    if (LPK == 16) v += dpp_f<0x140>(v);
    return v;
}

// ---- 1. scores ------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_split_scores_kernel(SplitP p) {
    constexpr int LPK = DH / 8;                  // lanes per key (16 bytes of the K row each)
    constexpr int KPP = 256 / LPK;               // keys per pass
    __shared__ float qs[RMAX][DH];               // F16-rounded q of the group's heads
    __shared__ float sc[RMAX][CK];               // this chunk's scores
    const int c = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = p.H / p.Hkv;
    int seq, pos, slot, n_kv; cur_pos(p, seq, pos, slot, n_kv);
    const int k0 = c * CK;
    if (k0 >= n_kv) return;
    const uint16_t * kc = p.kc + (long) seq * p.seq_stride + (long) g * DH;
    __shared__ float kcur[DH];                   // the token's own rotated key (do_rope)
    if (!p.do_rope) {
        for (int i = tid; i < R * DH; i += 256) { const int h = i / DH, e = i - h * DH; qs[h][e] = h2f(f2h(p.q[(long) (g * R + h) * DH + e])); }
    } else {
        // same rotation as attn_rope_fused_kernel / rope_kv_store_kernel (ggml_compute_forward_rope_f32, ggml.c:14143). cos / sin
        // depend on (position, pair) only: ONE rope_cs per thread (its sinf / cosf take the slow large-argument path at long
        // positions), shared by all heads through LDS
        __shared__ float cs2[DH / 2][2];
        const bool neox = p.r.mode & 2;
        const int half = p.r.n_dims / 2;
        if (tid < DH / 2) {
            float cs_ = 1.0f, sn_ = 0.0f;
            if (tid < half) rope_cs(p.r, (float) pos, tid, p.ff, cs_, sn_);
            cs2[tid][0] = cs_; cs2[tid][1] = sn_;
        }
        __syncthreads();
        const bool mine = c == slot / CK;        // this workgroup also owns the new K row / V column of its KV head
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
                uint16_t * d = p.kc_w + (long) seq * p.seq_stride + (long) slot * p.Hkv * DH + (long) g * DH;
                d[ia] = h0; d[ib] = h1;
            }
        }
        if (mine) for (int e = tid; e < DH; e += 256) {
            const uint16_t hv = f2h(p.v[(long) g * DH + e]);
            if (p.vm == 0) p.vc_w[(long) seq * p.seq_stride + (long) (g * DH + e) * p.n_ctx + slot] = hv;
            else           p.vc_w[(long) seq * p.seq_stride + (long) slot * p.Hkv * DH + (long) g * DH + e] = hv;
        }
    }
    __syncthreads();
    const int piece = tid % LPK, kslot = tid / LPK;
    const long krow = (long) p.Hkv * DH;
    constexpr int NPS = CK / KPP;                // passes (16 for dh 128, 8 for dh 64)
    // all K pieces of this thread are requested up front: one exposed memory latency instead of one per pass
    u32x4 kall[NPS];
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) kall[ps] = *(const u32x4 *) (kc + (long) min(k0 + ps * KPP + kslot, p.n_ctx - 1) * krow + 8 * piece);
    // this thread's slice of every head's q stays in registers for all passes; the heads are independent FMA / DPP chains
    // (only 2 waves per CU run here: instruction-level parallelism is what hides the ALU latency)
    float qr[RMAX][8];
#pragma unroll
    for (int h = 0; h < RMAX; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) qr[h][i] = h < R ? qs[h][8 * piece + i] : 0.0f;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
        const int kin = ps * KPP + kslot, key = k0 + kin;
        const u32x4 kk = kall[ps];
        float kf[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { kf[2 * j] = h2f((uint16_t) (kk[j] & 0xFFFF)); kf[2 * j + 1] = h2f((uint16_t) (kk[j] >> 16)); }
        if (p.do_rope && key == slot) {              // the token's own key: from LDS, the cache row is being written by this kernel
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
    __syncthreads();
    // chunk statistics and the scores themselves: wave w handles heads w, w + 4
    for (int h = wave; h < R; h += 4) {
        float m = -INFINITY;
        for (int i = lane; i < CK; i += 64) m = fmaxf(m, sc[h][i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float l = 0.0f;
        for (int i = lane; i < CK; i += 64) {    // (a fully masked chunk has m = -inf: its sum is 0, not exp(nan))
            const float s = sc[h][i]; l += s == -INFINITY ? 0.0f : expf(s - m); if (k0 + i < n_kv) p.S[(long) (g * R + h) * p.n_ctx + k0 + i] = s; }
        l = wave_sum(l);
        if (lane == 0) { p.M[(g * R + h) * p.nchunk + c] = m; p.L[(g * R + h) * p.nchunk + c] = l; }
    }
}

// ---- 2. probabilities and partial P.V ---------------------------------------------------------------------------------
template <int DH, int VM>
__global__ __launch_bounds__(256) void attn_split_pv_kernel(SplitP p) {
    constexpr int PARTS = 256 / DH;              // key sub-ranges per workgroup (2 for dh 128, 4 for dh 64)
    constexpr int KP = CK / PARTS;               // keys per sub-range
    __shared__ float pl[RMAX][CK];               // F16-rounded probabilities of this chunk
    __shared__ float mx[RMAX], inv[RMAX];
    __shared__ float part[VM == 1 ? 4 : PARTS][RMAX][DH];
    const int c = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int R = p.H / p.Hkv;
    int seq, pos_, slot_, n_kv; cur_pos(p, seq, pos_, slot_, n_kv);
    const int k0 = c * CK;
    if (k0 >= n_kv) return;
    const int nact = (n_kv + CK - 1) / CK;
    // global max and sum of every head of the group from the chunk statistics: 32 lanes per head, one chunk each (+32, ...)
    {
        const int h = tid >> 5, ci = tid & 31;
        float m = -INFINITY;
        if (h < R) for (int i = ci; i < nact; i += 32) m = fmaxf(m, p.M[(g * R + h) * p.nchunk + i]);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        double l = 0.0;                          // sum in double like the reference (ggml_float)
        if (h < R) for (int i = ci; i < nact; i += 32) { const float mi = p.M[(g * R + h) * p.nchunk + i]; if (mi != -INFINITY) l += (double) (p.L[(g * R + h) * p.nchunk + i] * expf(mi - m)); }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) l += __shfl_xor(l, off);
        if (h < R && ci == 0) { mx[h] = m; inv[h] = (float) (1.0 / l); }
    }
    __syncthreads();
    for (int i = tid; i < R * CK; i += 256) {
        const int h = i / CK, kin = i - h * CK;
        const int key = k0 + kin;
        const float pr = key < n_kv ? expf(p.S[(long) (g * R + h) * p.n_ctx + key] - mx[h]) * inv[h] : 0.0f;
        pl[h][kin] = VM == 0 ? h2f(f2h(pr)) : pr;
    }
    __syncthreads();
    if (VM == 1) {
        // row-major V: thread (16-byte chunk c8, key slot ks) takes rows k0 + ks, + NSL, ... of this chunk for all R heads; the key slots
        // are folded with lane shuffles inside a wave and through LDS across the 4 waves
        constexpr int C8 = DH / 8, NSL = 256 / C8, NR = CK / NSL;
        const int c8 = tid % C8, ks = tid / C8, wave = tid >> 6;
        const uint16_t * vr = p.vc + (long) seq * p.seq_stride + (long) g * DH + 8 * c8;
        const long vrow = (long) p.Hkv * DH;
        u32x4 vall[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) vall[j] = *(const u32x4 *) (vr + (long) min(k0 + ks + j * NSL, p.n_ctx - 1) * vrow);
        float acc[RMAX][8];
#pragma unroll
        for (int h = 0; h < RMAX; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[h][i] = 0.0f;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int kin = ks + j * NSL;
            if (k0 + kin >= n_kv) continue;
            const u32x4 vv = vall[j];
            float vf[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { vf[2 * i] = h2f((uint16_t) (vv[i] & 0xFFFF)); vf[2 * i + 1] = h2f((uint16_t) (vv[i] >> 16)); }
#pragma unroll
            for (int h = 0; h < RMAX; ++h) if (h < R) {
                const float pr = pl[h][kin];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[h][i] = fmaf(vf[i], pr, acc[h][i]);
            }
        }
        float * red = &part[0][0][0];                                    // [4 waves][RMAX][DH]
#pragma unroll
        for (int h = 0; h < RMAX; ++h) if (h < R) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v_ = acc[h][i];
#pragma unroll
                for (int off = C8; off < 64; off <<= 1) v_ += __shfl_xor(v_, off);
                if ((tid & 63) < C8) red[(wave * RMAX + h) * DH + 8 * c8 + i] = v_;
            }
        }
        __syncthreads();
        for (int i = tid; i < R * DH; i += 256) {
            const int h = i / DH, ee = i - h * DH;
            const float o = (red[(0 * RMAX + h) * DH + ee] + red[(1 * RMAX + h) * DH + ee]) + (red[(2 * RMAX + h) * DH + ee] + red[(3 * RMAX + h) * DH + ee]);
            p.P[((long) c * p.H + g * R + h) * DH + ee] = o;
        }
        return;
    }
    // thread (e, part): V^T row e of this KV head, keys [part * KP, +KP) of the chunk, all R heads at once
    const int e = tid % DH, pt = tid / DH;
    const uint16_t * vr = p.vc + (long) seq * p.seq_stride + (long) (g * DH + e) * p.n_ctx;
    float acc[RMAX];
#pragma unroll
    for (int h = 0; h < RMAX; ++h) acc[h] = 0.0f;
    // all V pieces of this thread are requested up front (one exposed latency), then the FMAs
    constexpr int NV = KP / 8;
    u32x4 vall[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) vall[j] = *(const u32x4 *) (vr + min(k0 + pt * KP + 8 * j, p.n_ctx - 8));   // clamped (p is 0 beyond n_kv)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int kin = pt * KP + 8 * j;
        const u32x4 vv = vall[j];
        float vf[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { vf[2 * i] = h2f((uint16_t) (vv[i] & 0xFFFF)); vf[2 * i + 1] = h2f((uint16_t) (vv[i] >> 16)); }
        if (k0 + kin >= n_kv) continue;                                 // whole group beyond the context
#pragma unroll
        for (int h = 0; h < RMAX; ++h) if (h < R) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[h] = fmaf(vf[i], pl[h][kin + i], acc[h]);
        }
    }
    for (int h = 0; h < R; ++h) part[pt][h][e] = acc[h];
    __syncthreads();
    for (int i = tid; i < R * DH; i += 256) {
        const int h = i / DH, ee = i - h * DH;
        float o = 0.0f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) o += part[q][h][ee];
        p.P[((long) c * p.H + g * R + h) * DH + ee] = o;
    }
}

// ---- 3. combine ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_split_combine_kernel(SplitP p, int dh) {
    int seq, pos_, slot_, n_kv; cur_pos(p, seq, pos_, slot_, n_kv);
    const int nact = (n_kv + CK - 1) / CK;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.H * dh) return;
    float o = 0.0f;
    for (int c = 0; c < nact; ++c) o += p.P[(long) c * p.H * dh + i];
    p.out[i] = o;
}

} // namespace

size_t pm_attn_split_scratch_floats(int H, int dh, int n_ctx) {
    const size_t nchunk = (size_t) (n_ctx + CK - 1) / CK;
    return (size_t) H * n_ctx + 2 * (size_t) H * nchunk + nchunk * H * dh;
}

// rope == nullptr: q = this token's ROTATED queries [H*dh] f32 and the caches already hold its K row / V column (k, v unused).
// rope != nullptr: q, k, v = the RAW projections; the rotation, the KV store and the attention are done here (3 launches).
int pm_launch_attn_split(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * pos0, const int32_t * seq,
                         long seq_stride, const float * freq_factors, float * out, float * scratch, int H, int Hkv, int dh, int n_ctx,
                         float scale, const pm_rope_cfg * rope, hipStream_t st, const int32_t * dyn, const void * mask, int v_rowmajor, int mask_f16) {
    if ((dh != 64 && dh != 128) || H % Hkv || H / Hkv > RMAX || n_ctx % 8 || !scratch) return -1;
    const int nchunk = (n_ctx + CK - 1) / CK;
    SplitP p = {};
    p.q = q; p.kc = (const uint16_t *) kc; p.vc = (const uint16_t *) vc; p.pos0_ptr = pos0; p.seq_ptr = seq; p.seq_stride = seq_stride;
    p.out = out; p.S = scratch; p.M = p.S + (size_t) H * n_ctx; p.L = p.M + (size_t) H * nchunk; p.P = p.L + (size_t) H * nchunk;
    p.H = H; p.Hkv = Hkv; p.n_ctx = n_ctx; p.nchunk = nchunk; p.scale = scale; p.dyn = dyn; p.mask = mask; p.mask_f16 = mask_f16; p.vm = v_rowmajor ? 1 : 0;
    if (rope) {
        const pm_rope_cfg & c = *rope;
        p.r.n_dims = c.n_dims; p.r.mode = c.mode; p.r.n_ctx_orig = c.n_ctx_orig; p.r.theta_scale = c.theta_scale;
        p.r.freq_scale = c.freq_scale; p.r.ext_factor = c.ext_factor; p.r.attn_factor = c.attn_factor; p.r.corr0 = c.corr0; p.r.corr1 = c.corr1;
        p.k = k; p.v = v; p.ff = freq_factors; p.kc_w = (uint16_t *) kc; p.vc_w = (uint16_t *) vc; p.do_rope = 1;
    }
    const dim3 grid(nchunk, Hkv);
    if (dh == 128) {
        hipLaunchKernelGGL(attn_split_scores_kernel<128>, grid, dim3(256), 0, st, p);
        if (p.vm) hipLaunchKernelGGL((attn_split_pv_kernel<128, 1>), grid, dim3(256), 0, st, p);
        else      hipLaunchKernelGGL((attn_split_pv_kernel<128, 0>), grid, dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL(attn_split_scores_kernel<64>, grid, dim3(256), 0, st, p);
        if (p.vm) hipLaunchKernelGGL((attn_split_pv_kernel<64, 1>), grid, dim3(256), 0, st, p);
        else      hipLaunchKernelGGL((attn_split_pv_kernel<64, 0>), grid, dim3(256), 0, st, p);
    }
    hipLaunchKernelGGL(attn_split_combine_kernel, dim3((H * dh + 255) / 256), dim3(256), 0, st, p, dh);
    return 0;
}
