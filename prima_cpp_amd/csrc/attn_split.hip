// attn_split.hip — single-token decode attention for LONG contexts (n_kv >~ 512): the keys are split over many workgroups.
//
// The fused decode kernel (attn_device.h) gives every query head ONE workgroup: perfect while the kernel is latency-bound
// (7 us at n_kv ~ 50), but a CU pulls ~25 GB/s, so at n_kv = 3800 the 1.9 MB of K/V per head cost 110 us per layer - 8.8 ms
// per Llama-3-70B token, more than streaming all the weights. Here one workgroup = (KV head, chunk of 256 keys) serves the
// whole group of query heads that shares the KV head (K / V bytes are read once per group, not once per head), over
// n_ctx/256 x n_head_kv workgroups:
//   1. scores : s[h][key] = scale * K[key] . q_h  for the chunk -> global score buffer; per (head, chunk) max and sum(exp)
//   2. pv     : global max / sum from the chunk statistics; p = exp(s - max) / sum rounded to F16 (same rounding point as
//               the reference, ggml.c:13783-13879 + the F16 conversion of mul_mat's src1); partial O = V^T[:, chunk] . p
//   3. combine: out[h][e] = sum over chunks of the partial O, fixed order (deterministic)
// preceded by the ordinary rope + KV-store kernel. 4 launches instead of 1, so the engine switches to this path only
// beyond PM355_ATTN_SPLIT_MIN positions (default 1024 = the measured crossover: the 4 launches cost ~38 us per layer whatever
// the context, the fused kernel 7 us + 27 ns per position; the host mirrors the device position counters to pick the captured
// graph). Llama-3-70B decode at 3.8k context: 57.6 -> 85.4 tok/s. Rounding points as in the fused kernel: q and p -> F16, F16 K / V, f32 accumulation; differences to it are
// summation order only (the sum of exponentials is assembled from per-chunk sums).
#include "pm355_device.h"
#include "pm355_layer_ops.h"

namespace {

constexpr int CK = 256;                          // keys per workgroup
constexpr int RMAX = 8;                          // query heads per KV head (GQA group) handled by one workgroup

struct SplitP {
    const float * q; const uint16_t * kc; const uint16_t * vc; const int32_t * pos0_ptr; const int32_t * seq_ptr; long seq_stride;
    float * out; float * S; float * M; float * L; float * P;     // scratch: S[H][n_ctx], M / L[H][nchunk], P[nchunk][H][dh]
    int H, Hkv, n_ctx, nchunk; float scale;
};

__device__ __forceinline__ int cur_pos(const SplitP & p, int & seq) {
    seq = p.seq_ptr ? *p.seq_ptr : 0;
    return p.pos0_ptr[seq];
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
    int seq; const int n_kv = cur_pos(p, seq) + 1;
    const int k0 = c * CK;
    if (k0 >= n_kv) return;
    const uint16_t * kc = p.kc + (long) seq * p.seq_stride + (long) g * DH;
    for (int i = tid; i < R * DH; i += 256) { const int h = i / DH, e = i - h * DH; qs[h][e] = h2f(f2h(p.q[(long) (g * R + h) * DH + e])); }
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
        float acc[RMAX];
#pragma unroll
        for (int h = 0; h < RMAX; ++h) {
            acc[h] = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[h] = fmaf(kf[i], qr[h][i], acc[h]);
        }
#pragma unroll
        for (int h = 0; h < RMAX; ++h) acc[h] = group_sum<LPK>(acc[h]);
#pragma unroll
        for (int h = 0; h < RMAX; ++h) if (h < R && piece == (h % LPK)) sc[h][kin] = key < n_kv ? acc[h] * p.scale : -INFINITY;
    }
    __syncthreads();
    // chunk statistics and the scores themselves: wave w handles heads w, w + 4
    for (int h = wave; h < R; h += 4) {
        float m = -INFINITY;
        for (int i = lane; i < CK; i += 64) m = fmaxf(m, sc[h][i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float l = 0.0f;
        for (int i = lane; i < CK; i += 64) { const float s = sc[h][i]; l += expf(s - m); if (k0 + i < n_kv) p.S[(long) (g * R + h) * p.n_ctx + k0 + i] = s; }
        l = wave_sum(l);
        if (lane == 0) { p.M[(g * R + h) * p.nchunk + c] = m; p.L[(g * R + h) * p.nchunk + c] = l; }
    }
}

// ---- 2. probabilities and partial P.V ---------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_split_pv_kernel(SplitP p) {
    constexpr int PARTS = 256 / DH;              // key sub-ranges per workgroup (2 for dh 128, 4 for dh 64)
    constexpr int KP = CK / PARTS;               // keys per sub-range
    __shared__ float pl[RMAX][CK];               // F16-rounded probabilities of this chunk
    __shared__ float mx[RMAX], inv[RMAX];
    __shared__ float part[PARTS][RMAX][DH];
    const int c = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int R = p.H / p.Hkv;
    int seq; const int n_kv = cur_pos(p, seq) + 1;
    const int k0 = c * CK;
    if (k0 >= n_kv) return;
    const int nact = (n_kv + CK - 1) / CK;
    if (tid < R) {
        const int hh = g * R + tid;
        float m = -INFINITY;
        for (int i = 0; i < nact; ++i) m = fmaxf(m, p.M[hh * p.nchunk + i]);
        double l = 0.0;                          // sum in double like the reference (ggml_float), chunk by chunk
        for (int i = 0; i < nact; ++i) l += (double) (p.L[hh * p.nchunk + i] * expf(p.M[hh * p.nchunk + i] - m));
        mx[tid] = m; inv[tid] = (float) (1.0 / l);
    }
    __syncthreads();
    for (int i = tid; i < R * CK; i += 256) {
        const int h = i / CK, kin = i - h * CK;
        const int key = k0 + kin;
        pl[h][kin] = key < n_kv ? h2f(f2h(expf(p.S[(long) (g * R + h) * p.n_ctx + key] - mx[h]) * inv[h])) : 0.0f;
    }
    __syncthreads();
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
    int seq; const int n_kv = cur_pos(p, seq) + 1;
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

// q: this token's ROTATED queries [H*dh] f32 (rope_kv_store output); the caches already hold the token's K row / V column.
int pm_launch_attn_split(const float * q, const void * kc, const void * vc, const int32_t * pos0, const int32_t * seq, long seq_stride,
                         float * out, float * scratch, int H, int Hkv, int dh, int n_ctx, float scale, hipStream_t st) {
    if ((dh != 64 && dh != 128) || H % Hkv || H / Hkv > RMAX || n_ctx % 8 || !scratch) return -1;
    const int nchunk = (n_ctx + CK - 1) / CK;
    SplitP p;
    p.q = q; p.kc = (const uint16_t *) kc; p.vc = (const uint16_t *) vc; p.pos0_ptr = pos0; p.seq_ptr = seq; p.seq_stride = seq_stride;
    p.out = out; p.S = scratch; p.M = p.S + (size_t) H * n_ctx; p.L = p.M + (size_t) H * nchunk; p.P = p.L + (size_t) H * nchunk;
    p.H = H; p.Hkv = Hkv; p.n_ctx = n_ctx; p.nchunk = nchunk; p.scale = scale;
    const dim3 grid(nchunk, Hkv);
    if (dh == 128) {
        hipLaunchKernelGGL(attn_split_scores_kernel<128>, grid, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_split_pv_kernel<128>, grid, dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL(attn_split_scores_kernel<64>, grid, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_split_pv_kernel<64>, grid, dim3(256), 0, st, p);
    }
    hipLaunchKernelGGL(attn_split_combine_kernel, dim3((H * dh + 255) / 256), dim3(256), 0, st, p, dh);
    return 0;
}
