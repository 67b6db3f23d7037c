// attn_cached.hip — single-token attention over cells that are ALL in the cache, for the decode path whose wq | wk | wv launch has
// already applied RoPE, rounded q to F16 and stored the token's K row / V column (QkvEpi, mmvq_device.h), plus the per-token
// cos / sin table that launch reads.
//
// Replaces (with the QKV epilogue) the same reference nodes as attn_rope_fused_kernel: ROPE x2 (ggml.c:14143), CPY F32->F16 x2
// (llm_build_kv_store, src/llama.cpp:9673-9718), MUL_MAT(K, q) with q converted to F16 (ggml.c:12445-12473), SOFT_MAX_EXT
// (ggml.c:13783-13879), MUL_MAT(V^T, p) with p converted to F16, same rounding points.
//
// Why a second kernel: at short contexts the fused kernel is a chain of dependent latencies (position -> rope -> LDS -> barrier ->
// 128-term dot per THREAD -> two block reductions -> P.V -> reduction): 7.4 us per 70B layer for ~0 bytes. Here
//   * up to 64 cells: wave w owns head dimensions [DH/4 w, DH/4 (w+1)), lane = key. All loads (its K slices, its V^T chunks, q through
//     the scalar cache) are issued before anything is known about the position; ONE workgroup barrier (the four partial dot products of
//     a key meet in LDS); max / sum / probabilities are wave reductions that every wave repeats for itself; each wave finishes its own
//     quarter of the output.
//   * more cells (up to the long-context threshold): the per-head body of attn_device.h in its CACHED form.
#include "attn_device.h"
#include "attn_tail_device.h"
#include "pm355_layer_ops.h"

namespace {

// this token's (cos, sin) per rotation pair: one launch per token serves every layer's wq | wk | wv epilogue
__global__ __launch_bounds__(128) void rope_table_kernel(RopeP r, const int32_t * pos_ptr, const int32_t * seq_ptr, const float * freq_factors, float * tab) {
    const int seq = seq_ptr ? *seq_ptr : 0;
    const int pos = pos_ptr[seq] + (int) blockIdx.x;             // (blockIdx.x: token of a small batch, its table n_dims floats further on)
    tab += (size_t) blockIdx.x * r.n_dims;
    for (int pair = threadIdx.x; pair < r.n_dims / 2; pair += blockDim.x) {
        float c, s;
        rope_cs(r, (float) pos, pair, freq_factors, c, s);
        tab[2 * pair] = c; tab[2 * pair + 1] = s;
    }
}

__device__ __forceinline__ double wave_sum_f64_shfl(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

template <int DH, int VM>
__global__ __launch_bounds__(256) void attn_cached_kernel(AttnP a_in) {
    // grid.y = tokens of a small batch (engine: 2..64-token steps after rope_kv_store has rotated / rounded q and stored every token's K / V):
    // token t attends cells [0, pos0 + t], its q / out rows are t * H * DH further on
    AttnP a = a_in;
    a.tok = (int) blockIdx.y;
    a.q += (long) a.tok * a.H * DH; a.out += (long) a.tok * a.H * DH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float redf[8];
    __shared__ double redd[4];
    __shared__ float part[4][256];             // (short path: [4][64]; keys-in-lanes form up to 256 cells: [4][64 KPL])
    __shared__ float pw[4][256];
    const int h = blockIdx.x;
    constexpr bool SHORT_OK = VM == 0 && (DH == 64 || DH == 128 || DH == 256);
    if (SHORT_OK) {
        constexpr int DPW = DH / 4;            // head dimensions per wave
        constexpr int NK = DPW / 8;            // 16-byte pieces of a key's slice
        constexpr int KP = 64 / DPW;           // key parts of the P.V lanes: lane = (dimension e = lane % DPW, part = lane / DPW)
        constexpr int KPP = 64 / KP;           // keys per part (== DPW)
        constexpr int NV = KPP / 8;            // 16-byte pieces of a V^T chunk
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int H = a.H, Hkv = a.Hkv, n_ctx = a.n_ctx;
        const int hk = h / (H / Hkv);
        unsigned long long tsv[6] = {PM_TS_NOW(), 0, 0, 0, 0, 0};
        // ---- position / cells attended / slab (scalar cache), then every load of the short path
        int seq = 0, n_kv;
        if (a.dyn) n_kv = uniform_const_ptr(a.dyn)[1];
        else {
            seq = a.seq_ptr ? uniform_const_ptr(a.seq_ptr)[0] : 0;
            n_kv = uniform_const_ptr(a.pos0_ptr)[seq] + 1 + a.tok;
        }
        // (short path: a lane owns a key and reads whole V^T chunks of KPP cells - only where the cache holds at least 64 cells per row; a smaller
        //  n_ctx takes the general body below, whose loads are bounded by n_kv: ADVICE r3)
        if (n_kv <= 64 && n_ctx >= 64) {
            const PM_G uint16_t * kc = (const PM_G uint16_t *) a.kc + (long) seq * a.seq_stride;
            const PM_G uint16_t * vc = (const PM_G uint16_t *) a.vc + (long) seq * a.seq_stride;
            const int key = lane < n_ctx ? lane : 0;
            const PM_G uint16_t * kr = kc + (long) key * Hkv * DH + (long) hk * DH + DPW * wave;
            u32x4 kreg[NK], vreg[NV];
#pragma unroll
            for (int j = 0; j < NK; ++j) kreg[j] = *(const PM_G u32x4 *) (kr + 8 * j);
            const int e = lane % DPW, kp = lane / DPW;
            const PM_G uint16_t * vr = vc + (long) (hk * DH + DPW * wave + e) * n_ctx + (KPP * kp + KPP <= n_ctx ? KPP * kp : 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) vreg[j] = *(const PM_G u32x4 *) (vr + 8 * j);
            const float * qw = uniform_const_ptr(a.q + (long) h * DH + DPW * wave);      // this wave's q slice: scalar loads
            const float m_add = lane < n_kv ? attn_mask_at(a.mask, a.mask_f16, lane) : 0.0f;
            // ---- partial dot products of key `lane` over this wave's dimensions
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NK; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc = fmaf(h2f((uint16_t) (kreg[j][t] & 0xFFFF)), qw[8 * j + 2 * t], acc);
                    acc = fmaf(h2f((uint16_t) (kreg[j][t] >> 16)), qw[8 * j + 2 * t + 1], acc);
                }
            part[wave][lane] = acc;
            __syncthreads();                       // the only workgroup barrier of this path
            tsv[1] = PM_TS_NOW();
            const bool valid = lane < n_kv;
            const float s_ = valid ? ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) * a.scale + m_add : -INFINITY;
            const float mx = wave_max_all(s_);
            const float ex = valid ? expf(s_ - mx) : 0.0f;
            const double tot = wave_sum_f64_shfl((double) ex);
            const float inv = (float) (1.0 / tot);
            pw[wave][lane] = h2f(f2h(ex * inv));   // p rounded to F16 (src1 of the V^T.p product)
            __builtin_amdgcn_wave_barrier();        // (a wave's LDS accesses execute in order: no workgroup barrier needed for its own copy)
            tsv[2] = PM_TS_NOW();
            // ---- P.V: lane (e, kp) sums keys [KPP kp, KPP kp + KPP) of its V^T row; parts meet through shuffles
            float o = 0.0f;
            const bool chunk_ok = KPP * kp + KPP <= n_ctx;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    o = fmaf(h2f((uint16_t) (vreg[j][t] & 0xFFFF)), pw[wave][KPP * kp + 8 * j + 2 * t], o);
                    o = fmaf(h2f((uint16_t) (vreg[j][t] >> 16)), pw[wave][KPP * kp + 8 * j + 2 * t + 1], o);
                }
            if (!chunk_ok) o = 0.0f;               // (n_ctx < 64: that chunk was clamped to another one and holds no attended cell)
#pragma unroll
            for (int off = DPW; off < 64; off <<= 1) o += __shfl_xor(o, off);
            if (lane < DPW) st_g(a.out + (long) h * DH + DPW * wave + lane, o);
            tsv[5] = PM_TS_NOW();
            pm_ts_store(a.ts, 3, tsv);
            return;
        }
        // up to 256 cells: keys in the lanes, two / four per lane (attn_tail_device.h) instead of the general body's score buffer and three reductions
        if constexpr (DH <= 128) {
            const uint16_t * kc = a.kc + (long) seq * a.seq_stride, * vc = a.vc + (long) seq * a.seq_stride;
            auto bar = [&]() __attribute__((always_inline)) { __syncthreads(); };
            if (n_kv <= 128 && n_ctx >= 128) {
                attn_keys_in_lanes<DH, 2, false>(a.q + (long) h * DH, kc, vc, a.out + (long) h * DH, hk, Hkv, n_ctx, n_kv, a.scale, a.mask, a.mask_f16, &part[0][0], &pw[0][0], nullptr, lane, wave, bar);
                return;
            }
            if (n_kv <= 256 && n_ctx >= 256) {
                attn_keys_in_lanes<DH, 4, false>(a.q + (long) h * DH, kc, vc, a.out + (long) h * DH, hk, Hkv, n_ctx, n_kv, a.scale, a.mask, a.mask_f16, &part[0][0], &pw[0][0], nullptr, lane, wave, bar);
                return;
            }
        }
    }
    attn_rope_body<DH, false, VM, true>(a, h, smem, redf, redd);
}

} // namespace

void pm_launch_rope_table(const pm_rope_cfg & c, const int32_t * pos, const int32_t * seq, const float * freq_factors, float * tab, hipStream_t st, int n_tok) {
    RopeP r;
    r.n_dims = c.n_dims; r.mode = c.mode; r.n_ctx_orig = c.n_ctx_orig; r.theta_scale = c.theta_scale;
    r.freq_scale = c.freq_scale; r.ext_factor = c.ext_factor; r.attn_factor = c.attn_factor; r.corr0 = c.corr0; r.corr1 = c.corr1;
    hipLaunchKernelGGL(rope_table_kernel, dim3(n_tok < 1 ? 1 : n_tok), dim3(128), 0, st, r, pos, seq, freq_factors, tab);
}

// q = rotated, F16-rounded query rows; caches already hold this token. Same arguments as pm_launch_attn_rope_fused otherwise.
int pm_launch_attn_cached(const float * q, void * kc, void * vc, const int32_t * pos0, const int32_t * seq, long seq_stride, float * out,
                          int H, int Hkv, int dh, int n_ctx, float scale, hipStream_t st, const int32_t * dyn, const void * mask,
                          int max_keys, int v_rowmajor, int mask_f16, int n_tok) {
    if ((dh != 64 && dh != 128 && dh != 256) || n_ctx % 8 || n_tok < 1 || (n_tok > 1 && dyn)) return -1;
    const size_t lds = (size_t) (4 * dh + (v_rowmajor ? 2048 : 256) + (max_keys > 0 ? ((max_keys + 7) & ~7) : n_ctx) + 8) * 4;
    if (lds > 150 * 1024) return -1;
    RopeP r = {};
    if (!pos0) pos0 = dyn;                       // ggml-graph mode: cells come from dyn; keep every pointer the kernel may touch valid
    if (!pos0) return -1;
    auto launch = [&](auto kern) {
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        AttnP a = {q, nullptr, nullptr, (uint16_t *) kc, (uint16_t *) vc, pos0, seq, seq_stride, nullptr, out, H, Hkv, n_ctx, scale, r, dyn,
                   (const float *) mask, mask_f16, pm_ts_next_slot()};
        hipLaunchKernelGGL(kern, dim3(H, n_tok), dim3(256), lds, st, a);
    };
    if (v_rowmajor) {
        if (dh == 64) launch(attn_cached_kernel<64, 1>);
        else if (dh == 128) launch(attn_cached_kernel<128, 1>);
        else launch(attn_cached_kernel<256, 1>);
    } else if (dh == 64) launch(attn_cached_kernel<64, 0>);
    else if (dh == 128) launch(attn_cached_kernel<128, 0>);
    else launch(attn_cached_kernel<256, 0>);
    return 0;
}
