// quantize.hip — activation quantizers for the decode path (gfx950).
//
//   f32 row -> Q8_K  (reference: quantize_row_q8_K_ref, ggml/src/ggml-quants.c:3785-3826)
//   f32 row -> Q8_0  (reference: quantize_row_q8_0_ref, ggml/src/ggml-quants.c:848-871)
//   rms_norm(x) * w -> f32 and/or Q8_K in one pass
//              (reference: ggml_compute_forward_rms_norm_f32 ggml.c:11950 + ggml_compute_forward_mul_f32 :10077)
//
// Bit-exact with the reference for q / bsums / d: same float operations in the same order per
// element (iscale = -127/max; nearest_int(iscale*x); d = 1/iscale), and the same "first element with
// the largest |x| decides the sign" rule, implemented as (wave max) + (wave min over candidate index).
//
// Device layout of a quantized activation row ("row-SoA", internal to this library; K values):
//   Q8_K : int8 qs[K] | float d[K/256] | int16 bsums[K/16]       -> pm_q8k_row_bytes(K)
//   Q8_0 : int8 qs[K] | half  d[K/32]                            -> pm_q80_row_bytes(K)
// HBM-bound, trivially small (K <= 32k): one wave per 256-value block, 4 values per lane (16-B loads).
#include "pm355_device.h"
#include "pm355_kernels.h"

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_keep_f(float v) {   // lanes outside ROW_MASK / out of range keep v
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_keep_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_keep_f<0xB1>(v));
    v = fmaxf(v, dpp_keep_f<0x4E>(v));
    v = fmaxf(v, dpp_keep_f<0x141>(v));
    v = fmaxf(v, dpp_keep_f<0x140>(v));
    v = fmaxf(v, dpp_keep_f<0x142, 0xA>(v));
    v = fmaxf(v, dpp_keep_f<0x143, 0xC>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_keep_i<0xB1>(v));
    v = min(v, dpp_keep_i<0x4E>(v));
    v = min(v, dpp_keep_i<0x141>(v));
    v = min(v, dpp_keep_i<0x140>(v));
    v = min(v, dpp_keep_i<0x142, 0xA>(v));
    v = min(v, dpp_keep_i<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// One wave quantizes one 256-value block held as 4 consecutive values per lane.
__device__ __forceinline__ void q8k_block_from_regs(const float v[4], int lane, uint8_t * row_out, int K, int blk, const pm_q8k_tables & tb = pm_q8k_tables(), int row = 0) {
    float a0 = fabsf(v[0]), a1 = fabsf(v[1]), a2 = fabsf(v[2]), a3 = fabsf(v[3]);
    const float amax = wave_max(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
    int8_t  * qs    = (int8_t *) row_out;
    float   * dd    = (float *) (row_out + K);
    int16_t * bsums = (int16_t *) (row_out + K + (K / PM_QK_K) * 4);
    uint32_t packed = 0;
    int psum = 0;
    if (amax != 0.0f) {
        // lowest index with |x| == amax; low bit carries its sign
        int key = 0x7fffffff;
        if (a3 == amax) key = ((4 * lane + 3) << 1) | (v[3] < 0.0f);
        if (a2 == amax) key = ((4 * lane + 2) << 1) | (v[2] < 0.0f);
        if (a1 == amax) key = ((4 * lane + 1) << 1) | (v[1] < 0.0f);
        if (a0 == amax) key = ((4 * lane + 0) << 1) | (v[0] < 0.0f);
        key = wave_min_i(key);
        const float vmax = (key & 1) ? -amax : amax;
        const float iscale = -127.f / vmax;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int q = nearest_int_rne(iscale * v[i]);
            q = q > 127 ? 127 : q;
            psum += q;
            packed |= (uint32_t) (q & 0xFF) << (8 * i);
        }
        if (lane == 0) dd[blk] = 1 / iscale;
        if (tb.base && lane == 0) ((float *) (tb.base + (size_t) (row >> 5) * tb.tab_bytes + (size_t) tb.nsb * 1024))[blk * 32 + (row & 31)] = 1 / iscale;
    } else if (lane == 0) {
        dd[blk] = 0.0f;
        if (tb.base) ((float *) (tb.base + (size_t) (row >> 5) * tb.tab_bytes + (size_t) tb.nsb * 1024))[blk * 32 + (row & 31)] = 0.0f;
    }
    ((uint32_t *) (qs + (size_t) blk * PM_QK_K))[lane] = packed;
    // A-operand order (pm_q8k_tables::qbase): values 4 lane .. 4 lane + 3 = bytes 4 (lane % 4) of sub-block lane / 8, half (lane / 4) % 2
    if (tb.qbase) *(uint32_t *) (tb.qbase + (size_t) (row >> 5) * tb.nsb * 8192 + (size_t) blk * 8192 + (lane >> 3) * 1024 + ((((lane >> 2) & 1) * 32 + (row & 31)) * 16) + 4 * (lane & 3)) = packed;
    // bsums: 16 values = 4 lanes (one quad)
    psum += dpp_i<0xB1>(psum);
    psum += dpp_i<0x4E>(psum);
    if ((lane & 3) == 0) bsums[blk * 16 + (lane >> 2)] = (int16_t) psum;
    if (tb.base && (lane & 3) == 0) {              // group G = lane / 4 -> operand lane (row % 32) + 32 (G & 1), slot G / 2 (mmq_i8.hip mmq_prep_kernel)
        const int G = lane >> 2;
        ((_Float16 *) (tb.base + (size_t) (row >> 5) * tb.tab_bytes))[((size_t) blk * 64 + (row & 31) + 32 * (G & 1)) * 8 + (G >> 1)] = (_Float16) (float) psum;
    }
}

__global__ __launch_bounds__(256) void quantize_q8k_kernel(const float * __restrict__ x, uint8_t * __restrict__ y,
                                                           int K, int rows, size_t y_row_bytes, pm_q8k_tables tb) {
    const int lane = threadIdx.x & 63;
    const int nblk = K / PM_QK_K;
    const long wave = (long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (long) rows * nblk) return;
    const int row = (int) (wave / nblk), blk = (int) (wave % nblk);
    const float4 f = ((const float4 *) (x + (size_t) row * K + (size_t) blk * PM_QK_K))[lane];
    const float v[4] = {f.x, f.y, f.z, f.w};
    q8k_block_from_regs(v, lane, y + (size_t) row * y_row_bytes, K, blk, tb, row);
}

// silu(gate) * up (ggml_silu + ggml_mul of llm_build_ffn, src/llama.cpp:9804; same f32 expressions as layer_ops.hip's silu_mul_kernel) fused with
// the Q8_K quantization of the product: the ffn_down activations of a small batch, without the f32 round trip through HBM
__global__ __launch_bounds__(256) void silu_mul_q8k_kernel(const float * __restrict__ gate, const float * __restrict__ up, uint8_t * __restrict__ y,
                                                           int K, int rows, size_t y_row_bytes, pm_q8k_tables tb) {
    const int lane = threadIdx.x & 63;
    const int nblk = K / PM_QK_K;
    const long wave = (long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (long) rows * nblk) return;
    const int row = (int) (wave / nblk), blk = (int) (wave % nblk);
    const float4 g = ((const float4 *) (gate + (size_t) row * K + (size_t) blk * PM_QK_K))[lane];
    const float4 u = ((const float4 *) (up + (size_t) row * K + (size_t) blk * PM_QK_K))[lane];
    const float v[4] = {g.x / (1.0f + expf(-g.x)) * u.x, g.y / (1.0f + expf(-g.y)) * u.y, g.z / (1.0f + expf(-g.z)) * u.z, g.w / (1.0f + expf(-g.w)) * u.w};
    q8k_block_from_regs(v, lane, y + (size_t) row * y_row_bytes, K, blk, tb, row);
}

// Q8_0: 32-value blocks, 8 lanes x 4 values; amax over the 8-lane group.
__global__ __launch_bounds__(256) void quantize_q80_kernel(const float * __restrict__ x, uint8_t * __restrict__ y,
                                                           int K, int rows, size_t y_row_bytes) {
    const long t = (long) blockIdx.x * 256 + threadIdx.x;        // one thread = 4 values
    const long per_row = K / 4;
    if (t >= (long) rows * per_row) return;
    const int row = (int) (t / per_row), i4 = (int) (t % per_row);
    const float4 f = ((const float4 *) (x + (size_t) row * K))[i4];
    float amax = fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w)));
    amax = group8_max(amax);
    const float d  = amax / 127;
    const float id = d ? 1.0f / d : 0.0f;
    uint8_t * ro = y + (size_t) row * y_row_bytes;
    const float v[4] = {f.x, f.y, f.z, f.w};
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) packed |= (uint32_t) ((int) roundf(v[i] * id) & 0xFF) << (8 * i);
    ((uint32_t *) ro)[i4] = packed;
    if ((i4 & 7) == 0) ((uint16_t *) (ro + K))[i4 >> 3] = f2h(d);
}

// silu(gate) * up fused with the Q8_0 quantization of the product, written STRAIGHT into the small-batch mat-mul's activation tables (mmq_i8.hip, Q8_0 weights:
// per 32-token pass [blk][64 lanes][16 B] values in A-operand order + [blk][32] f32 scales) - the ffn_down activations of the files whose n_ff is no multiple of
// 256 (Qwen2.5-72B: K = 29568 falls back to Q8_0, src/llama.cpp:19447). Replaces three launches (silu_mul, quantize_q80, the mat-mul's table prologue).
// Rows T .. 32 * passes - 1 write scale 0 (their values are never loaded).
__global__ __launch_bounds__(256) void silu_mul_q80_tab_kernel(const float * __restrict__ gate, const float * __restrict__ up, uint8_t * __restrict__ tab,
                                                               int K, int rows, int rows_padded, size_t pass_bytes, size_t qtab_bytes) {
    const long t = (long) blockIdx.x * 256 + threadIdx.x;        // one thread = 4 values, 8 lanes = one block
    const long per_row = K / 4;
    if (t >= (long) rows_padded * per_row) return;
    const int row = (int) (t / per_row), i4 = (int) (t % per_row);
    const int blk = i4 >> 3, sub = i4 & 7;
    uint8_t * base = tab + (size_t) (row >> 5) * pass_bytes;
    float * dslot = (float *) (base + qtab_bytes) + ((size_t) blk * 32 + (row & 31));
    if (row >= rows) { if (sub == 0) *dslot = 0.0f; return; }
    const float4 g = ((const float4 *) (gate + (size_t) row * K))[i4];
    const float4 u = ((const float4 *) (up + (size_t) row * K))[i4];
    const float v[4] = {g.x / (1.0f + expf(-g.x)) * u.x, g.y / (1.0f + expf(-g.y)) * u.y, g.z / (1.0f + expf(-g.z)) * u.z, g.w / (1.0f + expf(-g.w)) * u.w};
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = group8_max(amax);
    const float d  = amax / 127;
    const float id = d ? 1.0f / d : 0.0f;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) packed |= (uint32_t) ((int) roundf(v[i] * id) & 0xFF) << (8 * i);
    // lane (token slot row % 32, half g = sub / 4) of block blk holds bytes [16 g, 16 g + 16) of the block
    *(uint32_t *) (base + (size_t) blk * 1024 + (size_t) (32 * (sub >> 2) + (row & 31)) * 16 + 4 * (sub & 3)) = packed;
    if (sub == 0) *dslot = h2f(f2h(d));
}

// rms_norm (+ weight) fused with Q8_K quantization. One 256-thread workgroup per row.
// Optionally also writes the normalised f32 row (ynorm != nullptr).
// NWV waves per row: 4 for many rows (prompts: one small workgroup per row fills the chip), 16 for a handful of rows (decode batches of 2..64
// tokens are ONE latency chain per row - load, reduce, quantize block after block: with 4 waves a K = 8192 row took 11 us, a fifth of a 2-token layer)
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void rmsnorm_q8k_kernel(const float * __restrict__ x, const float * __restrict__ w,
                                                          float * __restrict__ ynorm, uint8_t * __restrict__ yq,
                                                          int K, float eps, size_t yq_row_bytes, _Float16 * __restrict__ yh, pm_q8k_tables tb) {
    __shared__ double red[NWV];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = K / PM_QK_K;
    const float * xr = x + (size_t) row * K;
    // A wave's blocks (wv, wv + 4, ...) are loaded ONCE, all loads in flight together, and stay in registers between the two passes
    // (K <= 8192: 8 blocks per wave). With few rows (decode batches) the kernel is one latency chain per row: the former form - one
    // dependent load per block and pass - took 12 us for K = 8192.
    constexpr int HOLD = NWV == 4 ? 8 : 2;
    const bool held = nblk <= NWV * HOLD;
    float4 f[HOLD];
    if (held) {
#pragma unroll
        for (int i = 0; i < HOLD; ++i) { const int blk = min(wv + NWV * i, nblk - 1); f[i] = ((const float4 *) (xr + (size_t) blk * PM_QK_K))[lane]; }
    }
    // pass 1: sum of squares; products rounded to f32 like the reference, accumulated in f64 (same order as before: blocks ascending)
    double s = 0.0;
    if (held) {
#pragma unroll
        for (int i = 0; i < HOLD; ++i) if (wv + NWV * i < nblk) {
            s += (double) (f[i].x * f[i].x); s += (double) (f[i].y * f[i].y); s += (double) (f[i].z * f[i].z); s += (double) (f[i].w * f[i].w);
        }
    } else {
        for (int blk = wv; blk < nblk; blk += NWV) {
            const float4 g = ((const float4 *) (xr + (size_t) blk * PM_QK_K))[lane];
            s += (double) (g.x * g.x); s += (double) (g.y * g.y); s += (double) (g.z * g.z); s += (double) (g.w * g.w);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    double tot = (red[0] + red[1]) + (red[2] + red[3]);
#pragma unroll
    for (int i = 4; i < NWV; i += 4) tot += (red[i] + red[i + 1]) + (red[i + 2] + red[i + 3]);
    const float mean  = (float) (tot / K);
    const float scale = 1.0f / sqrtf(mean + eps);
    // pass 2: normalise (+weight), write f32 and/or quantize
    auto emit = [&](const float4 & fx, int blk) __attribute__((always_inline)) {
        float v[4] = {fx.x * scale, fx.y * scale, fx.z * scale, fx.w * scale};
        if (w) {
            const float4 g = ((const float4 *) (w + (size_t) blk * PM_QK_K))[lane];
            v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w;
        }
        if (ynorm) ((float4 *) (ynorm + (size_t) row * K + (size_t) blk * PM_QK_K))[lane] = make_float4(v[0], v[1], v[2], v[3]);
        if (yh) {                                // F16 copy for the MFMA GEMMs (RNE, the rounding their conversion pass applies)
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            ((h4 *) (yh + (size_t) row * K + (size_t) blk * PM_QK_K))[lane] = h4{(_Float16) v[0], (_Float16) v[1], (_Float16) v[2], (_Float16) v[3]};
        }
        if (yq) q8k_block_from_regs(v, lane, yq + (size_t) row * yq_row_bytes, K, blk, tb, row);
    };
    if (held) {
#pragma unroll
        for (int i = 0; i < HOLD; ++i) if (wv + NWV * i < nblk) emit(f[i], wv + NWV * i);
    } else {
        for (int blk = wv; blk < nblk; blk += NWV) emit(((const float4 *) (xr + (size_t) blk * PM_QK_K))[lane], blk);
    }
}

// ---- host launchers -------------------------------------------------------------------------------
void pm_launch_quantize_q8k(const float * x, void * y, int K, int rows, hipStream_t st, pm_q8k_tables tab) {
    const long waves = (long) rows * (K / PM_QK_K);
    hipLaunchKernelGGL(quantize_q8k_kernel, dim3((unsigned) ((waves + 3) / 4)), dim3(256), 0, st,
                       x, (uint8_t *) y, K, rows, pm_q8k_row_bytes(K), tab);
}
void pm_launch_silu_mul_q8k(const float * gate, const float * up, void * y, int K, int rows, hipStream_t st, pm_q8k_tables tab) {
    const long waves = (long) rows * (K / PM_QK_K);
    hipLaunchKernelGGL(silu_mul_q8k_kernel, dim3((unsigned) ((waves + 3) / 4)), dim3(256), 0, st, gate, up, (uint8_t *) y, K, rows, pm_q8k_row_bytes(K), tab);
}
void pm_launch_silu_mul_q80_tab(const float * gate, const float * up, void * tab, size_t qtab_bytes, size_t dtab_bytes, int K, int rows, hipStream_t st) {
    const int rows_padded = (rows + 31) / 32 * 32;
    const long n = (long) rows_padded * (K / 4);
    hipLaunchKernelGGL(silu_mul_q80_tab_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, gate, up, (uint8_t *) tab, K, rows, rows_padded,
                       qtab_bytes + dtab_bytes, qtab_bytes);
}
void pm_launch_quantize_q80(const float * x, void * y, int K, int rows, hipStream_t st) {
    const long n = (long) rows * (K / 4);
    hipLaunchKernelGGL(quantize_q80_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st,
                       x, (uint8_t *) y, K, rows, pm_q80_row_bytes(K));
}
void pm_launch_rmsnorm_q8k(const float * x, const float * w, float * ynorm, void * yq, int K, int rows, float eps, hipStream_t st, void * ynorm_f16, pm_q8k_tables tab) {
    if (rows <= 64) hipLaunchKernelGGL(rmsnorm_q8k_kernel<16>, dim3(rows), dim3(1024), 0, st, x, w, ynorm, (uint8_t *) yq, K, eps, pm_q8k_row_bytes(K), (_Float16 *) ynorm_f16, tab);
    else            hipLaunchKernelGGL(rmsnorm_q8k_kernel<4>, dim3(rows), dim3(256), 0, st, x, w, ynorm, (uint8_t *) yq, K, eps, pm_q8k_row_bytes(K), (_Float16 *) ynorm_f16, tab);
}
