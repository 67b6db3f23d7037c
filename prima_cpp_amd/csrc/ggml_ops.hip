// ggml_ops.hip — node-equivalent, stride-aware versions of the ggml ops on the hot path, used by the ggml-backend
// plug-in when a graph node does not match one of the fused fast paths (arbitrary views / permutes / broadcasts as
// ggml_backend_sched hands them over). Semantics follow the reference CPU ops (paths relative to the reference repo):
//   cpy / cont      ggml_compute_forward_dup_f32 / _f16            ggml/src/ggml.c:8509, :8238 (F32<->F16 RNE conversion)
//   add / mul       ggml_compute_forward_add_f32 / mul_f32          ggml.c:9002, :10077   (src1 broadcast over src0)
//   scale, silu     ggml_compute_forward_scale_f32 :11262, silu_f32 :11581
//   rms_norm        ggml_compute_forward_rms_norm_f32               ggml.c:11950
//   soft_max_ext    ggml_compute_forward_soft_max_f32               ggml.c:13783  (mask F32/F16, scale, ALiBi slopes)
//   rope            ggml_compute_forward_rope_f32                   ggml.c:14143  (NORM / NEOX, freq factors, YaRN)
//   mul_mat F16xF32 ggml_compute_forward_mul_mat with vec_dot_type F16: src1 rounded to F16, f32 accumulate, broadcast r2/r3
//   get_rows        ggml_compute_forward_get_rows_f32 / _q          ggml.c:13414, :13288
// None of these is bandwidth-critical at decode (KBs per call); they exist so that every node the scheduler assigns to
// the backend has an implementation with the reference's rounding points.
#include "../../include/prima_mi355.h"
#include "pm355_device.h"
#include "pm355_kernels.h"
#include "pm355_layer_ops.h"

struct TD { char * data; long ne[4]; long nb[4]; };       // byte strides

static TD to_td(const pm355_tensor * t) {
    TD d; d.data = (char *) t->data;
    for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = (long) t->nb[i]; }
    return d;
}
static long nelem(const pm355_tensor * t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }

__device__ __forceinline__ float ld_as_f32(const char * p, int type) { return type == PM_F16 ? h2f(*(const uint16_t *) p) : *(const float *) p; }
__device__ __forceinline__ void st_from_f32(char * p, int type, float v) { if (type == PM_F16) *(uint16_t *) p = f2h(v); else *(float *) p = v; }

// dst element order = dst's own (i0,i1,i2,i3); the source is addressed by the same LOGICAL flat index (ggml_cpy allows a
// reshape between src and dst as long as the element counts match)
__global__ void cpy_kernel(TD s, int st, TD d, int dt, long n) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long r = i;
    const long d0 = r % d.ne[0]; r /= d.ne[0]; const long d1 = r % d.ne[1]; r /= d.ne[1]; const long d2 = r % d.ne[2]; const long d3 = r / d.ne[2];
    r = i;
    const long s0 = r % s.ne[0]; r /= s.ne[0]; const long s1 = r % s.ne[1]; r /= s.ne[1]; const long s2 = r % s.ne[2]; const long s3 = r / s.ne[2];
    const char * sp = s.data + s0 * s.nb[0] + s1 * s.nb[1] + s2 * s.nb[2] + s3 * s.nb[3];
    char * dp = d.data + d0 * d.nb[0] + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3];
    if (st == dt) { if (st == PM_F16) *(uint16_t *) dp = *(const uint16_t *) sp; else *(uint32_t *) dp = *(const uint32_t *) sp; }
    else st_from_f32(dp, dt, ld_as_f32(sp, st));
}

// op: 0 add, 1 mul ; b broadcast (ne_b[i] divides ne_a[i])
__global__ void bin_kernel(TD a, TD b, TD d, int op, long n) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long r = i;
    const long i0 = r % d.ne[0]; r /= d.ne[0]; const long i1 = r % d.ne[1]; r /= d.ne[1]; const long i2 = r % d.ne[2]; const long i3 = r / d.ne[2];
    const float x = *(const float *) (a.data + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    const float y = *(const float *) (b.data + (i0 % b.ne[0]) * b.nb[0] + (i1 % b.ne[1]) * b.nb[1] + (i2 % b.ne[2]) * b.nb[2] + (i3 % b.ne[3]) * b.nb[3]);
    *(float *) (d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = op == 0 ? x + y : x * y;
}

// op: 0 scale by s, 1 silu
__global__ void unary_kernel(TD a, TD d, int op, float sc, long n) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long r = i;
    const long i0 = r % d.ne[0]; r /= d.ne[0]; const long i1 = r % d.ne[1]; r /= d.ne[1]; const long i2 = r % d.ne[2]; const long i3 = r / d.ne[2];
    const float x = *(const float *) (a.data + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    *(float *) (d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = op == 0 ? x * sc : x / (1.0f + expf(-x));
}

// one 256-thread workgroup per row; rows may be strided (nb[1..3]), elements contiguous
__global__ __launch_bounds__(256) void rms_norm_rows_kernel(TD a, TD d, float eps) {
    __shared__ double red[4];
    const long row = blockIdx.x;
    long r = row; const long i1 = r % a.ne[1]; r /= a.ne[1]; const long i2 = r % a.ne[2]; const long i3 = r / a.ne[2];
    const float * x = (const float *) (a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    float * y = (float *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (long i = tid; i < a.ne[0]; i += 256) s += (double) (x[i] * x[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const double tot = (red[0] + red[1]) + (red[2] + red[3]);
    const float mean = (float) (tot / a.ne[0]);
    const float scale = 1.0f / sqrtf(mean + eps);
    for (long i = tid; i < a.ne[0]; i += 256) y[i] = x[i] * scale;
}

// soft_max_ext: x [ne0, ne1, ne2, ne3] contiguous rows; mask [ne0, >=ne1] F32 or F16 or null; one workgroup per row
__global__ __launch_bounds__(256) void soft_max_kernel(TD a, const char * mask, int mask_type, long mask_nb1, TD d,
                                                       float scale, float max_bias, float m0, float m1, unsigned n_head_log2) {
    __shared__ float redf[4];
    __shared__ double redd[4];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float * wp = (float *) smem;
    const long row = blockIdx.x;
    long r = row; const long i1 = r % a.ne[1]; r /= a.ne[1]; const long i2 = r % a.ne[2]; const long i3 = r / a.ne[2];
    const float * x = (const float *) (a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    float * y = (float *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned h = (unsigned) i2;
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    const long nc = a.ne[0];
    float mx = -INFINITY;
    for (long i = tid; i < nc; i += 256) {
        float v = x[i] * scale;
        if (mask) v += slope * ld_as_f32(mask + i1 * mask_nb1 + i * (mask_type == PM_F16 ? 2 : 4), mask_type);
        wp[i] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double s = 0.0;
    for (long i = tid; i < nc; i += 256) { const float e = expf(wp[i] - mx); wp[i] = e; s += (double) e; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) redd[wave] = s;
    __syncthreads();
    const float inv = (float) (1.0 / ((redd[0] + redd[1]) + (redd[2] + redd[3])));
    for (long i = tid; i < nc; i += 256) y[i] = wp[i] * inv;
}

// rope on x [d, heads, T, ne3] (strided) -> y (strided). pos int32 [T]. One 64-thread workgroup per (head, token, i3).
struct RopeG { int n_dims, mode; float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1; };
// F16: x converted to f32, rotated, rounded back (ggml_compute_forward_rope_f16, ggml.c:14269-14391: the in-place K-shift of an F16 cache)
template <bool F16>
__global__ __launch_bounds__(64) void rope_generic_kernel(TD a, const int32_t * pos, const float * ff, TD d, RopeG r) {
    auto ld = [](const char * p_) { return F16 ? h2f(*(const uint16_t *) p_) : *(const float *) p_; };
    auto st = [](char * p_, float v) { if (F16) *(uint16_t *) p_ = f2h(v); else *(float *) p_ = v; };
    const long h = blockIdx.x, t = blockIdx.y, i3 = blockIdx.z;
    const char * src = a.data + h * a.nb[1] + t * a.nb[2] + i3 * a.nb[3];
    char * dst = d.data + h * d.nb[1] + t * d.nb[2] + i3 * d.nb[3];
    const bool neox = r.mode & 2;
    const int half = r.n_dims / 2, dh = (int) a.ne[0];
    for (int pair = threadIdx.x; pair < dh / 2; pair += 64) {
        int ia, ib; float o0, o1;
        if (pair < half) {
            ia = neox ? pair : 2 * pair; ib = neox ? pair + half : 2 * pair + 1;
            float theta = (float) pos[t];
            for (int j = 0; j < pair; ++j) theta *= r.theta_scale;
            const float te = theta / (ff ? ff[pair] : 1.0f);
            float ti = r.freq_scale * te, th = ti, ms = r.attn_factor;
            if (r.ext_factor != 0.0f) {
                const float y = ((float) pair - r.corr0) / fmaxf(0.001f, r.corr1 - r.corr0);
                const float mix = (1 - fminf(1, fmaxf(0, y))) * r.ext_factor;
                th = ti * (1 - mix) + te * mix;
                ms *= 1.0f + 0.1f * logf(1.0f / r.freq_scale);
            }
            const float c = cosf(th) * ms, s = sinf(th) * ms;
            const float x0 = ld(src + ia * a.nb[0]), x1 = ld(src + ib * a.nb[0]);
            o0 = x0 * c - x1 * s; o1 = x0 * s + x1 * c;
        } else {
            ia = r.n_dims + 2 * (pair - half); ib = ia + 1;
            o0 = ld(src + ia * a.nb[0]); o1 = ld(src + ib * a.nb[0]);
        }
        st(dst + ia * d.nb[0], o0); st(dst + ib * d.nb[0], o1);
    }
}

// dst[i0, i1, i2, i3] = sum_k src0[k, i0, i2/r2, i3/r3] * f16(src1[k, i1, i2, i3]); src0 F16 or F32; one wave per element
__global__ __launch_bounds__(256) void mul_mat_f_kernel(TD a, int at, TD b, TD d, long r2, long r3, long n_out) {
    const long o = (long) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= n_out) return;
    long r = o;
    const long i0 = r % d.ne[0]; r /= d.ne[0]; const long i1 = r % d.ne[1]; r /= d.ne[1]; const long i2 = r % d.ne[2]; const long i3 = r / d.ne[2];
    const char * ap = a.data + i0 * a.nb[1] + (i2 / r2) * a.nb[2] + (i3 / r3) * a.nb[3];
    const char * bp = b.data + i1 * b.nb[1] + i2 * b.nb[2] + i3 * b.nb[3];
    float acc = 0.0f;
    for (long k = lane; k < a.ne[0]; k += 64) {
        const float w = ld_as_f32(ap + k * a.nb[0], at);
        float x = *(const float *) (bp + k * b.nb[0]);
        if (at == PM_F16) x = h2f(f2h(x));                    // vec_dot_type F16: the activation is rounded to F16
        acc += w * x;
    }
    acc = wave_sum(acc);
    if (lane == 0) *(float *) (d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = acc;
}

// MUL_MAT with a Q8_0 src0 in ggml's NATIVE 34-byte blocks and arbitrary row strides: the K.q product of llm_build_kqv on a `-ctk q8_0` cache
// WITHOUT flash attention (src/llama.cpp:10062-10068: k = view of the quantized cache [dh, n_kv, Hkv]). vec_dot_type of Q8_0 is Q8_0
// (ggml.c:865-881): the f32 row of src1 is quantized per 32 values (quantize_row_q8_0_ref, ggml-quants.c:848: d = max|x| / 127 stored as F16,
// q = roundf(x / d)), the dot is ggml_vec_dot_q8_0_q8_0 (:5518): sum over blocks of (int32 dot) * d_k * d_q. One wave per output element,
// one lane per block of the row (head_dim / 32 lanes busy: a node-equivalent op for a few KB, not a bandwidth kernel).
__global__ __launch_bounds__(256) void mul_mat_q80_kernel(TD a, TD b, TD d, long r2, long r3, long n_out) {
    const long o = (long) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= n_out) return;
    long r = o;
    const long i0 = r % d.ne[0]; r /= d.ne[0]; const long i1 = r % d.ne[1]; r /= d.ne[1]; const long i2 = r % d.ne[2]; const long i3 = r / d.ne[2];
    const char * ap = a.data + i0 * a.nb[1] + (i2 / r2) * a.nb[2] + (i3 / r3) * a.nb[3];
    const char * bp = b.data + i1 * b.nb[1] + i2 * b.nb[2] + i3 * b.nb[3];
    float acc = 0.0f;
    for (long blk = lane; blk < a.ne[0] / 32; blk += 64) {
        float x[32], amax = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { x[j] = *(const float *) (bp + (blk * 32 + j) * b.nb[0]); amax = fmaxf(amax, fabsf(x[j])); }
        const float dq = amax / 127, id = dq ? 1.0f / dq : 0.0f;
        const uint8_t * kb = (const uint8_t *) (ap + blk * 34);
        const float dk = h2f((uint16_t) (kb[0] | (kb[1] << 8)));
        int sumi = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) sumi += (int) (int8_t) kb[2 + j] * (int) roundf(x[j] * id);
        acc += (float) sumi * (dk * h2f(f2h(dq)));
    }
    acc = wave_sum(acc);
    if (lane == 0) *(float *) (d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = acc;
}

// CPY / CAST Q8_0 (native blocks, any row strides) -> F32: the K-shift graph of a quantized cache dequantizes the rows it rotates
// (build_k_shift, src/llama.cpp:10665-10690: ggml_cast(k, F32) -> rope -> cpy back); dequantize_row_q8_0 (ggml-quants.c:1616)
__global__ void cpy_q80_f32_kernel(TD s, TD d, long n) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long r = i;
    const long d0 = r % d.ne[0]; r /= d.ne[0]; const long d1 = r % d.ne[1]; r /= d.ne[1]; const long d2 = r % d.ne[2]; const long d3 = r / d.ne[2];
    r = i;
    const long s0 = r % s.ne[0]; r /= s.ne[0]; const long s1 = r % s.ne[1]; r /= s.ne[1]; const long s2 = r % s.ne[2]; const long s3 = r / s.ne[2];
    const uint8_t * blk = (const uint8_t *) (s.data + (s0 / 32) * 34 + s1 * s.nb[1] + s2 * s.nb[2] + s3 * s.nb[3]);
    const float v = h2f((uint16_t) (blk[0] | (blk[1] << 8))) * (float) (int8_t) blk[2 + (s0 & 31)];
    *(float *) (d.data + d0 * d.nb[0] + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3]) = v;
}

// CPY Q8_0 -> Q8_0 between views of native blocks (any row strides): the cell moves of build_defrag on a quantized cache
// (src/llama.cpp:10721-10790; the CPU backend copies bytes: ggml_compute_forward_dup_bytes, ggml.c:8527). One thread per 34-byte block.
__global__ void cpy_q80_q80_kernel(TD s, TD d, long n_blocks) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) return;
    const long nb0 = d.ne[0] / 32;
    long r = i;
    const long b0 = r % nb0; r /= nb0; const long i1 = r % d.ne[1]; r /= d.ne[1]; const long i2 = r % d.ne[2]; const long i3 = r / d.ne[2];
    const uint16_t * sp = (const uint16_t *) (s.data + b0 * 34 + i1 * s.nb[1] + i2 * s.nb[2] + i3 * s.nb[3]);
    uint16_t * dp = (uint16_t *) (d.data + b0 * 34 + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
#pragma unroll
    for (int k = 0; k < 17; ++k) dp[k] = sp[k];
}

__global__ void get_rows_f32_kernel(TD a, const int32_t * idx, long n_idx, TD d) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.ne[0] * n_idx) return;
    const long c = i % a.ne[0], r = i / a.ne[0];
    *(float *) (d.data + c * d.nb[0] + r * d.nb[1]) = *(const float *) (a.data + c * a.nb[0] + (long) idx[r] * a.nb[1]);
}

#define S(st) ((hipStream_t) (st))
#define GRID(n) dim3((unsigned) (((n) + 255) / 256)), dim3(256)
#define OKRET() do { return hipGetLastError() == hipSuccess ? PM355_OK : PM355_E_HIP; } while (0)


// ---- GGML_OP_FLASH_ATTN_EXT (ggml_compute_forward_flash_attn_ext_f16, ggml.c:15538-15760; llm_build_kqv flash path
//      src/llama.cpp:10075-10095): q F32 [D, N, H, B], k / v F16 [D, n_kv, Hkv, B] (V NOT transposed), mask F16 [n_kv, >= N] or null,
//      dst F32 [D, H, N, B]. One 256-thread workgroup per (query, head, batch): scores over all keys into LDS (q rounded to F16
//      like the reference's q_to_vec_dot, f32 accumulate), softcap / ALiBi slope / mask, max, exp, f32 sum, then P.V with the keys
//      dealt to 256 / (D/8) slots of 8 channels each. The reference runs an online softmax that re-scales an F16 accumulator;
//      here the accumulation is f32 (differences far below the reference's own backend tolerance, NMSE 5e-4).
__global__ __launch_bounds__(256) void flash_attn_ext_kernel(TD q, TD k, TD v, const char * mask, long mask_nb1, TD d, float scale,
                                                           float max_bias, float softcap, float m0, float m1, unsigned n_head_log2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float redf[8];
    const int D = (int) q.ne[0];
    const int n_kv = (int) k.ne[1];
    float * qs = (float *) smem;                              // [Dp]
    const int Dp = (D + 7) & ~7;
    float * sc = qs + Dp;                                     // [n_kv]
    float * part = sc + ((n_kv + 3) & ~3);                    // [nslot][Dp]
    const int iq1 = blockIdx.x, h = blockIdx.y, b3 = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = h / (int) (q.ne[2] / k.ne[2]), hv = h / (int) (q.ne[2] / v.ne[2]);
    const int bk = b3 / (int) (q.ne[3] / k.ne[3]), bv = b3 / (int) (q.ne[3] / v.ne[3]);
    const char * qp = q.data + (long) iq1 * q.nb[1] + (long) h * q.nb[2] + (long) b3 * q.nb[3];
    for (int e = tid; e < Dp; e += 256) qs[e] = e < D ? h2f(f2h(*(const float *) (qp + (long) e * 4))) : 0.0f;
    const float slope = max_bias > 0.0f ? ((unsigned) h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - (int) n_head_log2) + 1))) : 1.0f;
    const uint16_t * mp = mask ? (const uint16_t *) (mask + (long) iq1 * mask_nb1) : nullptr;
    __syncthreads();
    float lmax = -INFINITY;
    for (int i = tid; i < n_kv; i += 256) {
        const float mv = mp ? slope * h2f(mp[i]) : 0.0f;
        float s = -INFINITY;
        if (mv != -INFINITY) {
            const uint16_t * kr = (const uint16_t *) (k.data + (long) i * k.nb[1] + (long) hk * k.nb[2] + (long) bk * k.nb[3]);
            float acc = 0.0f;
            for (int e = 0; e < D; ++e) acc += h2f(kr[e]) * qs[e];
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
    // P.V: thread = (8-channel chunk c, key slot)
    const int nchunk = Dp / 8, nslot = 256 / nchunk;
    const int c = tid % nchunk, slot = tid / nchunk;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (slot < nslot) {
        for (int i = slot; i < n_kv; i += nslot) {
            const float p = sc[i];
            if (p == 0.0f) continue;
            const uint16_t * vr = (const uint16_t *) (v.data + (long) i * v.nb[1] + (long) hv * v.nb[2] + (long) bv * v.nb[3]) + 8 * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (8 * c + j < D) acc[j] += h2f(vr[j]) * p;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) part[slot * Dp + 8 * c + j] = acc[j];
    }
    __syncthreads();
    char * dp = d.data + (long) h * d.nb[1] + (long) iq1 * d.nb[2] + (long) b3 * d.nb[3];
    for (int e = tid; e < D; e += 256) {
        float o = 0.0f;
        for (int sl = 0; sl < nslot; ++sl) o += part[sl * Dp + e];
        *(float *) (dp + (long) e * 4) = S > 0.0f ? o / S : 0.0f;
    }
}

extern "C" {

int pm355_op_cpy(const pm355_tensor * src, const pm355_tensor * dst, pm355_stream_t st) {
    const int ts = src->type, td = dst->type;
    if (ts == PM_Q8_0 && td == PM_Q8_0) {            // cell moves of the defragmentation graph on a quantized cache: same shape, block-wise bytes
        for (int i = 0; i < 4; ++i) if (src->ne[i] != dst->ne[i]) return PM355_E_SHAPE;
        if (src->ne[0] % 32 || src->nb[0] != 34 || dst->nb[0] != 34) return PM355_E_SHAPE;
        const long nblk = nelem(dst) / 32;
        (void) hipGetLastError();
        hipLaunchKernelGGL(cpy_q80_q80_kernel, GRID(nblk), 0, S(st), to_td(src), to_td(dst), nblk);
        OKRET();
    }
    if (td == PM_Q8_0) {                             // KV store into a quantized cache (attn_q8.hip); dst = contiguous native blocks
        if (nelem(src) != nelem(dst)) return PM355_E_SHAPE;
        (void) hipGetLastError();
        if (pm_launch_cpy_f32_q8_0(src, dst->data, S(st))) return PM355_E_UNSUPPORTED;
        OKRET();
    }
    if (ts == PM_Q8_0 && td == PM_F32) {             // dequantizing copy (K-shift of a quantized cache); src = native 34-byte blocks
        if (nelem(src) != nelem(dst) || src->ne[0] % 32) return PM355_E_SHAPE;
        const long nq = nelem(dst);
        (void) hipGetLastError();
        hipLaunchKernelGGL(cpy_q80_f32_kernel, GRID(nq), 0, S(st), to_td(src), to_td(dst), nq);
        OKRET();
    }
    if ((ts != PM_F32 && ts != PM_F16) || (td != PM_F32 && td != PM_F16) || nelem(src) != nelem(dst)) return PM355_E_UNSUPPORTED;
    const long n = nelem(dst);
    (void) hipGetLastError();
    hipLaunchKernelGGL(cpy_kernel, GRID(n), 0, S(st), to_td(src), ts, to_td(dst), td, n);
    OKRET();
}
int pm355_op_binary(int op, const pm355_tensor * a, const pm355_tensor * b, const pm355_tensor * dst, pm355_stream_t st) {
    if (a->type != PM_F32 || b->type != PM_F32 || dst->type != PM_F32 || (op != 0 && op != 1)) return PM355_E_UNSUPPORTED;
    for (int i = 0; i < 4; ++i) if (b->ne[i] <= 0 || a->ne[i] % b->ne[i]) return PM355_E_SHAPE;
    const long n = nelem(dst);
    (void) hipGetLastError();
    hipLaunchKernelGGL(bin_kernel, GRID(n), 0, S(st), to_td(a), to_td(b), to_td(dst), op, n);
    OKRET();
}
int pm355_op_unary(int op, const pm355_tensor * a, const pm355_tensor * dst, float param, pm355_stream_t st) {
    if (a->type != PM_F32 || dst->type != PM_F32 || (op != 0 && op != 1)) return PM355_E_UNSUPPORTED;
    const long n = nelem(dst);
    (void) hipGetLastError();
    hipLaunchKernelGGL(unary_kernel, GRID(n), 0, S(st), to_td(a), to_td(dst), op, param, n);
    OKRET();
}
int pm355_op_rms_norm(const pm355_tensor * a, const pm355_tensor * dst, float eps, pm355_stream_t st) {
    if (a->type != PM_F32 || dst->type != PM_F32 || a->nb[0] != 4 || dst->nb[0] != 4) return PM355_E_UNSUPPORTED;
    const long rows = a->ne[1] * a->ne[2] * a->ne[3];
    (void) hipGetLastError();
    hipLaunchKernelGGL(rms_norm_rows_kernel, dim3((unsigned) rows), dim3(256), 0, S(st), to_td(a), to_td(dst), eps);
    OKRET();
}
int pm355_op_soft_max(const pm355_tensor * a, const pm355_tensor * mask, const pm355_tensor * dst, float scale, float max_bias,
                      pm355_stream_t st) {
    if (a->type != PM_F32 || dst->type != PM_F32 || a->nb[0] != 4 || dst->nb[0] != 4) return PM355_E_UNSUPPORTED;
    if (mask && ((mask->type != PM_F32 && mask->type != PM_F16) || mask->ne[0] != a->ne[0] || mask->ne[1] < a->ne[1])) return PM355_E_UNSUPPORTED;
    const size_t lds = (size_t) a->ne[0] * 4;
    if (lds > 150 * 1024) return PM355_E_RANGE;
    const long rows = a->ne[1] * a->ne[2] * a->ne[3];
    const unsigned n_head = (unsigned) a->ne[2];
    const unsigned n_head_log2 = 1u << (unsigned) floor(log2((double) n_head));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    static bool set[16] = {};
    const int dv = lds > 48 * 1024 ? pm_cur_dev() : 0;
    if (lds > 48 * 1024 && !set[dv]) { (void) hipFuncSetAttribute((const void *) soft_max_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); set[dv] = true; }
    (void) hipGetLastError();
    hipLaunchKernelGGL(soft_max_kernel, dim3((unsigned) rows), dim3(256), lds, S(st), to_td(a), mask ? (const char *) mask->data : nullptr,
                       mask ? mask->type : 0, mask ? (long) mask->nb[1] : 0, to_td(dst), scale, max_bias, m0, m1, n_head_log2);
    OKRET();
}
int pm355_op_flash_attn_ext(const pm355_tensor * q, const pm355_tensor * k, const pm355_tensor * v, const pm355_tensor * mask,
                            const pm355_tensor * dst, float scale, float max_bias, float logit_softcap, pm355_stream_t st) {
    if (q->type == PM_F32 && dst->type == PM_F32 && (k->type == PM_Q8_0 || v->type == PM_Q8_0)) {      // quantized KV cache (attn_q8.hip)
        if ((k->type != PM_Q8_0 && k->type != PM_F16) || (v->type != PM_Q8_0 && v->type != PM_F16) || q->nb[0] != 4 || dst->nb[0] != 4 || max_bias != 0.0f)
            return PM355_E_UNSUPPORTED;
        if (k->ne[0] != q->ne[0] || v->ne[0] != q->ne[0] || v->ne[1] != k->ne[1]) return PM355_E_SHAPE;
        if (q->ne[2] % k->ne[2] || q->ne[2] % v->ne[2] || q->ne[3] % k->ne[3] || q->ne[3] % v->ne[3]) return PM355_E_SHAPE;
        if (mask && (mask->type != PM_F16 || mask->ne[0] != k->ne[1] || mask->ne[1] < q->ne[1] || mask->nb[0] != 2)) return PM355_E_UNSUPPORTED;
        if (dst->ne[0] != q->ne[0] || dst->ne[1] != q->ne[2] || dst->ne[2] != q->ne[1]) return PM355_E_SHAPE;
        (void) hipGetLastError();
        const int rc = pm_launch_flash_attn_ext_q8(q, k, v, mask, dst, scale, logit_softcap, S(st));
        if (rc) return rc == -2 ? PM355_E_RANGE : PM355_E_UNSUPPORTED;
        OKRET();
    }
    if (q->type != PM_F32 || k->type != PM_F16 || v->type != PM_F16 || dst->type != PM_F32) return PM355_E_UNSUPPORTED;
    if (q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2 || dst->nb[0] != 4) return PM355_E_UNSUPPORTED;
    if (k->ne[0] != q->ne[0] || v->ne[0] != q->ne[0] || v->ne[1] != k->ne[1] || q->ne[0] > 256 || q->ne[0] < 8) return PM355_E_SHAPE;
    if (q->ne[2] % k->ne[2] || q->ne[2] % v->ne[2] || q->ne[3] % k->ne[3] || q->ne[3] % v->ne[3]) return PM355_E_SHAPE;
    if (mask && (mask->type != PM_F16 || mask->ne[0] != k->ne[1] || mask->ne[1] < q->ne[1] || mask->nb[0] != 2)) return PM355_E_UNSUPPORTED;
    if (dst->ne[0] != q->ne[0] || dst->ne[1] != q->ne[2] || dst->ne[2] != q->ne[1]) return PM355_E_SHAPE;
    const int Dp = ((int) q->ne[0] + 7) & ~7, nslot = 256 / (Dp / 8);
    const size_t lds = ((size_t) Dp + (((size_t) k->ne[1] + 3) & ~(size_t) 3) + (size_t) nslot * Dp) * 4;
    if (lds > 150 * 1024) return PM355_E_RANGE;
    if (logit_softcap != 0.0f) scale /= logit_softcap;
    const unsigned n_head = (unsigned) q->ne[2];
    const unsigned n_head_log2 = 1u << (unsigned) floor(log2((double) n_head));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    static bool set[16] = {};
    const int dv = lds > 48 * 1024 ? pm_cur_dev() : 0;
    if (lds > 48 * 1024 && !set[dv]) { (void) hipFuncSetAttribute((const void *) flash_attn_ext_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); set[dv] = true; }
    (void) hipGetLastError();
    hipLaunchKernelGGL(flash_attn_ext_kernel, dim3((unsigned) q->ne[1], (unsigned) q->ne[2], (unsigned) q->ne[3]), dim3(256), lds, S(st),
                       to_td(q), to_td(k), to_td(v), mask ? (const char *) mask->data : nullptr, mask ? (long) mask->nb[1] : 0, to_td(dst),
                       scale, max_bias, logit_softcap, m0, m1, n_head_log2);
    OKRET();
}
int pm355_op_rope(const pm355_tensor * a, const int32_t * d_pos, const float * freq_factors, const pm355_tensor * dst,
                  const pm355_rope_params * rp, pm355_stream_t st) {
    if ((a->type != PM_F32 && a->type != PM_F16) || dst->type != a->type || !rp || rp->n_dims % 2 || rp->n_dims > a->ne[0] || a->ne[0] % 2) return PM355_E_UNSUPPORTED;
    if (rp->mode & ~2) return PM355_E_UNSUPPORTED;                   // only NORM (0) and NEOX (2)
    pm_rope_cfg c;
    c.n_dims = rp->n_dims; c.mode = rp->mode; c.n_ctx_orig = rp->n_ctx_orig; c.freq_base = rp->freq_base; c.freq_scale = rp->freq_scale;
    c.ext_factor = rp->ext_factor; c.attn_factor = rp->attn_factor; c.beta_fast = rp->beta_fast; c.beta_slow = rp->beta_slow;
    pm_rope_params(c);
    RopeG r = {c.n_dims, c.mode, c.theta_scale, c.freq_scale, c.ext_factor, c.attn_factor, c.corr0, c.corr1};
    (void) hipGetLastError();
    if (a->type == PM_F16)
        hipLaunchKernelGGL(rope_generic_kernel<true>, dim3((unsigned) a->ne[1], (unsigned) a->ne[2], (unsigned) a->ne[3]), dim3(64), 0, S(st),
                           to_td(a), d_pos, freq_factors, to_td(dst), r);
    else
        hipLaunchKernelGGL(rope_generic_kernel<false>, dim3((unsigned) a->ne[1], (unsigned) a->ne[2], (unsigned) a->ne[3]), dim3(64), 0, S(st),
                           to_td(a), d_pos, freq_factors, to_td(dst), r);
    OKRET();
}
int pm355_op_mul_mat_f(const pm355_tensor * a, const pm355_tensor * b, const pm355_tensor * dst, pm355_stream_t st) {
    if ((a->type != PM_F16 && a->type != PM_F32 && a->type != PM_Q8_0) || b->type != PM_F32 || dst->type != PM_F32) return PM355_E_UNSUPPORTED;
    if (a->ne[0] != b->ne[0] || b->ne[2] % a->ne[2] || b->ne[3] % a->ne[3]) return PM355_E_SHAPE;
    const long n = nelem(dst);
    (void) hipGetLastError();
    if (a->type == PM_Q8_0) {                        // native Q8_0 blocks (a view of a quantized K cache): Q8_0 x Q8_0 integer dots
        if (a->ne[0] % 32) return PM355_E_SHAPE;
        hipLaunchKernelGGL(mul_mat_q80_kernel, dim3((unsigned) ((n + 3) / 4)), dim3(256), 0, S(st), to_td(a), to_td(b), to_td(dst),
                           b->ne[2] / a->ne[2], b->ne[3] / a->ne[3], n);
        OKRET();
    }
    hipLaunchKernelGGL(mul_mat_f_kernel, dim3((unsigned) ((n + 3) / 4)), dim3(256), 0, S(st), to_td(a), a->type, to_td(b), to_td(dst),
                       b->ne[2] / a->ne[2], b->ne[3] / a->ne[3], n);
    OKRET();
}
int pm355_op_get_rows_f32(const pm355_tensor * a, const int32_t * d_idx, int64_t n_idx, const pm355_tensor * dst, pm355_stream_t st) {
    if (a->type != PM_F32 || dst->type != PM_F32) return PM355_E_UNSUPPORTED;
    const long n = a->ne[0] * n_idx;
    (void) hipGetLastError();
    hipLaunchKernelGGL(get_rows_f32_kernel, GRID(n), 0, S(st), to_td(a), d_idx, (long) n_idx, to_td(dst));
    OKRET();
}

} // extern "C"
