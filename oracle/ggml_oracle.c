// oracle/ggml_oracle.c — CPU restatement (plain scalar C) of the reference's quantized decode path.
//
// *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so,
// and only as the checker. prima_cpp_amd/ never links, loads or falls back to anything here.
//
// Parity status: PINNED. tests/test_oracle_vs_ref.py checks every function below bit-for-bit
// against oracle/_ref/libggml_ref_scalar.so = the unmodified reference sources compiled with
// -march=x86-64 (the canonical scalar `#else` branches of ggml-quants.c), and within float
// summation-order tolerance against the AVX2 build; tests/golden/*.npz hold vectors produced
// by that reference build (generator: tests/golden/make_golden.py) so the pin also holds on
// machines where /root/reference and oracle/_ref are absent.
//
// Every function cites the reference code it restates (paths relative to /root/reference).
// The arithmetic (rounding points, accumulation widths and order) follows the reference's
// scalar code path exactly; the code itself is written from the format definitions.
#include "ggml_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define QK_K 256

// ------------------------------------------------------------------------------------------
// fp16 <-> fp32, IEEE-754 binary16 round-to-nearest-even
// (reference: GGML_FP32_TO_FP16 / GGML_FP16_TO_FP32, ggml/src/ggml-impl.h; table ggml.c:374)
// ------------------------------------------------------------------------------------------
float orc_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t) (h & 0x8000u) << 16;
    uint32_t exp  = (h >> 10) & 0x1fu;
    uint32_t man  = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {                      // subnormal: normalise
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            bits = sign | ((uint32_t) (127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f; memcpy(&f, &bits, 4); return f;
}

uint16_t orc_f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint16_t sign = (uint16_t) ((x >> 16) & 0x8000u);
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t) (sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (ax >= 0x477ff000u) return (uint16_t) (sign | 0x7c00u);          // rounds to >= 65520 -> inf
    if (ax < 0x33000001u) return sign;                                   // < 2^-25 (or == 2^-25 ties to even 0)
    int32_t e = (int32_t) (ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    uint32_t shift, half_bits;
    if (e < -14) {                   // subnormal half: value = m * 2^(e-23); unit = 2^-24
        shift = (uint32_t) (13 + (-14 - e));
        half_bits = 0;
    } else {
        shift = 13;
        half_bits = (uint32_t) (e + 15) << 10;
        m &= 0x7fffffu;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
    return (uint16_t) (sign | (half_bits + q));                          // carry into exponent is correct
}

// ------------------------------------------------------------------------------------------
// block formats (reference: ggml/src/ggml-common.h:187-191, :286-335). Parsed by offset, so
// no struct packing assumptions.
// ------------------------------------------------------------------------------------------
enum { BS_Q8_0 = 34, BS_Q4_K = 144, BS_Q5_K = 176, BS_Q6_K = 210, BS_Q8_K = 292 };

int64_t orc_row_size(int type, int64_t k) {
    switch (type) {
        case ORC_F32:  return 4 * k;
        case ORC_F16:  return 2 * k;
        case ORC_Q8_0: return k / 32 * BS_Q8_0;
        case ORC_Q4_K: return k / QK_K * BS_Q4_K;
        case ORC_Q5_K: return k / QK_K * BS_Q5_K;
        case ORC_Q6_K: return k / QK_K * BS_Q6_K;
        case ORC_Q8_K: return k / QK_K * BS_Q8_K;
    }
    return -1;
}

// type_traits[].vec_dot_type (reference: ggml/src/ggml.c:773-784, :865-881, :916-951)
int orc_vec_dot_type(int type) {
    switch (type) {
        case ORC_F32:  return ORC_F32;
        case ORC_F16:  return ORC_F16;
        case ORC_Q8_0: return ORC_Q8_0;
        default:       return ORC_Q8_K;
    }
}

static inline uint16_t rd_u16(const uint8_t * p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline float    rd_f32(const uint8_t * p) { float v; memcpy(&v, p, 4); return v; }
static inline int16_t  rd_i16(const uint8_t * p) { int16_t v; memcpy(&v, p, 2); return v; }

// round-to-nearest-even via the 1.5*2^23 trick (reference: nearest_int, ggml-quants.c:1638-1644)
static inline int round_magic(float v) {
    float t = v + 12582912.f;
    int32_t i; memcpy(&i, &t, 4);
    return (i & 0x007fffff) - 0x00400000;
}

// ------------------------------------------------------------------------------------------
// activation quantizers
// ------------------------------------------------------------------------------------------

// reference: quantize_row_q8_K_ref, ggml-quants.c:3785-3826
// block: float d | int8 qs[256] | int16 bsums[16]
void orc_quantize_row_q8_K(const float * x, void * vy, int64_t k) {
    uint8_t * y = vy;
    for (int64_t b = 0; b < k / QK_K; ++b, x += QK_K, y += BS_Q8_K) {
        float vmax = 0.f, amax = 0.f;
        for (int j = 0; j < QK_K; ++j) {
            float a = fabsf(x[j]);
            if (a > amax) { amax = a; vmax = x[j]; }         // first element with the largest |x|
        }
        int8_t * qs = (int8_t *) (y + 4);
        if (amax == 0.f) {
            float z = 0.f; memcpy(y, &z, 4);
            memset(qs, 0, QK_K);
            // NB: the reference leaves bsums untouched in this case (they are never read with d == 0
            // contributing: d multiplies everything). We zero them for determinism.
            memset(y + 4 + QK_K, 0, 32);
            continue;
        }
        const float iscale = -127.f / vmax;
        for (int j = 0; j < QK_K; ++j) {
            int v = round_magic(iscale * x[j]);
            qs[j] = (int8_t) (v > 127 ? 127 : v);
        }
        for (int g = 0; g < 16; ++g) {
            int s = 0;
            for (int i = 0; i < 16; ++i) s += qs[16 * g + i];
            int16_t s16 = (int16_t) s; memcpy(y + 4 + QK_K + 2 * g, &s16, 2);
        }
        float d = 1 / iscale; memcpy(y, &d, 4);
    }
}

// reference: quantize_row_q8_0_ref, ggml-quants.c:848-871.  block: half d | int8 qs[32]
void orc_quantize_row_q8_0(const float * x, void * vy, int64_t k) {
    uint8_t * y = vy;
    for (int64_t b = 0; b < k / 32; ++b, x += 32, y += BS_Q8_0) {
        float amax = 0.f;
        for (int j = 0; j < 32; ++j) { float a = fabsf(x[j]); if (a > amax) amax = a; }
        const float d  = amax / 127;
        const float id = d ? 1.0f / d : 0.0f;
        uint16_t dh = orc_f32_to_f16(d); memcpy(y, &dh, 2);
        for (int j = 0; j < 32; ++j) ((int8_t *) (y + 2))[j] = (int8_t) roundf(x[j] * id);
    }
}

// ------------------------------------------------------------------------------------------
// K-quant helpers
// ------------------------------------------------------------------------------------------

// 12 bytes -> 8 six-bit scales + 8 six-bit mins (reference: get_scale_min_k4, ggml-quants.c:1898-1906)
static void unpack_k4_scales(const uint8_t * s12, uint8_t sc[8], uint8_t mn[8]) {
    for (int j = 0; j < 4; ++j) {
        sc[j]     = s12[j] & 63;
        mn[j]     = s12[j + 4] & 63;
        sc[j + 4] = (uint8_t) ((s12[j + 8] & 0x0F) | ((s12[j] >> 6) << 4));
        mn[j + 4] = (uint8_t) ((s12[j + 8] >> 4)   | ((s12[j + 4] >> 6) << 4));
    }
}

// expand one super-block to 256 small integers, in element order
static void expand_q4_K(const uint8_t * blk, int8_t q[QK_K]) {     // layout: ggml-common.h:286-297
    const uint8_t * qs = blk + 16;
    for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 32; ++l) {
            q[64 * g + l]      = (int8_t) (qs[32 * g + l] & 0x0F);
            q[64 * g + 32 + l] = (int8_t) (qs[32 * g + l] >> 4);
        }
}
static void expand_q5_K(const uint8_t * blk, int8_t q[QK_K]) {     // layout: ggml-common.h:303-315
    const uint8_t * qh = blk + 16, * qs = blk + 48;
    for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 32; ++l) {
            q[64 * g + l]      = (int8_t) ((qs[32 * g + l] & 0x0F) + (((qh[l] >> (2 * g)) & 1) << 4));
            q[64 * g + 32 + l] = (int8_t) ((qs[32 * g + l] >> 4)   + (((qh[l] >> (2 * g + 1)) & 1) << 4));
        }
}
static void expand_q6_K(const uint8_t * blk, int8_t q[QK_K]) {     // layout: ggml-common.h:321-327
    const uint8_t * ql = blk, * qh = blk + 128;
    for (int h = 0; h < 2; ++h)
        for (int l = 0; l < 32; ++l) {
            uint8_t hb = qh[32 * h + l];
            q[128 * h + l]      = (int8_t) (((ql[64 * h + l] & 0x0F)      | (((hb >> 0) & 3) << 4)) - 32);
            q[128 * h + 32 + l] = (int8_t) (((ql[64 * h + 32 + l] & 0x0F) | (((hb >> 2) & 3) << 4)) - 32);
            q[128 * h + 64 + l] = (int8_t) (((ql[64 * h + l] >> 4)        | (((hb >> 4) & 3) << 4)) - 32);
            q[128 * h + 96 + l] = (int8_t) (((ql[64 * h + 32 + l] >> 4)   | (((hb >> 6) & 3) << 4)) - 32);
        }
}

// ------------------------------------------------------------------------------------------
// dequantize (reference: dequantize_row_q8_0 :1616, _q4_K :2555-2579, _q5_K :2763-2789, _q6_K :2977-3007)
// ------------------------------------------------------------------------------------------
void orc_dequantize_row(int type, const void * vx, float * y, int64_t k) {
    const uint8_t * x = vx;
    int8_t q[QK_K]; uint8_t sc[8], mn[8];
    switch (type) {
    case ORC_F32: memcpy(y, x, (size_t) k * 4); return;
    case ORC_F16: for (int64_t i = 0; i < k; ++i) y[i] = orc_f16_to_f32(rd_u16(x + 2 * i)); return;
    case ORC_Q8_0:
        for (int64_t b = 0; b < k / 32; ++b, x += BS_Q8_0, y += 32) {
            const float d = orc_f16_to_f32(rd_u16(x));
            for (int j = 0; j < 32; ++j) y[j] = ((const int8_t *) (x + 2))[j] * d;
        }
        return;
    case ORC_Q4_K:
    case ORC_Q5_K:
        for (int64_t b = 0; b < k / QK_K; ++b, x += (type == ORC_Q4_K ? BS_Q4_K : BS_Q5_K), y += QK_K) {
            const float d = orc_f16_to_f32(rd_u16(x)), dmin = orc_f16_to_f32(rd_u16(x + 2));
            unpack_k4_scales(x + 4, sc, mn);
            if (type == ORC_Q4_K) expand_q4_K(x, q); else expand_q5_K(x, q);
            for (int s = 0; s < 8; ++s) {
                const float ds = d * sc[s], ms = dmin * mn[s];
                for (int l = 0; l < 32; ++l) y[32 * s + l] = ds * q[32 * s + l] - ms;
            }
        }
        return;
    case ORC_Q6_K:
        for (int64_t b = 0; b < k / QK_K; ++b, x += BS_Q6_K, y += QK_K) {
            const float d = orc_f16_to_f32(rd_u16(x + 208));
            const int8_t * scl = (const int8_t *) (x + 192);
            expand_q6_K(x, q);
            for (int s = 0; s < 16; ++s)
                for (int l = 0; l < 16; ++l) y[16 * s + l] = d * scl[s] * q[16 * s + l];
        }
        return;
    }
}

// ------------------------------------------------------------------------------------------
// dot products
// ------------------------------------------------------------------------------------------

// Integer core shared by the float dot and the partials API. For one super-block:
//   lane[l] (l = 0..7) = sum over sub-blocks j of scale_j * sum_{i = l mod 8} q_w[i] * q_a[i]
// which is the reference scalar path's aux32[l] (ggml-quants.c:8255-8270 / :8893-8908 / :9552-9562).
static void kq_block_lanes(int type, const uint8_t * w, const uint8_t * a, int32_t lane[8], int32_t * msum) {
    int8_t q[QK_K]; uint8_t sc[8], mn[8];
    const int8_t * q8 = (const int8_t *) (a + 4);
    memset(lane, 0, 8 * sizeof(int32_t));
    *msum = 0;
    if (type == ORC_Q6_K) {
        const int8_t * scl = (const int8_t *) (w + 192);
        expand_q6_K(w, q);
        for (int s = 0; s < 16; ++s)
            for (int i = 0; i < 16; ++i) lane[i & 7] += scl[s] * (int32_t) (int16_t) (q8[16 * s + i] * q[16 * s + i]);
        return;
    }
    unpack_k4_scales(w + 4, sc, mn);
    if (type == ORC_Q4_K) expand_q4_K(w, q); else expand_q5_K(w, q);
    for (int g = 0; g < 16; ++g) *msum += rd_i16(a + 4 + QK_K + 2 * g) * mn[g / 2];
    for (int s = 0; s < 8; ++s)
        for (int i = 0; i < 32; ++i) lane[i & 7] += sc[s] * (int32_t) (int16_t) (q8[32 * s + i] * q[32 * s + i]);
}

void orc_vec_dot_int_partials(int type, int64_t n, const void * vw, const void * va, int32_t * isum, int32_t * msum) {
    const uint8_t * w = vw, * a = va;
    if (type == ORC_Q8_0) {
        for (int64_t b = 0; b < n / 32; ++b, w += BS_Q8_0, a += BS_Q8_0) {
            int32_t s = 0;
            for (int j = 0; j < 32; ++j) s += ((const int8_t *) (w + 2))[j] * ((const int8_t *) (a + 2))[j];
            isum[b] = s; if (msum) msum[b] = 0;
        }
        return;
    }
    const int bs = type == ORC_Q4_K ? BS_Q4_K : type == ORC_Q5_K ? BS_Q5_K : BS_Q6_K;
    for (int64_t b = 0; b < n / QK_K; ++b, w += bs, a += BS_Q8_K) {
        int32_t lane[8], ms;
        kq_block_lanes(type, w, a, lane, &ms);
        int32_t s = 0; for (int l = 0; l < 8; ++l) s += lane[l];
        isum[b] = s; if (msum) msum[b] = ms;
    }
}

// reference: ggml_vec_dot_q4_K_q8_K scalar branch ggml-quants.c:8222-8278, _q5_K :8854-8914,
// _q6_K :9523-9566, ggml_vec_dot_q8_0_q8_0 scalar tail :5752-5765 (pattern), ggml_vec_dot_f16 ggml.c:2202,
// ggml_vec_dot_f32 ggml.c:2050.
float orc_vec_dot(int type, int64_t n, const void * vw, const void * va) {
    const uint8_t * w = vw, * a = va;
    switch (type) {
    case ORC_F32: {
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += (double) (rd_f32(w + 4 * i) * rd_f32(a + 4 * i));
        return (float) s;
    }
    case ORC_F16: {
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += (double) (orc_f16_to_f32(rd_u16(w + 2 * i)) * orc_f16_to_f32(rd_u16(a + 2 * i)));
        return (float) s;
    }
    case ORC_Q8_0: {
        float sumf = 0.f;
        for (int64_t b = 0; b < n / 32; ++b, w += BS_Q8_0, a += BS_Q8_0) {
            int32_t s = 0;
            for (int j = 0; j < 32; ++j) s += ((const int8_t *) (w + 2))[j] * ((const int8_t *) (a + 2))[j];
            sumf += s * (orc_f16_to_f32(rd_u16(w)) * orc_f16_to_f32(rd_u16(a)));
        }
        return sumf;
    }
    default: break;
    }
    const int bs = type == ORC_Q4_K ? BS_Q4_K : type == ORC_Q5_K ? BS_Q5_K : BS_Q6_K;
    float sums[8] = {0}, sumf = 0.f;
    for (int64_t b = 0; b < n / QK_K; ++b, w += bs, a += BS_Q8_K) {
        int32_t lane[8], ms;
        kq_block_lanes(type, w, a, lane, &ms);
        const float yd = rd_f32(a);
        const float d = orc_f16_to_f32(rd_u16(type == ORC_Q6_K ? w + 208 : w)) * yd;
        for (int l = 0; l < 8; ++l) sums[l] += d * lane[l];
        if (type != ORC_Q6_K) {
            const float dmin = orc_f16_to_f32(rd_u16(w + 2)) * yd;
            sumf -= dmin * ms;
        }
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

// ------------------------------------------------------------------------------------------
// mul_mat (reference: ggml_compute_forward_mul_mat, ggml.c:12377-12590: src1 rows are converted to
// vec_dot_type first (:12445-12473), then every dst element is ONE vec_dot call)
// ------------------------------------------------------------------------------------------
static void to_vec_dot_type(int vdt, const float * x, void * y, int64_t k) {
    switch (vdt) {
    case ORC_F32:  memcpy(y, x, (size_t) k * 4); break;
    case ORC_F16:  for (int64_t i = 0; i < k; ++i) { uint16_t h = orc_f32_to_f16(x[i]); memcpy((uint8_t *) y + 2 * i, &h, 2); } break;
    case ORC_Q8_0: orc_quantize_row_q8_0(x, y, k); break;
    default:       orc_quantize_row_q8_K(x, y, k); break;
    }
}

void orc_mul_mat(int type, const void * W, int64_t K, int64_t N, const float * x, int64_t ncols, float * out) {
    const int vdt = orc_vec_dot_type(type);
    const int64_t wrow = orc_row_size(type, K), arow = orc_row_size(vdt, K);
    uint8_t * aq = malloc((size_t) arow);
    for (int64_t c = 0; c < ncols; ++c) {
        to_vec_dot_type(vdt, x + c * K, aq, K);
        for (int64_t r = 0; r < N; ++r) out[c * N + r] = orc_vec_dot(type, K, (const uint8_t *) W + r * wrow, aq);
    }
    free(aq);
}

// ------------------------------------------------------------------------------------------
// layer ops
// ------------------------------------------------------------------------------------------

// reference: ggml_compute_forward_rms_norm_f32 ggml.c:11950-11996 (+ ggml_compute_forward_mul_f32 :10077)
void orc_rms_norm(const float * x, const float * w, int64_t n, int64_t rows, float eps, float * out) {
    for (int64_t r = 0; r < rows; ++r, x += n, out += n) {
        double sum = 0.0;
        for (int64_t i = 0; i < n; ++i) sum += (double) (x[i] * x[i]);
        const float mean = (float) (sum / n);
        const float scale = 1.0f / sqrtf(mean + eps);
        for (int64_t i = 0; i < n; ++i) { float v = x[i] * scale; out[i] = w ? v * w[i] : v; }
    }
}

// reference: rope_yarn_ramp ggml.c:14086, rope_yarn :14094-14109, ggml_rope_yarn_corr_dim(s) :14113-14141,
// ggml_rope_cache_init :14117-14131, ggml_compute_forward_rope_f32 :14143-14266
static float yarn_ramp(float low, float high, int i0) {
    const float y = (i0 / 2 - low) / fmaxf(0.001f, high - low);
    return 1 - fminf(1, fmaxf(0, y));
}
static float yarn_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}
void orc_rope(const float * x, int64_t d, int64_t heads, int64_t ntok, const int32_t * pos,
              const float * ff, int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale,
              float ext_factor, float attn_factor, float beta_fast, float beta_slow, float * out) {
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    float corr[2];
    corr[0] = fmaxf(0, floorf(yarn_corr_dim(n_dims, n_ctx_orig, beta_fast, freq_base)));
    corr[1] = fminf((float) (n_dims - 1), ceilf(yarn_corr_dim(n_dims, n_ctx_orig, beta_slow, freq_base)));
    const int neox = mode & 2;
    float * cs = malloc(sizeof(float) * (size_t) d);
    for (int64_t t = 0; t < ntok; ++t) {
        float theta = (float) pos[t];
        for (int i0 = 0; i0 < n_dims; i0 += 2) {
            const float f = ff ? ff[i0 / 2] : 1.0f;
            const float te = theta / f;
            float ti = freq_scale * te, th = ti, ms = attn_factor;
            if (ext_factor != 0.0f) {
                float mix = yarn_ramp(corr[0], corr[1], i0) * ext_factor;
                th = ti * (1 - mix) + te * mix;
                ms *= 1.0f + 0.1f * logf(1.0f / freq_scale);
            }
            cs[i0] = cosf(th) * ms;
            cs[i0 + 1] = sinf(th) * ms;
            theta *= theta_scale;
        }
        for (int64_t h = 0; h < heads; ++h) {
            const float * s = x + (t * heads + h) * d;
            float * o = out + (t * heads + h) * d;
            for (int i0 = 0; i0 < n_dims; i0 += 2) {
                const int a = neox ? i0 / 2 : i0, b = neox ? i0 / 2 + n_dims / 2 : i0 + 1;
                const float x0 = s[a], x1 = s[b];
                o[a] = x0 * cs[i0] - x1 * cs[i0 + 1];
                o[b] = x0 * cs[i0 + 1] + x1 * cs[i0];
            }
            for (int64_t i = n_dims; i < d; ++i) o[i] = s[i];
        }
    }
    free(cs);
}

// reference: ggml_compute_forward_soft_max_f32 ggml.c:13783-13879, ggml_vec_soft_max_f32 :2786 (scalar tail)
void orc_soft_max_ext(const float * x, const float * mask, int64_t nc, int64_t nr, int64_t heads,
                      float scale, float max_bias, float * out) {
    const uint32_t n_head_log2 = 1u << (uint32_t) floor(log2((double) heads));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    float * wp = malloc(sizeof(float) * (size_t) nc);
    for (int64_t h = 0; h < heads; ++h)
        for (int64_t r = 0; r < nr; ++r) {
            const float slope = (max_bias > 0.0f) ? ((uint32_t) h < n_head_log2 ? powf(m0, (float) (h + 1))
                                                   : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
            const float * sp = x + (h * nr + r) * nc;
            float * dp = out + (h * nr + r) * nc;
            for (int64_t i = 0; i < nc; ++i) wp[i] = sp[i] * scale;
            if (mask) for (int64_t i = 0; i < nc; ++i) wp[i] += slope * mask[r * nc + i];
            float mx = -INFINITY;
            for (int64_t i = 0; i < nc; ++i) mx = fmaxf(mx, wp[i]);
            double sum = 0.0;
            for (int64_t i = 0; i < nc; ++i) { float v = expf(wp[i] - mx); dp[i] = v; sum += (double) v; }
            const float inv = (float) (1.0 / sum);
            for (int64_t i = 0; i < nc; ++i) dp[i] *= inv;
        }
    free(wp);
}

// reference: ggml_silu_f32 ggml.c:2560-2562, ggml_compute_forward_mul_f32 :10077
void orc_silu_mul(const float * g, const float * u, int64_t n, float * out) {
    for (int64_t i = 0; i < n; ++i) {
        float s = g[i] / (1.0f + expf(-g[i]));
        out[i] = u ? s * u[i] : s;
    }
}

// ------------------------------------------------------------------------------------------
// whole decoder stack — same graph as oracle/ref_ops.c:ref_model_eval, computed with the
// functions above (reference builders: build_llama src/llama.cpp:11000-11215, build_qwen2 :12736).
// ------------------------------------------------------------------------------------------
typedef struct {
    orc_model_desc d;
    uint16_t ** k_l, ** v_l;      // K: [n_ctx][Hkv*dh] f16 ; V transposed: [Hkv*dh][n_ctx] f16 (llm_build_kv_store :9707)
} orc_model;

void * orc_model_new(const orc_model_desc * d) {
    orc_model * m = calloc(1, sizeof(*m));
    m->d = *d;
    const size_t n = (size_t) d->head_dim * d->n_head_kv * d->n_ctx;
    m->k_l = calloc(d->n_layer, sizeof(void *));
    m->v_l = calloc(d->n_layer, sizeof(void *));
    for (int l = 0; l < d->n_layer; ++l) { m->k_l[l] = calloc(n, 2); m->v_l[l] = calloc(n, 2); }
    return m;
}
void orc_model_free(void * vm) {
    orc_model * m = vm;
    if (!m) return;
    for (int l = 0; l < m->d.n_layer; ++l) { free(m->k_l[l]); free(m->v_l[l]); }
    free(m->k_l); free(m->v_l); free(m);
}
void orc_model_kv_clear(void * vm) {
    orc_model * m = vm;
    const size_t n = (size_t) m->d.head_dim * m->d.n_head_kv * m->d.n_ctx;
    for (int l = 0; l < m->d.n_layer; ++l) { memset(m->k_l[l], 0, n * 2); memset(m->v_l[l], 0, n * 2); }
}
const void * orc_model_kv_ptr(void * vm, int il, int which) {
    orc_model * m = vm;
    return which ? m->v_l[il] : m->k_l[il];
}

static void add_bias(float * y, const orc_tensor_t * b, int64_t n, int64_t cols) {
    if (!b || !b->data) return;
    const float * bv = b->data;
    for (int64_t c = 0; c < cols; ++c) for (int64_t i = 0; i < n; ++i) y[c * n + i] += bv[i];
}

int orc_model_eval(void * vm, const int32_t * tokens, const float * embd_in, int T, int pos0,
                   int layer_lo, int layer_hi, int with_head, float * hidden_out, float * logits_out, int n_threads) {
    (void) n_threads;
    orc_model * m = vm;
    const orc_model_desc * d = &m->d;
    const int64_t E = d->n_embd, dh = d->head_dim, H = d->n_head, Hkv = d->n_head_kv, F = d->n_ff;
    const int64_t Eq = dh * H, Ekv = dh * Hkv, n_ctx = d->n_ctx;
    int64_t n_kv = ((pos0 + T + 31) / 32) * 32; if (n_kv > n_ctx) n_kv = n_ctx;
    const int rope_mode = d->arch == 1 ? 2 : 0;
    const float kq_scale = 1.0f / sqrtf((float) dh);
    const int gqa = (int) (H / Hkv);

    float * x    = malloc(sizeof(float) * (size_t) (E * T));
    float * cur  = malloc(sizeof(float) * (size_t) (E * T));
    float * q    = malloc(sizeof(float) * (size_t) (Eq * T));
    float * kk   = malloc(sizeof(float) * (size_t) (Ekv * T));
    float * vv   = malloc(sizeof(float) * (size_t) (Ekv * T));
    float * qr   = malloc(sizeof(float) * (size_t) (Eq * T));
    float * kr   = malloc(sizeof(float) * (size_t) (Ekv * T));
    float * att  = malloc(sizeof(float) * (size_t) (Eq * T));
    float * g    = malloc(sizeof(float) * (size_t) (F * T));
    float * u    = malloc(sizeof(float) * (size_t) (F * T));
    float * kq   = malloc(sizeof(float) * (size_t) (n_kv * T * H));
    float * kqs  = malloc(sizeof(float) * (size_t) (n_kv * T * H));
    float * mask = malloc(sizeof(float) * (size_t) (n_kv * T));
    int32_t * pos = malloc(sizeof(int32_t) * (size_t) T);
    uint16_t * qh = malloc(2 * (size_t) dh), * ph = malloc(2 * (size_t) n_kv);

    if (embd_in) memcpy(x, embd_in, sizeof(float) * (size_t) (E * T));
    else for (int t = 0; t < T; ++t)      // ggml_compute_forward_get_rows_q ggml.c:13288
        orc_dequantize_row(d->tok_embd.type, (const uint8_t *) d->tok_embd.data + tokens[t] * orc_row_size(d->tok_embd.type, E), x + t * E, E);
    for (int t = 0; t < T; ++t) {
        pos[t] = pos0 + t;
        for (int64_t i = 0; i < n_kv; ++i) mask[t * n_kv + i] = i <= pos0 + t ? 0.0f : -INFINITY;
    }

    for (int il = layer_lo; il < layer_hi; ++il) {
        orc_rms_norm(x, d->attn_norm[il].data, E, T, d->rms_eps, cur);
        orc_mul_mat(d->wq[il].type, d->wq[il].data, E, Eq, cur, T, q);   add_bias(q, d->bq ? &d->bq[il] : NULL, Eq, T);
        orc_mul_mat(d->wk[il].type, d->wk[il].data, E, Ekv, cur, T, kk); add_bias(kk, d->bk ? &d->bk[il] : NULL, Ekv, T);
        orc_mul_mat(d->wv[il].type, d->wv[il].data, E, Ekv, cur, T, vv); add_bias(vv, d->bv ? &d->bv[il] : NULL, Ekv, T);
        orc_rope(q, dh, H, T, pos, d->rope_freqs, (int) dh, rope_mode, d->n_ctx_orig, d->rope_freq_base, d->rope_freq_scale, 0.0f, 1.0f, 32.0f, 1.0f, qr);
        orc_rope(kk, dh, Hkv, T, pos, d->rope_freqs, (int) dh, rope_mode, d->n_ctx_orig, d->rope_freq_base, d->rope_freq_scale, 0.0f, 1.0f, 32.0f, 1.0f, kr);
        // KV store: CPY f32 -> f16 (ggml_compute_forward_dup_f32 ggml.c:8509)
        for (int t = 0; t < T; ++t)
            for (int64_t c = 0; c < Ekv; ++c) {
                m->k_l[il][(size_t) (pos0 + t) * Ekv + c] = orc_f32_to_f16(kr[t * Ekv + c]);
                m->v_l[il][(size_t) c * n_ctx + pos0 + t] = orc_f32_to_f16(vv[t * Ekv + c]);
            }
        // kq[h][t][i] = K[i, hk, :] . f16(q[t, h, :])     (mul_mat with vec_dot_type F16: ggml.c:12445-12473)
        for (int64_t h = 0; h < H; ++h)
            for (int t = 0; t < T; ++t) {
                for (int64_t e = 0; e < dh; ++e) qh[e] = orc_f32_to_f16(qr[(t * H + h) * dh + e]);
                for (int64_t i = 0; i < n_kv; ++i)
                    kq[(h * T + t) * n_kv + i] = orc_vec_dot(ORC_F16, dh, m->k_l[il] + (size_t) i * Ekv + (h / gqa) * dh, qh);
            }
        orc_soft_max_ext(kq, mask, n_kv, T, H, kq_scale, 0.0f, kqs);
        // kqv[h][t][e] = Vt[hk, e, 0:n_kv] . f16(p[h][t][:])
        for (int64_t h = 0; h < H; ++h)
            for (int t = 0; t < T; ++t) {
                for (int64_t i = 0; i < n_kv; ++i) ph[i] = orc_f32_to_f16(kqs[(h * T + t) * n_kv + i]);
                for (int64_t e = 0; e < dh; ++e)
                    att[t * Eq + h * dh + e] = orc_vec_dot(ORC_F16, n_kv, m->v_l[il] + (size_t) ((h / gqa) * dh + e) * n_ctx, ph);
            }
        orc_mul_mat(d->wo[il].type, d->wo[il].data, Eq, E, att, T, cur);
        for (int64_t i = 0; i < E * T; ++i) x[i] = cur[i] + x[i];                         // ffn_inp = cur + inpSA
        orc_rms_norm(x, d->ffn_norm[il].data, E, T, d->rms_eps, cur);
        orc_mul_mat(d->ffn_up[il].type, d->ffn_up[il].data, E, F, cur, T, u);
        orc_mul_mat(d->ffn_gate[il].type, d->ffn_gate[il].data, E, F, cur, T, g);
        orc_silu_mul(g, u, F * T, g);
        orc_mul_mat(d->ffn_down[il].type, d->ffn_down[il].data, F, E, g, T, cur);
        for (int64_t i = 0; i < E * T; ++i) x[i] = cur[i] + x[i];
    }
    if (hidden_out) memcpy(hidden_out, x, sizeof(float) * (size_t) (E * T));
    if (with_head && logits_out) {
        orc_rms_norm(x + (size_t) (T - 1) * E, d->out_norm.data, E, 1, d->rms_eps, cur);
        orc_mul_mat(d->output.type, d->output.data, E, d->n_vocab, cur, 1, logits_out);
    }
    free(x); free(cur); free(q); free(kk); free(vv); free(qr); free(kr); free(att); free(g); free(u);
    free(kq); free(kqs); free(mask); free(pos); free(qh); free(ph);
    return 0;
}
