// mmvq.hip — batch-1 quantized mat-vec for gfx950 (the decode hot loop).
//
// Replaces, for tensors in our buffers, the reference's
//   ggml_compute_forward_mul_mat (ggml/src/ggml.c:12377) -> ggml_vec_dot_q4_K_q8_K (ggml-quants.c:7713),
//   ggml_vec_dot_q5_K_q8_K (:8281), ggml_vec_dot_q6_K_q8_K (:8918), ggml_vec_dot_q8_0_q8_0 (:5518).
// Integer arithmetic is identical to the reference (int8 x int4/5/6/8 products, int32 sums, 6-bit
// scales/mins, Q8_K bsums for the min / -32 terms) and is checked bit-for-bit against the oracle
// through the dbg_int output; the float scale-and-accumulate uses a fixed, deterministic order.
//
// Design ("x-stationary"): HBM-bound, 4.5-8.5 bits per weight, no reuse of W -> no LDS staging of W.
//  * A row of W is cut into UNITS (16-48 contiguous bytes = 32/64 weights). A workgroup of 256 threads
//    assigns each thread the SAME unit positions for every row it processes, so the matching slice of
//    the quantized activation (Q8_K int8 + bsums + d) is loaded ONCE into VGPRs and stays there.
//  * Per row a thread then issues 2-5 independent 16-byte non-temporal loads (global_load_dwordx4 nt),
//    R rows deep, before consuming any of them: >= 8 KB in flight per workgroup, several workgroups per CU.
//  * 64 lanes of a wave cover a contiguous 1-3 KB span of the row -> fully coalesced; all bytes of every
//    fetched line are used. Alignment: Q4_K/Q5_K blocks are 9/11 x 16 B; Q6_K/Q8_0 are stored row-SoA
//    (repack.hip) so every field stream is 16-B aligned.
//  * v_dot4_i32_i8 for the products, DPP row reductions + one LDS word per wave for the row sum.
#include "pm355_device.h"
#include "pm355_kernels.h"

namespace {

struct GemvP {
    const uint8_t * W; const uint8_t * W2; const uint8_t * xq;
    float * y; const float * bias; const float * resid; int32_t * dbg;
    int K, N, U /*units per row*/, tpr /*threads per row: 64|128|256*/, rows_per_wg;
    long row_bytes;
};

template <int TYPE> struct QT;

// ---------------------------------------------------------------- Q4_K (native 144-B blocks) -------
template <> struct QT<PM_Q4_K> {
    static constexpr int VPU = 32;
    struct X { uint32_t a[4], b[4]; int bs; float yd; };
    struct Wr { u32x4 q, h; };
    static __device__ __forceinline__ void load_x(X & x, const uint8_t * xq, int K, int u) {
        const int b = u >> 3, c = u & 7, j = c >> 1, h = c & 1;
        const uint8_t * p = xq + b * 256 + 64 * j + 16 * h;
        const u32x4 lo = *(const u32x4 *) p, hi = *(const u32x4 *) (p + 32);
        for (int i = 0; i < 4; ++i) { x.a[i] = lo[i]; x.b[i] = hi[i]; }
        x.yd = ((const float *) (xq + K))[b];
        const int16_t * bs = (const int16_t *) (xq + K + (K / 256) * 4) + b * 16 + 2 * c;
        x.bs = bs[0] + bs[1];
    }
    static __device__ __forceinline__ void issue(Wr & w, const uint8_t * row, int, int u) {
        const uint8_t * blk = row + (long) (u >> 3) * PM_BS_Q4_K;
        w.q = ld_nt16(blk + 16 + 16 * (u & 7));
        w.h = ld_nt16(blk);
    }
    static __device__ __forceinline__ float consume(const Wr & w, const X & x, int u, int & isum, int & msum) {
        const int c = u & 7, j = c >> 1;
        int slo = 0, shi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            slo = dot4(w.q[i] & 0x0F0F0F0Fu, x.a[i], slo);
            shi = dot4((w.q[i] >> 4) & 0x0F0F0F0Fu, x.b[i], shi);
        }
        int sc0, sc1, m0, mc;
        k4_scale_min(w.h[1], w.h[2], w.h[3], 2 * j, sc0, m0);
        k4_scale_min(w.h[1], w.h[2], w.h[3], 2 * j + 1, sc1, m0);
        k4_scale_min(w.h[1], w.h[2], w.h[3], c, m0, mc);
        isum = sc0 * slo + sc1 * shi;
        msum = mc * x.bs;
        const float d = h2f((uint16_t) (w.h[0] & 0xFFFF)), dmin = h2f((uint16_t) (w.h[0] >> 16));
        return x.yd * (d * (float) isum - dmin * (float) msum);
    }
};

// ---------------------------------------------------------------- Q5_K (native 176-B blocks) -------
template <> struct QT<PM_Q5_K> {
    static constexpr int VPU = 32;
    typedef QT<PM_Q4_K>::X X;
    struct Wr { u32x4 q, h, qh; };
    static __device__ __forceinline__ void load_x(X & x, const uint8_t * xq, int K, int u) { QT<PM_Q4_K>::load_x(x, xq, K, u); }
    static __device__ __forceinline__ void issue(Wr & w, const uint8_t * row, int, int u) {
        const uint8_t * blk = row + (long) (u >> 3) * PM_BS_Q5_K;
        w.q  = ld_nt16(blk + 48 + 16 * (u & 7));
        w.qh = ld_nt16(blk + 16 + 16 * (u & 1));
        w.h  = ld_nt16(blk);
    }
    static __device__ __forceinline__ float consume(const Wr & w, const X & x, int u, int & isum, int & msum) {
        const int c = u & 7, j = c >> 1;
        int slo = 0, shi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = (w.q[i] & 0x0F0F0F0Fu) | (((w.qh[i] >> (2 * j)) & 0x01010101u) << 4);
            const uint32_t hi = ((w.q[i] >> 4) & 0x0F0F0F0Fu) | (((w.qh[i] >> (2 * j + 1)) & 0x01010101u) << 4);
            slo = dot4(lo, x.a[i], slo);
            shi = dot4(hi, x.b[i], shi);
        }
        int sc0, sc1, m0, mc;
        k4_scale_min(w.h[1], w.h[2], w.h[3], 2 * j, sc0, m0);
        k4_scale_min(w.h[1], w.h[2], w.h[3], 2 * j + 1, sc1, m0);
        k4_scale_min(w.h[1], w.h[2], w.h[3], c, m0, mc);
        isum = sc0 * slo + sc1 * shi;
        msum = mc * x.bs;
        const float d = h2f((uint16_t) (w.h[0] & 0xFFFF)), dmin = h2f((uint16_t) (w.h[0] >> 16));
        return x.yd * (d * (float) isum - dmin * (float) msum);
    }
};

// ---------------------------------------------------------------- Q6_K (row-SoA) --------------------
// row: ql[nb][128] | qh[nb][64] | scales[nb][16] | d[nb]      unit = (block b, half hh, 16-col slice v)
template <> struct QT<PM_Q6_K> {
    static constexpr int VPU = 64;
    struct X { uint32_t q[4][4]; int bs[4]; float yd; };
    struct Wr { u32x4 l0, l1, h; u32x2 s; uint16_t d; };
    static __device__ __forceinline__ void load_x(X & x, const uint8_t * xq, int K, int u) {
        const int b = u >> 2, hh = (u >> 1) & 1, v = u & 1;
        const int16_t * bs = (const int16_t *) (xq + K + (K / 256) * 4) + b * 16 + 8 * hh + v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 t = *(const u32x4 *) (xq + b * 256 + 128 * hh + 32 * k + 16 * v);
            for (int i = 0; i < 4; ++i) x.q[k][i] = t[i];
            x.bs[k] = bs[2 * k];
        }
        x.yd = ((const float *) (xq + K))[b];
    }
    static __device__ __forceinline__ void issue(Wr & w, const uint8_t * row, int K, int u) {
        const long nb = K / 256;
        const int b = u >> 2, hh = (u >> 1) & 1, v = u & 1;
        const uint8_t * ql = row + (long) b * 128 + 64 * hh + 16 * v;
        w.l0 = ld_nt16(ql);
        w.l1 = ld_nt16(ql + 32);
        w.h  = ld_nt16(row + nb * 128 + (long) b * 64 + 32 * hh + 16 * v);
        w.s  = ld_nt8(row + nb * 192 + (long) b * 16 + 8 * hh);
        w.d  = ld_nt2(row + nb * 208 + (long) b * 2);
    }
    static __device__ __forceinline__ float consume(const Wr & w, const X & x, int u, int & isum, int & msum) {
        const int v = u & 1;
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t h = w.h[i];
            acc[0] = dot4((w.l0[i] & 0x0F0F0F0Fu)        | ((h << 4) & 0x30303030u), x.q[0][i], acc[0]);
            acc[1] = dot4((w.l1[i] & 0x0F0F0F0Fu)        | ((h << 2) & 0x30303030u), x.q[1][i], acc[1]);
            acc[2] = dot4(((w.l0[i] >> 4) & 0x0F0F0F0Fu) | (h & 0x30303030u),        x.q[2][i], acc[2]);
            acc[3] = dot4(((w.l1[i] >> 4) & 0x0F0F0F0Fu) | ((h >> 2) & 0x30303030u), x.q[3][i], acc[3]);
        }
        const uint64_t s8 = ((uint64_t) w.s[1] << 32) | w.s[0];
        isum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int sc = (int) (int8_t) (s8 >> (8 * (v + 2 * k)));
            isum += sc * (acc[k] - 32 * x.bs[k]);
        }
        msum = 0;
        return x.yd * h2f(w.d) * (float) isum;
    }
};

// ---------------------------------------------------------------- Q8_0 (row-SoA) --------------------
// row: qs[nb32][32] | d[nb32];  activations: Q8_0 row-SoA (qs[K] | half d[K/32]).   unit = one 32-block
template <> struct QT<PM_Q8_0> {
    static constexpr int VPU = 32;
    struct X { uint32_t q[8]; float yd; };
    struct Wr { u32x4 q0, q1; uint16_t d; };
    static __device__ __forceinline__ void load_x(X & x, const uint8_t * xq, int K, int u) {
        const u32x4 a = *(const u32x4 *) (xq + u * 32), b = *(const u32x4 *) (xq + u * 32 + 16);
        for (int i = 0; i < 4; ++i) { x.q[i] = a[i]; x.q[4 + i] = b[i]; }
        x.yd = h2f(((const uint16_t *) (xq + K))[u]);
    }
    static __device__ __forceinline__ void issue(Wr & w, const uint8_t * row, int K, int u) {
        w.q0 = ld_nt16(row + (long) u * 32);
        w.q1 = ld_nt16(row + (long) u * 32 + 16);
        w.d  = ld_nt2(row + (long) K + (long) u * 2);
    }
    static __device__ __forceinline__ float consume(const Wr & w, const X & x, int, int & isum, int & msum) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { s = dot4(w.q0[i], x.q[i], s); s = dot4(w.q1[i], x.q[4 + i], s); }
        isum = s; msum = 0;
        return (float) s * (h2f(w.d) * x.yd);
    }
};

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + expf(-g)); }

// One workgroup: rows [row0, row0 + rows_per_wg) of W (and W2 when PAIR).
//   tpr threads cooperate on a row; 256/tpr rows are processed side by side ("slots");
//   R = rows in flight per slot; UPT = units per thread.
template <int TYPE, int UPT, int R, bool PAIR, bool DBG>
__global__ __launch_bounds__(256) void gemv_q_kernel(GemvP p) {
    typedef QT<TYPE> T;
    constexpr int NM = PAIR ? 2 : 1;
    __shared__ float red[2][R * NM * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tpr = p.tpr, slots = 256 / tpr, wps = tpr >> 6 /*waves per slot*/;
    const int slot = tid / tpr, tin = tid - slot * tpr;

    typename T::X x[UPT];
    bool valid[UPT];
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        const int u = tin + i * tpr;
        valid[i] = u < p.U;
        if (valid[i]) T::load_x(x[i], p.xq, p.K, u);
    }

    const int row0 = blockIdx.x * p.rows_per_wg;
    const int row1 = min(row0 + p.rows_per_wg, p.N);
    int it = 0;
    for (int base = row0; base < row1; base += slots * R, ++it) {
        typename T::Wr w[R][NM][UPT];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = base + r * slots + slot;
            if (row < row1) {
#pragma unroll
                for (int i = 0; i < UPT; ++i) if (valid[i]) {
                    T::issue(w[r][0][i], p.W + (long) row * p.row_bytes, p.K, tin + i * tpr);
                    if (PAIR) T::issue(w[r][NM - 1][i], p.W2 + (long) row * p.row_bytes, p.K, tin + i * tpr);
                }
            }
        }
        float * rb = red[it & 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = base + r * slots + slot;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                float acc = 0.0f;
                if (row < row1) {
#pragma unroll
                    for (int i = 0; i < UPT; ++i) if (valid[i]) {
                        int isum, msum;
                        acc += T::consume(w[r][m][i], x[i], tin + i * tpr, isum, msum);
                        if (DBG) {
                            int32_t * o = p.dbg + ((long) (m * p.N + row) * p.U + (tin + i * tpr)) * 2;
                            o[0] = isum; o[1] = msum;
                        }
                    }
                }
                acc = wave_sum(acc);
                if (lane == 0) rb[(r * NM + m) * 4 + wave] = acc;
            }
        }
        __syncthreads();
        // finalize: thread t < R*slots handles (r = t / slots, slot s = t % slots)
        if (tid < R * slots) {
            const int r = tid / slots, s = tid - r * slots;
            const int row = base + r * slots + s;
            if (row < row1) {
                float v[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    float t = 0.0f;
                    for (int k = 0; k < wps; ++k) t += rb[(r * NM + m) * 4 + s * wps + k];
                    v[m] = t;
                }
                float out = PAIR ? silu_f(v[0]) * v[NM - 1] : v[0];
                if (p.bias)  out += p.bias[row];
                if (p.resid) out += p.resid[row];
                p.y[row] = out;
            }
        }
    }
}

template <int TYPE, int UPT, int R, bool PAIR>
int launch_t(const GemvP & p, int grid, bool dbg, hipStream_t st) {
    if (dbg) hipLaunchKernelGGL((gemv_q_kernel<TYPE, UPT, R, PAIR, true>), dim3(grid), dim3(256), 0, st, p);
    else     hipLaunchKernelGGL((gemv_q_kernel<TYPE, UPT, R, PAIR, false>), dim3(grid), dim3(256), 0, st, p);
    return 0;
}

template <int TYPE>
int launch_type(const GemvP & p, int upt, bool pair, int grid, bool dbg, hipStream_t st) {
    // rows in flight per slot: keep (R * UPT * NM) around 4 x 16-B load groups per thread
    if (!pair) {
        switch (upt) {
            case 1: return launch_t<TYPE, 1, 4, false>(p, grid, dbg, st);
            case 2: return launch_t<TYPE, 2, 2, false>(p, grid, dbg, st);
            case 4: return launch_t<TYPE, 4, 1, false>(p, grid, dbg, st);
        }
    } else {
        switch (upt) {
            case 1: return launch_t<TYPE, 1, 2, true>(p, grid, dbg, st);
            case 2: return launch_t<TYPE, 2, 1, true>(p, grid, dbg, st);
            case 4: return launch_t<TYPE, 4, 1, true>(p, grid, dbg, st);
        }
    }
    return -1;
}

} // namespace

size_t pm_weight_row_bytes(int type, int64_t K) {
    switch (type) {
        case PM_F32:  return (size_t) K * 4;
        case PM_F16:  return (size_t) K * 2;
        case PM_Q8_0: return (size_t) (K / 32) * PM_BS_Q8_0;
        case PM_Q4_K: return (size_t) (K / 256) * PM_BS_Q4_K;
        case PM_Q5_K: return (size_t) (K / 256) * PM_BS_Q5_K;
        case PM_Q6_K: return (size_t) (K / 256) * PM_BS_Q6_K;
    }
    return 0;
}

size_t pm_weight_row_stride(int type, int64_t K) {
    switch (type) {
        case PM_Q8_0: { const size_t nb = K / 32;  return nb * 32 + ((nb * 2 + 15) & ~(size_t) 15); }
        case PM_Q6_K: { const size_t nb = K / 256; return nb * 208 + ((nb * 2 + 15) & ~(size_t) 15); }
    }
    return pm_weight_row_bytes(type, K);
}

int pm_launch_gemv(const pm_gemv_args & a, hipStream_t st) {
    const int vpu = a.type == PM_Q6_K ? 64 : 32;
    if (a.K % 256 != 0 && !(a.type == PM_Q8_0 && a.K % 32 == 0)) return -2;
    if (a.type == PM_Q8_0 && a.K % 32 != 0) return -3;
    GemvP p;
    p.K = a.K; p.N = a.N; p.U = a.K / vpu;
    p.row_bytes = (long) pm_weight_row_stride(a.type, a.K);
    p.tpr = p.U >= 256 ? 256 : (p.U > 64 ? 128 : 64);
    int upt = (p.U + p.tpr - 1) / p.tpr;
    upt = upt <= 1 ? 1 : (upt <= 2 ? 2 : 4);
    if ((long) upt * p.tpr < p.U) return -4;                   // K too large for one pass (K <= 32768 for 32-w units)
    const bool pair = a.W2 != nullptr;
    const int slots = 256 / p.tpr;
    const int R = pair ? (upt == 1 ? 2 : 1) : (upt == 1 ? 4 : (upt == 2 ? 2 : 1));
    const int batch = slots * R;
    // ~4 workgroups per CU when there is enough work; whole batches per workgroup
    int rpw = (a.N + 1023) / 1024;
    rpw = ((rpw + batch - 1) / batch) * batch;
    p.rows_per_wg = rpw;
    const int grid = (a.N + rpw - 1) / rpw;
    for (int c = 0; c < a.ncols; ++c) {
        const size_t xrow = a.type == PM_Q8_0 ? pm_q80_row_bytes(a.K) : pm_q8k_row_bytes(a.K);
        p.W = (const uint8_t *) a.W; p.W2 = (const uint8_t *) a.W2;
        p.xq = (const uint8_t *) a.xq + (size_t) c * xrow;
        p.y = a.y + (size_t) c * a.y_stride;
        p.bias = a.bias; p.resid = a.resid ? a.resid + (size_t) c * a.y_stride : nullptr;
        p.dbg = a.dbg_int;
        int rc;
        switch (a.type) {
            case PM_Q4_K: rc = launch_type<PM_Q4_K>(p, upt, pair, grid, a.dbg_int != nullptr, st); break;
            case PM_Q5_K: rc = launch_type<PM_Q5_K>(p, upt, pair, grid, a.dbg_int != nullptr, st); break;
            case PM_Q6_K: rc = launch_type<PM_Q6_K>(p, upt, pair, grid, a.dbg_int != nullptr, st); break;
            case PM_Q8_0: rc = launch_type<PM_Q8_0>(p, upt, pair, grid, a.dbg_int != nullptr, st); break;
            default: return -1;
        }
        if (rc) return rc;
    }
    return 0;
}
