// attn_tail_device.h — single-token attention over cached cells for callers INSIDE a launch that also produced q and this token's cell
// (the QKV launch's tail: mmvq.hip; the persistent engine's attention phase: decode_engine.hip). The arithmetic, its order and its rounding points are
// attn_cached.hip's (ggml.c:12445-12473 K.q rows, :13783-13879 soft_max, F16-rounded probabilities against the F16 V rows: the reference's
// ggml_compute_forward_mul_mat with an F16 src1 conversion) - the outputs are the same bits as attn_cached.hip's OF THE SAME ROUND (for 65-256 cells both take the
// keys-in-the-lanes form of round 5, whose fmaf chains over 4-wave partial sums differ in the low-order bits from the per-head body the round-4 path ran there);
// what differs is how memory is read: q and the newest
// cell were written by OTHER compute units in this launch, so every load of q / K / V bypasses this XCD's caches (sc0 sc1).
#pragma once
#include "pm355_device.h"

// ---- coherent buffer loads: tracked by the compiler's waitcnt logic, 16 bytes per instruction. `sc0 sc1` (aux 17), not `sc1` alone: a buffer that is
// handed over more than once per launch (the residual stream, q, the attention output, the ffn activation, the partial sums - every layer re-uses
// them) may still sit in THIS XCD's L2 from the previous layer's read, and an sc1 load is served from there: two 70B layers differed by 1e-4 from
// the five launches, one layer by 1e-14 (found on the hardware). The write side stays sc1 write-through (st_act<true>). --------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void * base) { return __builtin_amdgcn_make_buffer_rsrc((void *) base, 0, 0x7FFFFFF0, 0x00020000); }
__device__ __forceinline__ float4 coh_ld16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    // (the whole vector is re-typed at once: element-wise __builtin_bit_cast(float, t[i]) of the loaded vector compiles to a ONE-dword load that
    //  feeds all four components - ROCm 7.2 clang; found on the hardware, tools/r5)
    const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int) off, 0, 17));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ u32x4 coh_ld16u(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int) off, 0, 17)); }
__device__ __forceinline__ float coh_ld4(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int) off, 0, 17)); }


// ---- up to 64 * KPL cells with the KEYS IN THE LANES (round 5; KPL = 1 is attn_cached.hip's short path, one key per lane): lane l owns keys l, l + 64, ...,
// wave w the head dimensions [DPW w, DPW w + DPW); every load is issued before anything is computed, ONE barrier of the four waves. The general body
// (thread per key, three reductions, a score buffer) costs 7.1 us per launch at 65 cells and 9.1 at 256 where this form costs ~4 (tools/r5/attn_short_probe.py).
// Rounding points as everywhere: q, K, V F16, the normalised probabilities rounded to F16, f32 accumulation. n_kv <= 64 * KPL <= n_ctx, DH 64 / 128.
// COH: q / K / V through cache-bypassing loads, the output written through (callers inside a launch that produced them: attn_tail_head below).
// part / pw: LDS [4][64 * KPL] floats each; qstrip: LDS [DH] floats (COH only); bar(): barrier of exactly the four waves.
template <int DH, int KPL, bool COH, class Bar>
__device__ __forceinline__ void attn_keys_in_lanes(const float * q_head, const uint16_t * kc, const uint16_t * vc, float * out_head, int hk, int Hkv, int n_ctx, int n_kv,
                                                   float scale, const void * mask, int mask_f16, float * part, float * pw, float * qstrip, int lane, int wave, Bar && bar) {
    constexpr int NKEY = 64 * KPL, DPW = DH / 4, NK = DPW / 8, KP = 64 / DPW, KPP = NKEY / KP, NV = KPP / 8;
    u32x4 kreg[KPL][NK], vreg[NV];
    const __amdgpu_buffer_rsrc_t rk = coh_rsrc(kc), rv = coh_rsrc(vc), rq = coh_rsrc(q_head);
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const long koff = (long) (lane + 64 * k) * Hkv * DH + (long) hk * DH + DPW * wave;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            if constexpr (COH) kreg[k][j] = coh_ld16u(rk, (uint32_t) ((koff + 8 * j) * 2));
            else kreg[k][j] = *(const PM_G u32x4 *) (kc + koff + 8 * j);
        }
    }
    const int e = lane % DPW, kp = lane / DPW;
    const long vrow = (long) (hk * DH + DPW * wave + e) * n_ctx + KPP * kp;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if constexpr (COH) vreg[j] = coh_ld16u(rv, (uint32_t) ((vrow + 8 * j) * 2));
        else vreg[j] = *(const PM_G u32x4 *) (vc + vrow + 8 * j);
    }
    const float * qw;
    if constexpr (COH) {
        float * qs = qstrip + wave * DPW;
        if (lane < DPW) qs[lane] = coh_ld4(rq, (uint32_t) ((DPW * wave + lane) * 4));
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        qw = qs;
    } else qw = uniform_const_ptr(q_head + DPW * wave);
    float m_add[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int key = lane + 64 * k;
        m_add[k] = (mask && key < n_kv) ? (mask_f16 ? h2f(((const PM_G uint16_t *) mask)[key]) : ((const PM_G float *) mask)[key]) : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc = fmaf(h2f((uint16_t) (kreg[k][j][t] & 0xFFFF)), qw[8 * j + 2 * t], acc);
                acc = fmaf(h2f((uint16_t) (kreg[k][j][t] >> 16)), qw[8 * j + 2 * t + 1], acc);
            }
        part[wave * NKEY + lane + 64 * k] = acc;
    }
    bar();
    float s_[KPL], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int key = lane + 64 * k;
        s_[k] = key < n_kv ? ((part[key] + part[NKEY + key]) + (part[2 * NKEY + key] + part[3 * NKEY + key])) * scale + m_add[k] : -INFINITY;
        mx = fmaxf(mx, s_[k]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float ex[KPL];
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < KPL; ++k) { ex[k] = lane + 64 * k < n_kv ? expf(s_[k] - mx) : 0.0f; tot += (double) ex[k]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    const float inv = (float) (1.0 / tot);
#pragma unroll
    for (int k = 0; k < KPL; ++k) pw[wave * NKEY + lane + 64 * k] = h2f(f2h(ex[k] * inv));        // p rounded to F16 (src1 of the V^T.p product)
    __builtin_amdgcn_wave_barrier();                       // (a wave's LDS accesses execute in order: its own copy needs no workgroup barrier)
    float o = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            o = fmaf(h2f((uint16_t) (vreg[j][t] & 0xFFFF)), pw[wave * NKEY + KPP * kp + 8 * j + 2 * t], o);
            o = fmaf(h2f((uint16_t) (vreg[j][t] >> 16)), pw[wave * NKEY + KPP * kp + 8 * j + 2 * t + 1], o);
        }
#pragma unroll
    for (int off = DPW; off < 64; off <<= 1) o += __shfl_xor(o, off);
    if (lane < DPW) st_act<COH>(out_head + DPW * wave + lane, o);
}

struct AttnTailP {
    const float * q; const uint16_t * kc, * vc; const int32_t * pos, * seq; long seq_stride; float * out;
    int H, Hkv, n_ctx; float scale;
};
// LDS bytes of the body for head_dim DH and up to max_keys cells: part[4][256] / pw[4][256] / reductions | qs[DH] | part[256] | sc[max_keys + 8]
static inline __host__ __device__ size_t attn_tail_lds(int dh, int max_keys) { return (size_t) (2048 + 8 + 8) * 4 + (size_t) (dh + 256 + ((max_keys + 15) & ~7)) * 4; }

// head h by FOUR waves (threads 0 .. 255 of the caller's workgroup); bar() = a barrier of exactly those four waves that also orders their LDS traffic.
// attn_cached.hip's short path (<= 64 cells, one barrier) and the cached form of attn_rope_body beyond.
template <int DH, class Bar>
__device__ __forceinline__ void attn_tail_head(const AttnTailP & a, int h, char * smem, Bar && bar) {
    // (the thread id goes through an opaque asm: everything derived from it - dozens of LDS / cache offsets - is then recomputed here instead of being
    //  hoisted to the caller's entry and kept alive, i.e. spilled, across its row loops)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, Hkv = a.Hkv, n_ctx = a.n_ctx;
    const int hk = h / (H / Hkv);
    const float scale = a.scale;
    const int seq = a.seq ? uniform_const_ptr(a.seq)[0] : 0;
    const int n_kv = uniform_const_ptr(a.pos)[seq] + 1;
    const long soff = (long) seq * a.seq_stride;
    const __amdgpu_buffer_rsrc_t rk = coh_rsrc(a.kc + soff), rv = coh_rsrc(a.vc + soff), rq = coh_rsrc(a.q);
    float * part = (float *) smem;                         // [4][256]
    float * pw = part + 1024;                              // [4][256]
    float * redf = pw + 1024;                              // [8]
    double * redd = (double *) (redf + 8);                 // [4]
    float * body = (float *) (redd + 4);                   // general path: qs[DH] | part[256] | sc[max_keys]
    if (n_kv <= 64 && n_ctx >= 64) {
        constexpr int DPW = DH / 4, NK = DPW / 8, KP = 64 / DPW, KPP = 64 / KP, NV = KPP / 8;
        const int key = lane < n_ctx ? lane : 0;
        u32x4 kreg[NK], vreg[NV];
#pragma unroll
        for (int j = 0; j < NK; ++j) kreg[j] = coh_ld16u(rk, (uint32_t) (((long) key * Hkv * DH + (long) hk * DH + DPW * wave + 8 * j) * 2));
        const int e = lane % DPW, kp = lane / DPW;
        const long vrow = (long) (hk * DH + DPW * wave + e) * n_ctx + (KPP * kp + KPP <= n_ctx ? KPP * kp : 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) vreg[j] = coh_ld16u(rv, (uint32_t) ((vrow + 8 * j) * 2));
        // this wave's q slice -> its own LDS strip, read back as broadcasts (a scalar load could hit a stale scalar-cache line: the rows were written by
        // other CUs in this launch; DPW values in registers next to the K / V pieces spill)
        float * qw = body + wave * DPW;
        if (lane < DPW) qw[lane] = coh_ld4(rq, (uint32_t) (((long) h * DH + DPW * wave + lane) * 4));
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc = fmaf(h2f((uint16_t) (kreg[j][t] & 0xFFFF)), qw[8 * j + 2 * t], acc);
                acc = fmaf(h2f((uint16_t) (kreg[j][t] >> 16)), qw[8 * j + 2 * t + 1], acc);
            }
        part[wave * 64 + lane] = acc;
        bar();
        const bool valid = lane < n_kv;
        const float s_ = valid ? ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane])) * scale : -INFINITY;
        float mx = s_;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const float ex = valid ? expf(s_ - mx) : 0.0f;
        double tot = (double) ex;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        const float inv = (float) (1.0 / tot);
        pw[wave * 64 + lane] = h2f(f2h(ex * inv));         // p rounded to F16 (src1 of the V^T.p product)
        __builtin_amdgcn_wave_barrier();
        float o = 0.0f;
        const bool chunk_ok = KPP * kp + KPP <= n_ctx;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                o = fmaf(h2f((uint16_t) (vreg[j][t] & 0xFFFF)), pw[wave * 64 + KPP * kp + 8 * j + 2 * t], o);
                o = fmaf(h2f((uint16_t) (vreg[j][t] >> 16)), pw[wave * 64 + KPP * kp + 8 * j + 2 * t + 1], o);
            }
        if (!chunk_ok) o = 0.0f;
#pragma unroll
        for (int off = DPW; off < 64; off <<= 1) o += __shfl_xor(o, off);
        if (lane < DPW) st_act<true>(a.out + (long) h * DH + DPW * wave + lane, o);
        return;
    }
    // ---- up to 256 cells: keys in the lanes, two / four per lane (the same function attn_cached.hip calls: same bits)
    if constexpr (DH <= 128) {
        if (n_kv <= 128 && n_ctx >= 128) { attn_keys_in_lanes<DH, 2, true>(a.q + (long) h * DH, a.kc + soff, a.vc + soff, a.out + (long) h * DH, hk, Hkv, n_ctx, n_kv, scale, nullptr, 0, part, pw, body, lane, wave, bar); return; }
        if (n_kv <= 256 && n_ctx >= 256) { attn_keys_in_lanes<DH, 4, true>(a.q + (long) h * DH, a.kc + soff, a.vc + soff, a.out + (long) h * DH, hk, Hkv, n_ctx, n_kv, scale, nullptr, 0, part, pw, body, lane, wave, bar); return; }
    }
    // ---- general cached body (attn_rope_body<DH, COH, 0, true>): thread per key, three 4-wave reductions
    constexpr int PARTS = 256 / DH, KQ = DH / 8;
    float * qs = body, * part2 = qs + DH, * sc = part2 + 256;
    if (tid < DH) qs[tid] = coh_ld4(rq, (uint32_t) (((long) h * DH + tid) * 4));
    const int ve = tid % DH, vpt = tid / DH;
    const int n_pad = (n_kv + 7) & ~7;
    bar();
    float lmax = -INFINITY;
#pragma unroll 1
    for (int i = tid; i < n_kv; i += 256) {
        // (the key's row in two halves: same accumulation order as one pass, half the registers)
        float acc = 0.0f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            u32x4 kk[KQ / 2];
#pragma unroll
            for (int j = 0; j < KQ / 2; ++j) kk[j] = coh_ld16u(rk, (uint32_t) (((long) i * Hkv * DH + (long) hk * DH + 8 * (hf * (KQ / 2) + j)) * 2));
#pragma unroll
            for (int jj = 0; jj < KQ / 2; ++jj)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc += h2f((uint16_t) (kk[jj][j] & 0xFFFF)) * qs[8 * (hf * (KQ / 2) + jj) + 2 * j];
                    acc += h2f((uint16_t) (kk[jj][j] >> 16)) * qs[8 * (hf * (KQ / 2) + jj) + 2 * j + 1];
                }
        }
        const float s_ = acc * scale;
        sc[i] = s_;
        lmax = fmaxf(lmax, s_);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if (lane == 0) redf[wave] = lmax;
    bar();
    const float mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double lsum = 0.0;
    for (int i = tid; i < n_kv; i += 256) {
        const float e = expf(sc[i] - mx);
        sc[i] = e;
        lsum += (double) e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
    if (lane == 0) redd[wave] = lsum;
    bar();
    const double tot = (redd[0] + redd[1]) + (redd[2] + redd[3]);
    const float inv = (float) (1.0 / tot);
    bar();
    for (int i = tid; i < n_pad; i += 256) sc[i] = i < n_kv ? h2f(f2h(sc[i] * inv)) : 0.0f;
    bar();
    {
        float acc = 0.0f;
        const long vrow = (long) (hk * DH + ve) * n_ctx;
#pragma unroll 1
        for (int i = vpt * 8; i < n_pad; i += PARTS * 8) {
            const u32x4 vv = coh_ld16u(rv, (uint32_t) ((vrow + i) * 2));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc += h2f((uint16_t) (vv[j] & 0xFFFF)) * sc[i + 2 * j];
                acc += h2f((uint16_t) (vv[j] >> 16)) * sc[i + 2 * j + 1];
            }
        }
        part2[tid] = acc;
    }
    bar();
    if (tid < DH) {
        float acc = 0.0f;
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) acc += part2[pt * DH + tid];
        st_act<true>(a.out + (long) h * DH + tid, acc);
    }
}

