// pm355_device.h — device-side helpers shared by the gfx950 kernels (wave64, CDNA4).
//
// GGUF block formats consumed here (reference: ggml/src/ggml-common.h:187-191, :286-335):
//   Q4_K 144 B / 256 w : half d | half dmin | u8 scales[12] | u8 qs[128]         (native layout in HBM)
//   Q5_K 176 B / 256 w : half d | half dmin | u8 scales[12] | u8 qh[32] | u8 qs[128]   (native)
//   Q6_K 210 B / 256 w : u8 ql[128] | u8 qh[64] | i8 scales[16] | half d         (row-SoA in HBM, see repack.hip)
//   Q8_0  34 B /  32 w : half d | i8 qs[32]                                      (row-SoA in HBM)
//   Q8_K 292 B / 256 a : float d | i8 qs[256] | i16 bsums[16]                    (activations, native)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define PM_QK_K 256
#define PM_WAVE 64

enum pm_type : int { PM_F32 = 0, PM_F16 = 1, PM_Q8_0 = 8, PM_Q4_K = 12, PM_Q5_K = 13, PM_Q6_K = 14, PM_Q8_K = 15 };

#define PM_BS_Q8_0 34
#define PM_BS_Q4_K 144
#define PM_BS_Q5_K 176
#define PM_BS_Q6_K 210
#define PM_BS_Q8_K 292

// Q6_K row-SoA tail (round 5): after the three 64-byte-per-block streams (la | lb | qh: nb * 192 bytes) come, per GROUP of 8 blocks, the 8 x 16 int8
// scales followed by the 8 fp16 super-block scales d (144 bytes): the 2-byte d a mat-vec lane loads sits in the cache lines its scale load has just
// requested (separate `scales[nb][16] | d[nb]` streams: the 32 bytes of d a wave step needs were a line of their own, requested again by each of the
// next three steps). -DPM_Q6K_SCD=0 keeps the separate streams (A/B). Same bytes per row whenever K % 2048 == 0.
// Measured (profiles/r05_ab_sumsq_q6k_tail.txt): the grouped tail changes nothing and tail loads through the caches cost 2 % of the 70B token
// (the nt loads' re-fetch of the d line is the cheaper evil) - the default stays the round-4 form; -DPM_Q6K_SCD=1 builds the grouped tail.
#ifndef PM_Q6K_SCD
#define PM_Q6K_SCD 0
#endif
// (a last group of r < 8 blocks is scales[r][16] | d[r]: the row is 210 bytes per block rounded up to 16 in both forms)
__host__ __device__ inline uint32_t pm_q6k_sc_off(uint32_t nb, uint32_t b) { return PM_Q6K_SCD ? nb * 192u + (b >> 3) * 144u + (b & 7u) * 16u : nb * 192u + b * 16u; }
__host__ __device__ inline uint32_t pm_q6k_d_off(uint32_t nb, uint32_t b) {
    const uint32_t left = nb - (b & ~7u), cnt = left < 8u ? left : 8u;                 // blocks in b's group
    return PM_Q6K_SCD ? nb * 192u + (b >> 3) * 144u + cnt * 16u + (b & 7u) * 2u : nb * 208u + b * 2u;
}
__host__ __device__ inline size_t pm_q6k_row_stride(size_t nb) { return (nb * 210 + 15) & ~(size_t) 15; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- loads --------------------------------------------------------------------------------------
// Weights are streamed exactly once per token: non-temporal (global_load ... nt) keeps them from
// displacing the activation / KV working set in L2 (MI355X_MICROARCH.md "nt-weights").
// Every pointer these helpers see is device-global memory. The explicit address-space cast makes the backend emit
// global_load / global_store even where it cannot prove it (pointers read from a descriptor in memory
// would otherwise become FLAT accesses, which count on vmcnt AND lgkmcnt and force
// s_waitcnt vmcnt(0) lgkmcnt(0) in the row loop).
#define PM_G __attribute__((address_space(1)))
__device__ __forceinline__ u32x4 ld_nt16(const void * p) { return __builtin_nontemporal_load((const PM_G u32x4 *) p); }
__device__ __forceinline__ u32x2 ld_nt8(const void * p)  { return __builtin_nontemporal_load((const PM_G u32x2 *) p); }
__device__ __forceinline__ uint32_t ld_nt4(const void * p) { return __builtin_nontemporal_load((const PM_G uint32_t *) p); }
__device__ __forceinline__ uint16_t ld_nt2(const void * p) { return __builtin_nontemporal_load((const PM_G uint16_t *) p); }
// A pointer that arrives in VGPRs (function argument) but is the same in every lane and points to read-only memory:
// rebuild it from SGPRs in the constant address space, so that what is read through it is fetched with s_load into SGPRs
// (and may be hoisted out of loops) instead of with per-lane FLAT loads.
#define PM_C __attribute__((address_space(4)))
template <typename T> __device__ __forceinline__ const T * uniform_const_ptr(const T * p) {
    const uint64_t u = (uint64_t) p;
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) u), hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (u >> 32));
    return (const T *) (const PM_C T *) (((uint64_t) hi << 32) | lo);
}
// same, for a pointer to WRITABLE global memory: only made scalar (the accesses through it cast to PM_G themselves)
template <typename T> __device__ __forceinline__ T * uniform_ptr(T * p) {
    const uint64_t u = (uint64_t) p;
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) u), hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (u >> 32));
    return (T *) (((uint64_t) hi << 32) | lo);
}
// plain (cached) global loads / stores
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ T ld_g(const T * p) { return *(const PM_G T *) p; }
__device__ __forceinline__ float4 ld_g(const float4 * p) { const f32x4 t = *(const PM_G f32x4 *) p; return make_float4(t.x, t.y, t.z, t.w); }
template <typename T> __device__ __forceinline__ void st_g(T * p, T v) { *(PM_G T *) p = v; }

__device__ __forceinline__ float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }   // RNE, == GGML_FP32_TO_FP16

// v_dot4_i32_i8: 4 x (i8 * i8) + acc
__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int acc) {
    return __builtin_amdgcn_sdot4((int) a, (int) b, acc, false);
}

// ---- DPP wave reductions (no LDS traffic) ----------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
    // old = 0 with the selected row mask and bound_ctrl: lanes that receive nothing add 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}

// Sum over the 64 lanes of a wave; the total is returned in EVERY lane (via readlane 63).
// Fixed combination order -> bitwise deterministic.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);         // row_half_mirror : 8-lane groups complete
    v += dpp_f<0x140>(v);         // row_mirror      : 16-lane rows complete
    v += dpp_f<0x142, 0xA>(v);    // row_bcast15 -> rows 1 and 3
    v += dpp_f<0x143, 0xC>(v);    // row_bcast31 -> rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// max over aligned groups of 8 lanes, result in all 8 lanes
__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    return v;
}
__device__ __forceinline__ int group8_min_i(int v) {
    v = min(v, dpp_i<0xB1>(v));
    v = min(v, dpp_i<0x4E>(v));
    v = min(v, dpp_i<0x141>(v));
    return v;
}

// round-to-nearest-even of |v| <= 4194303 (reference: nearest_int, ggml-quants.c:1638-1644)
__device__ __forceinline__ int nearest_int_rne(float v) {
    float t = v + 12582912.f;
    return (__builtin_bit_cast(int, t) & 0x007fffff) - 0x00400000;
}

// 6-bit scale / min number `j` (0..7) of a K-quant super-block; s = the 12 scale bytes as 3 dwords
// (reference: get_scale_min_k4, ggml-quants.c:1898-1906)
__device__ __forceinline__ void k4_scale_min(uint32_t s0, uint32_t s1, uint32_t s2, int j, int & sc, int & mn) {
    // branch-free (j differs between lanes: a divergent branch here would split the wave in the hot loop)
    const int sh = 8 * (j & 3);
    const uint32_t b0 = (s0 >> sh) & 0xFF, b1 = (s1 >> sh) & 0xFF, b2 = (s2 >> sh) & 0xFF;
    const int sc_lo = b0 & 63, mn_lo = b1 & 63;
    const int sc_hi = (b2 & 0x0F) | ((b0 >> 6) << 4), mn_hi = (b2 >> 4) | ((b1 >> 6) << 4);
    sc = j < 4 ? sc_lo : sc_hi;
    mn = j < 4 ? mn_lo : mn_hi;
}

// Both 6-bit scale/min numbers (2p, 2p+1) of pair p (0..3) at once, branch-free, on 16-bit byte pairs.
__device__ __forceinline__ void k4_scale_min_pair(uint32_t s0, uint32_t s1, uint32_t s2, int p, int & sc0, int & sc1, int & m0, int & m1) {
    const int sh = 16 * (p & 1);
    const uint32_t a = s0 >> sh, b = s1 >> sh, c = s2 >> sh;           // bytes (2p, 2p+1) mod 4 of each scale dword
    const uint32_t sc_lo = a & 0x3f3fu, m_lo = b & 0x3f3fu;            // numbers 0..3: low 6 bits of bytes 0..3 / 4..7
    const uint32_t sc_hi = (c & 0x0f0fu) | ((a & 0xc0c0u) >> 2);       // numbers 4..7: nibbles of bytes 8..11 + top 2 bits
    const uint32_t m_hi  = ((c >> 4) & 0x0f0fu) | ((b & 0xc0c0u) >> 2);
    const uint32_t sc = p < 2 ? sc_lo : sc_hi, m = p < 2 ? m_lo : m_hi;
    sc0 = (int) (sc & 0xFF); sc1 = (int) (sc >> 8);
    m0  = (int) (m & 0xFF);  m1  = (int) (m >> 8);
}

// ---- seam anatomy (measurement builds only: -DPM_TS, tools/seam_anatomy.py) ----------------------------------------
// Every instrumented launch gets a slot of PM_TS_WGS x 8 timestamps of the chip-wide 100 MHz counter (s_memrealtime: comparable
// across workgroups, XCDs and kernels); thread 0 of a workgroup keeps its stamps in SGPRs and stores them when it leaves.
#define PM_TS_WGS 256
#ifdef PM_TS
#define PM_TS_NOW() __builtin_amdgcn_s_memrealtime()
unsigned long long * pm_ts_next_slot();          // host (ts.hip): device address of the next launch's slot, or null when disabled
#else
#define PM_TS_NOW() 0ull
static inline unsigned long long * pm_ts_next_slot() { return nullptr; }
#endif
__device__ __forceinline__ void pm_ts_store(unsigned long long * ts, int tag, const unsigned long long (&t)[6]) {
#ifdef PM_TS
    if (ts && threadIdx.x == 0 && blockIdx.x < PM_TS_WGS) {
        unsigned long long * o = ts + (size_t) blockIdx.x * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = t[i];
        o[6] = (unsigned long long) tag; o[7] = (unsigned long long) gridDim.x;
    }
#endif
}

// ---- device-coherent activation traffic inside one launch ------------------------------------------------------
// Activations written by one workgroup and read by another INSIDE one kernel (attn_wo.hip) go through agent-scope
// relaxed atomics: global_load / global_store ... sc1, which are coherent across the 8 XCD L2s without any cache
// write-back / invalidate. COH = false: plain accesses (stand-alone launches: kernel boundaries do the maintenance).
// (loads: SYSTEM scope = `sc0 sc1` - an `sc1` load may be served from a line this XCD's L2 kept from an earlier read of the same buffer in the same
//  launch; the persistent decode engine re-uses every hand-off buffer once per layer, round 5)
template <bool COH> __device__ __forceinline__ float ld_act(const float * p) {
    if (COH) return __hip_atomic_load((const PM_G float *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return ld_g(p);
}
template <bool COH> __device__ __forceinline__ float4 ld_act4(const float4 * p) {
    if (COH) {
        const PM_G float * f = (const PM_G float *) p;
        float4 r;
        r.x = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        r.y = __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        r.z = __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        r.w = __hip_atomic_load(f + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return r;
    }
    return ld_g(p);
}
// the same for the 2- and 4-byte stores of the QKV epilogue (K / V cache cells, F16 pairs): COH = agent-scope write-through (global_store ... sc1)
template <bool COH, typename T> __device__ __forceinline__ void st_any(T * p, T v) {
    if (COH) __hip_atomic_store((PM_G T *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else st_g(p, v);
}
template <bool COH> __device__ __forceinline__ void st_act(float * p, float v) {
    if (COH) __hip_atomic_store((PM_G float *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else st_g(p, v);
}
