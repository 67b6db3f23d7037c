// decode_engine.hip — the single-token layer stack as ONE persistent launch (round 5).
//
// What it replaces: the five launches per layer of engine.hip's run_layers_fused (mmvq.hip + attn_cached.hip), i.e. the same reference
// nodes - ggml_compute_forward_mul_mat over ggml_vec_dot_q4_K/q5_K/q6_K_q8_K with quantize_row_q8_K and rms_norm fused in front
// (ggml.c:12377, ggml-quants.c:7713 / 8281 / 8918 / 3785, ggml.c:11950), rope + F16 KV store (ggml.c:14143, src/llama.cpp:9673-9718),
// attention over cached cells (ggml.c:12445-12473, :13783-13879), silu * mul, residual adds - with the SAME device code (the row loops
// Item<>::run_job / run_job_split, QT<>::consume, q8k_rows_to_lds, write_out, write_out_qkv of mmvq_device.h, the attn_cached / attn_rope_body
// arithmetic): every phase's output is bit-identical to the launch it stands for (tests/test_gpu_ops.py, test_gpu_engine.py).
//
// Status: OPT-IN (PM355_ENGINE=1). Correct - every phase the same bits as its launch, 80 70B-shape layers bit-identical in hidden rows and logits, watchdog
// clean - and SLOWER than the five launches it replaces: 10.15-10.4 ms against 8.55 ms per 70B token, 2.23 against 1.67 ms on the 8B shape
// (profiles/r05_engine_measured.txt). The default path stays run_layers_fused. What was learnt, in the order it was measured:
//   1. The skeleton that motivated it (tools/csrc/engine_probe.hip, profiles/r05_engine_seam_cost.txt: 92 us against 110 us per layer) streams the real
//      bytes through the real barriers but consumes them with a token amount of arithmetic. With the real consume() the row loops are not a pure
//      stream: a CU needs most of its issue slots to keep its share of HBM busy, so nothing that "runs ahead" is free.
//   2. A dedicated loader wave filling an LDS ring for 15 consumers (all weights, or only the head of each phase): an LDS-DMA instruction issues every
//      ~150-190 cycles in a CU whose other waves stream - 14-20 GB/s per CU with one or two loader waves against the 26.5 GB/s a CU's share of HBM is;
//      as a head-only prefetcher next to 15 register-streaming waves the loader got 4 GB/s and its items arrived after the phase they were meant to
//      start. 13.8 / 10.6 ms per token.
//   3. This file's form - no loader, every wave prefetches the first steps of its OWN next rows into a private LDS slot before it goes to the barrier:
//      the prefetch does stream (37 MB in ~5 us, DMA issue blocks on the queue) but the token time is the same with slots of 0, 2304, 4608 or 9216 bytes
//      per wave (10.35 / 10.32 / 10.15 / 10.42 ms): what is prefetched has to be consumed afterwards at the same rate the registers would have
//      delivered it. The in-launch seam itself (timeline in r05_engine_measured.txt) is write-through stores + s_waitcnt 0.7-3.4 us, arrival 1 us,
//      waiting for the slowest workgroup 2-9 us, system-scope activation fetch + quantize 2.6-3.8 us: ~8 us against the 6.2 us a kernel boundary
//      costs, six of them per layer (the attention is a phase of its own) against five.
//
// Structure: one 1024-thread workgroup per CU (all resident: the device-wide barrier needs that), 16 waves that all do the same thing.
//   * rows: exactly the mat-vec kernel's row loops, HBM -> registers (global_load_dwordx4 nt, two register sets software-pipelined, no LDS staging).
//   * seam cover: a wave that has stored its results issues, BEFORE it waits at the barrier, LDS-DMA loads (global_load_lds_dwordx4) of the first
//     steps of ITS OWN rows of the next mat-vec phase into a wave-private LDS slot (ENG_SLOT_BYTES: one 8192-weight Q4_K row); they need no
//     registers and are consumed from LDS (ds_read_b128, same consume() calls, same order, the per-lane partial sums handed to the register row
//     loop: Item::run_job's c_start / acc0) while the wave's first register loads travel. Nobody waits for anybody else's prefetch.
//     (Wave 0 polls the device-wide barrier and every poll waits on vmcnt: it issues its prefetch after the barrier.)
//   * seam: results -> epilogue (bias / residual / silu*mul / RoPE + KV store, write-through sc1 stores) -> prefetch -> workgroup barrier ->
//     two-level device-wide arrival -> wait -> fetch the next activation row (sc0 sc1 loads) and quantize it (rms_norm from the producer-side
//     partial sums of squares: no reduction pass).
//   * hand-off buffers are write-once per launch (engine.hip gives every layer its own q / attention output / ffn activation / residual rows).
// Every wait is bounded; a time-out raises the launch's watchdog word and every later wait of that workgroup returns at once.
#include "mmvq_device.h"
#include "attn_device.h"
#include "attn_tail_device.h"
#include "pm355_engine.h"
#include <vector>
#include <stdio.h>

using namespace pmv;

namespace {

constexpr int ENG_NW = PM_GEMV_NW, ENG_THREADS = ENG_NW * 64;
static_assert(ENG_NW == 16, "one 1024-thread workgroup per CU");
constexpr int ENG_LDS = 163328;                            // dynamic LDS: activation row | 16 prefetch slots | parked results (+ the small static control block)
constexpr int ENG_OUTF = 640;                              // parked results per workgroup (floats), at the top of the dynamic LDS
constexpr int ENG_OUT_OFF = ENG_LDS - ENG_OUTF * 4;
#ifndef ENG_SLOT_BYTES
#define ENG_SLOT_BYTES 4608
#endif
constexpr int ENG_SLOT_MAX = ENG_SLOT_BYTES;                         // bytes a wave prefetches at most (two Q4_K rows of 8192 weights / one gate + up pair)
constexpr int ENG_MAXG = 1024;

struct EngPhase {
    int kind;                                              // 0 = mat-vec, 1 = attention over cached cells
    int ta, tb, pair, epi;
    GemvP g;
    const float * aq; uint16_t * akc, * avc; const int32_t * apos, * aseq; long aseq_stride; float * aout;
    int aH, aHkv, adh, an_ctx, amax_keys; float ascale;
};
struct EngArgs { const EngPhase * ph; int n_ph; unsigned * ctr; int * err; float * dbg; };

typedef __attribute__((address_space(3))) void * lds_vp;
typedef __attribute__((address_space(3))) unsigned lds_u32;

// LDS control block. Between a wave's prefetch and the first use of its slot the seam's own LDS traffic is inline asm with explicit lgkmcnt waits, and the
// workgroup barrier is a bare s_barrier: __syncthreads() carries a fence that drains vmcnt - the prefetch would be waited for AT the barrier instead
// of travelling through it.
struct Ctl { unsigned abar, giveup, pad_[2]; };

__device__ __forceinline__ lds_u32 * L(unsigned * p) { return (lds_u32 *) p; }
__device__ __forceinline__ void lds_st_asm(unsigned * p, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned) (uintptr_t) L(p)), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned lds_ld_asm(unsigned * p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned) (uintptr_t) L(p)) : "memory");
    return v;
}
__device__ __forceinline__ unsigned lds_inc_asm(unsigned * p) {      // returns the old value
    unsigned v;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned) (uintptr_t) L(p)), "v"(1u) : "memory");
    return v;
}
__device__ __forceinline__ void give_up(Ctl * c, int * err, int code) {
    lds_st_asm(&c->giveup, 1);
    __hip_atomic_store((PM_G int *) err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// workgroup barrier WITHOUT the fence __syncthreads() carries (that one drains vmcnt - and with it the prefetch): this wave's LDS traffic is done
// (lgkmcnt), then s_barrier
__device__ __forceinline__ void wg_bar() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// barrier of `n` waves (the attention's four) on a monotonic LDS counter (gen-th use completes at gen * n arrivals)
__device__ __forceinline__ void wbar(unsigned * ctr, unsigned & gen, int n, int lane, Ctl * c, int * err, int code) {
    ++gen;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
        lds_inc_asm(ctr);
        int spins = 0;
        while ((int) (lds_ld_asm(ctr) - gen * (unsigned) n) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (lds_ld_asm(&c->giveup)) break;
            if (++spins > (1 << 22)) { give_up(c, err, code); break; }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- device-wide barrier: groups of 16 workgroups count on their own line, the last of a group counts on the top line, the last of all publishes
// the phase number on every group's flag line (256 atomics on ONE counter cost 20 us; two levels 2.1 us: DESIGN_HISTORY.md section 6) -------------
__device__ __forceinline__ void g_arrive(unsigned * ctr, unsigned phase, unsigned ngroups, unsigned gsize, bool last) {
    unsigned * g = ctr + 32 * (1 + blockIdx.x / gsize);
    const unsigned old = __hip_atomic_fetch_add((PM_G unsigned *) g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old + 1) % gsize == 0) {
        const unsigned t = __hip_atomic_fetch_add((PM_G unsigned *) ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == ngroups * (phase + 1)) {
            if (last) {                                    // the launch's last arrival re-arms the counters (nobody waits on the last phase)
                __hip_atomic_store((PM_G unsigned *) ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned k = 0; k < ngroups; ++k) {
                    __hip_atomic_store((PM_G unsigned *) (ctr + 32 * (1 + k)), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((PM_G unsigned *) (ctr + 32 * (1 + ENG_MAXG / 16 + k)), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                for (unsigned k = 0; k < ngroups; ++k)
                    __hip_atomic_store((PM_G unsigned *) (ctr + 32 * (1 + ENG_MAXG / 16 + k)), phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
__device__ __forceinline__ void g_wait(unsigned * ctr, unsigned phase, unsigned gsize, Ctl * c, int * err) {   // phases 0 .. phase - 1 complete
    const unsigned * flag = ctr + 32 * (1 + ENG_MAXG / 16 + blockIdx.x / gsize);
    int spins = 0;
    while (__hip_atomic_load((const PM_G unsigned *) flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) {
        __builtin_amdgcn_s_sleep(2);
        if (lds_ld_asm(&c->giveup)) return;
        if (++spins > (1 << 21)) { give_up(c, err, 3); return; }
    }
}

// ---- ring image of one STEP of a row: the pieces a wave reads with one instruction each, lane-major ------------------------------------------
//   Q4_K (2304 B): qa[64][16] | qb[64][16] | hdr[16][16]                    Q6_K (3360 B): la[64][16] | lb[64][16] | qh[64][16] | scales[16][16] | d[16]
//   Q5_K (2816 B): the step's 16 native 176-byte blocks (two 32-weight units per lane and step)
template <int TYPE> struct ST;
template <> struct ST<PM_Q4_K> { static constexpr int BYTES = 2304, NDMA = 3; };
template <> struct ST<PM_Q6_K> { static constexpr int BYTES = 3360, NDMA = 5; };
template <> struct ST<PM_Q5_K> { static constexpr int BYTES = 2816, NDMA = 3; };

__device__ __forceinline__ void dma16(const uint8_t * src, char * dst /*wave-uniform*/) {
    __builtin_amdgcn_global_load_lds((const PM_G void *) src, (lds_vp) dst, 16, 0, 2);        // aux 2 = nt: streamed once (nt-weights)
}
__device__ __forceinline__ void dma4(const uint8_t * src, char * dst) {
    __builtin_amdgcn_global_load_lds((const PM_G void *) src, (lds_vp) dst, 4, 0, 2);
}
// the same with the piece's place inside the step as the INSTRUCTION offset (it is added to the global and to the LDS address alike: the source
// pointer is biased by -OFF): every piece of a step then shares one M0 value - one s_mov m0 per step instead of a readfirstlane + s_mov + s_nop in
// front of every DMA instruction of the loader's one-wave instruction stream
template <int OFF> __device__ __forceinline__ void dma16o(const uint8_t * row /*uniform*/, uint32_t voff /*>= OFF*/, char * dst) {
    __builtin_amdgcn_global_load_lds((const PM_G void *) (row + (voff - (uint32_t) OFF)), (lds_vp) dst, 16, OFF, 2);
}
template <int OFF> __device__ __forceinline__ void dma4o(const uint8_t * row, uint32_t voff, char * dst) {
    __builtin_amdgcn_global_load_lds((const PM_G void *) (row + (voff - (uint32_t) OFF)), (lds_vp) dst, 4, OFF, 2);
}
// gather step `c` of `row` into the ring at dst (unit / block indices clamped into the row: a tail step repeats the last unit, whose products the
// consumer zeroes exactly like the mat-vec's clamped loads). FAST (K >= 4096: every stream starts at least its piece offset into the row; Q5_K: whole
// steps only): instruction-offset form.
template <int TYPE, bool FAST>
__device__ __forceinline__ void dma_step(char * dst, const uint8_t * row, int K, int U, int c, int lane) {
    const uint32_t nb = (uint32_t) K / 256;
    if (TYPE == PM_Q4_K) {
        const uint32_t u = (uint32_t) min(64 * c + lane, U - 1), hb = (uint32_t) min(16 * c + lane, (int) nb - 1);
        if (FAST) {
            dma16o<0>(row, u * 16u, dst);
            dma16o<1024>(row, nb * 64 + u * 16u, dst);
            if (lane < 16) dma16o<2048>(row, nb * 128 + hb * 16u, dst);
        } else {
            dma16(row + u * 16u, dst);
            dma16(row + nb * 64 + u * 16u, dst + 1024);
            if (lane < 16) dma16(row + nb * 128 + hb * 16u, dst + 2048);
        }
    } else if (TYPE == PM_Q6_K) {
        static_assert(PM_Q6K_SCD == 0, "the engine's d gather assumes the contiguous d[nb] stream");
        const uint32_t u = (uint32_t) min(64 * c + lane, U - 1), sb = (uint32_t) min(16 * c + lane, (int) nb - 1);
        const uint32_t db = (uint32_t) min(16 * c + 2 * lane, ((int) nb - 1) & ~1);      // (two d per lane: an even block index keeps the 4-byte source aligned)
        if (FAST) {
            dma16o<0>(row, u * 16u, dst);
            dma16o<1024>(row, nb * 64 + u * 16u, dst);
            dma16o<2048>(row, nb * 128 + u * 16u, dst);
            if (lane < 16) dma16o<3072>(row, pm_q6k_sc_off(nb, sb), dst);
            if (lane < 8) dma4o<3328>(row, pm_q6k_d_off(nb, db), dst);
        } else {
            dma16(row + u * 16u, dst);
            dma16(row + nb * 64 + u * 16u, dst + 1024);
            dma16(row + nb * 128 + u * 16u, dst + 2048);
            if (lane < 16) dma16(row + pm_q6k_sc_off(nb, sb), dst + 3072);
            if (lane < 8) dma4(row + pm_q6k_d_off(nb, db), dst + 3328);
        }
    } else {                                               // Q5_K: 16 native blocks = 2816 contiguous bytes
        const uint32_t o = 16u * (uint32_t) c * PM_BS_Q5_K + (uint32_t) lane * 16u;
        if (FAST) {
            dma16o<0>(row, o, dst);
            dma16o<1024>(row, o + 1024u, dst);
            if (lane < 48) dma16o<2048>(row, o + 2048u, dst);
        } else {
            const uint32_t lim = nb * PM_BS_Q5_K - 16u;
            dma16(row + min(o, lim), dst);
            dma16(row + min(o + 1024u, lim), dst + 1024);
            if (lane < 48) dma16(row + min(o + 2048u, lim), dst + 2048);
        }
    }
}
// the step's weight registers of this lane, from the ring image
template <int TYPE> struct LW;
// (ulast = last valid unit of the row relative to the step's first: Q4_K / Q6_K images hold clamped copies per lane already, the Q5_K image is a
//  plain copy of the row's bytes - its reader clamps, or the f16 scales of a tail step would be whatever follows the row)
template <> struct LW<PM_Q4_K> {
    static __device__ __forceinline__ void get(typename QT<PM_Q4_K>::Wr (&w)[1], const char * b, int lane, int) {
        w[0].q0 = *(const u32x4 *) (b + lane * 16); w[0].q1 = *(const u32x4 *) (b + 1024 + lane * 16); w[0].h = *(const u32x4 *) (b + 2048 + (lane >> 2) * 16);
    }
};
template <> struct LW<PM_Q6_K> {
    static __device__ __forceinline__ void get(typename QT<PM_Q6_K>::Wr (&w)[1], const char * b, int lane, int) {
        w[0].l0 = *(const u32x4 *) (b + lane * 16); w[0].l1 = *(const u32x4 *) (b + 1024 + lane * 16); w[0].h = *(const u32x4 *) (b + 2048 + lane * 16);
        w[0].s = *(const u32x2 *) (b + 3072 + (lane >> 2) * 16 + 8 * ((lane >> 1) & 1));
        w[0].d = *(const uint16_t *) (b + 3328 + (lane >> 2) * 2);
    }
};
template <> struct LW<PM_Q5_K> {
    static __device__ __forceinline__ void get(typename QT<PM_Q5_K>::Wr (&w)[2], const char * b, int lane, int ulast) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ul = min(lane + 64 * i, ulast), hb = (ul >> 3) * PM_BS_Q5_K;
            w[i].q = *(const u32x4 *) (b + hb + 48 + 16 * (ul & 7)); w[i].qh = *(const u32x4 *) (b + hb + 16 + 16 * (ul & 1)); w[i].h = *(const u32x4 *) (b + hb);
        }
    }
};

// the prefetch of one wave for one mat-vec phase: the first steps of its own items of job 0 (rows wave, wave + 16, ... of the workgroup's slice), whole rows
// first, then the leading steps of one more row, as many as its slot holds
struct PfGeo { int r0, r1, cpr, rows_full, steps_part, slot_bytes; };
__device__ __forceinline__ int type_cpr(int type, int U) { const int upl = (U + 63) >> 6, ch = (type == PM_Q4_K || type == PM_Q6_K) ? PM_CH64 : PM_CH32; return (upl + ch - 1) / ch; }
__device__ __forceinline__ int type_step_bytes(int type) { return type == PM_Q4_K ? ST<PM_Q4_K>::BYTES : type == PM_Q6_K ? ST<PM_Q6_K>::BYTES : ST<PM_Q5_K>::BYTES; }
__device__ __forceinline__ int acts_bytes(int K) { return (((K + 15) & ~15) + (K / 16) * 4 + (((K / 256) + 3) & ~3) * 4 + 255) & ~255; }
__device__ __forceinline__ PfGeo pf_geo(const GemvJob & jb, int type, int pair, int K, int b, int G, int wave) {
    PfGeo g;
    g.r0 = (int) ((long) jb.N * b / G); g.r1 = (int) ((long) jb.N * (b + 1) / G);
    g.cpr = type_cpr(type, jb.U);
    const int sb = type_step_bytes(type) * (pair ? 2 : 1);
    int slot = ((ENG_OUT_OFF - acts_bytes(K)) / ENG_NW) & ~255;
    g.slot_bytes = slot < ENG_SLOT_MAX ? slot : ENG_SLOT_MAX;
    const int own = g.r1 - g.r0 > wave ? (g.r1 - g.r0 - wave + ENG_NW - 1) / ENG_NW : 0;      // rows of this wave
    int steps = g.slot_bytes / sb;                          // steps the slot holds
    if (jb.split) steps = 0;                                // (a few long rows spread over all waves as (row, chunk) items: not prefetched)
    g.rows_full = steps / g.cpr; g.steps_part = steps - g.rows_full * g.cpr;
    if (g.rows_full >= own) { g.rows_full = own; g.steps_part = 0; }
    return g;
}
template <int TYPE, bool PAIR, bool FAST>
__device__ __forceinline__ void pf_issue(char * slot, const GemvJob & jb, int K, const PfGeo & g, int wave, int lane) {
    constexpr int SB = ST<TYPE>::BYTES;
    char * dst = slot;
    for (int i = 0; i <= g.rows_full; ++i) {
        const int ns = i < g.rows_full ? g.cpr : g.steps_part;
        if (ns == 0) break;
        const int lrow = g.r0 + wave + ENG_NW * i;
        const uint8_t * row = jb.W + (long) job_row(jb, lrow) * jb.row_stride;
        for (int s_ = 0; s_ < ns; ++s_) {
            dma_step<TYPE, FAST>(dst, row, K, jb.U, s_, lane); dst += SB;
            if (PAIR) { dma_step<TYPE, FAST>(dst, jb.W2 + (long) lrow * jb.row_stride, K, jb.U, s_, lane); dst += SB; }
        }
    }
}
// issue the prefetch of mat-vec phase `ph` (its job 0) for this wave
__device__ __forceinline__ void prefetch_phase(const EngPhase * ph, char * smem, int b, int G, int wave, int lane) {
    const int K = ph->g.K, pair = ph->pair, ta = ph->ta;
    const GemvJob & jb = ph->g.job[0];
    const PfGeo g = pf_geo(jb, ta, pair, K, b, G, wave);
    char * slot = smem + ENG_OUT_OFF - (wave + 1) * g.slot_bytes;
    const bool fast = K >= 4096;                            // (instruction-offset DMA form: dma_step)
    if (ta == PM_Q4_K) {
        if (pair) { if (fast) pf_issue<PM_Q4_K, true, true>(slot, jb, K, g, wave, lane); else pf_issue<PM_Q4_K, true, false>(slot, jb, K, g, wave, lane); }
        else      { if (fast) pf_issue<PM_Q4_K, false, true>(slot, jb, K, g, wave, lane); else pf_issue<PM_Q4_K, false, false>(slot, jb, K, g, wave, lane); }
    } else if (ta == PM_Q6_K) {
        if (fast) pf_issue<PM_Q6_K, false, true>(slot, jb, K, g, wave, lane); else pf_issue<PM_Q6_K, false, false>(slot, jb, K, g, wave, lane);
    }
}

// `steps` prefetched steps of one row (both matrices of a pair: [m0 s0 | m1 s0 | m0 s1 | ...]) against the LDS activation row: per-lane partial sums
template <int TYPE, bool PAIR>
__device__ __forceinline__ void eat_steps(const char * img, int U, const XLds & xs, int steps, int lane, float (&acc)[PAIR ? 2 : 1]) {
    typedef QT<TYPE> T;
    constexpr int CH = T::NV == 64 ? PM_CH64 : PM_CH32, NM = PAIR ? 2 : 1;
    for (int s = 0; s < steps; ++s) {
        typename T::Wr w[NM][CH];
#pragma unroll
        for (int m = 0; m < NM; ++m) LW<TYPE>::get(w[m], img + (s * NM + m) * ST<TYPE>::BYTES, lane, U - 1 - 64 * CH * s);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int uu = lane + 64 * (s * CH + i);
            const bool uv = uu < U;
            const int u = min(uu, U - 1);
            typename T::X x;
            load_x_lds<TYPE>(x, xs, u, 0);
            x.yd = uv ? x.yd : 0.0f;                       // a clamped (out-of-row) unit contributes exactly 0
#pragma unroll
            for (int m = 0; m < NM; ++m) { int isum, msum; acc[m] = T::consume(w[m][i], x, u, acc[m], isum, msum); }
        }
    }
}

// the activation row of a mat-vec phase -> Q8_K in LDS (quantize_row_q8_K_ref bits: q8k_rows_to_lds), after an optional rms_norm whose sum of
// squares comes from the producing phase's partials. 16 waves, 4 blocks per wave and pass, <= 2 passes (<= 1 with norm weights).
__device__ __forceinline__ void eng_prologue(const GemvP & p, int8_t * xs_q, int * xs_gs, float * xs_d, int wave, int lane) {
    const int K = p.K, nblk = K / 256, r = lane >> 4, j = lane & 15;
    const __amdgpu_buffer_rsrc_t rx = coh_rsrc(p.xf);
    const bool norm = p.xmode == 3;
    double ssp[4] = {0.0, 0.0, 0.0, 0.0};
    if (norm) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (lane + 64 * i < p.n_ss) ssp[i] = __hip_atomic_load((const PM_G double *) (p.ss_in + lane + 64 * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float4 f[2][4], g[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) if (4 * ENG_NW * t < nblk) {
        const int B = min(4 * (wave + ENG_NW * t) + r, nblk - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) f[t][k] = coh_ld16(rx, (uint32_t) (B * 64 + 16 * k + j) * 16u);
    }
    if (norm) {
        const int B = min(4 * wave + r, nblk - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = ld_g((const float4 *) p.norm_w + (B * 64 + 16 * k + j));
    }
    float scale = 1.0f;
    if (norm) {
        const double tot = wave_sum_f64((ssp[0] + ssp[1]) + (ssp[2] + ssp[3]));
        const float mean = (float) (tot / K);
        scale = 1.0f / sqrtf(mean + p.eps);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) if (4 * (wave + ENG_NW * t) < nblk) {
        float v[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k][0] = f[t][k].x; v[k][1] = f[t][k].y; v[k][2] = f[t][k].z; v[k][3] = f[t][k].w;
            if (norm) { v[k][0] = v[k][0] * scale * g[k].x; v[k][1] = v[k][1] * scale * g[k].y; v[k][2] = v[k][2] * scale * g[k].z; v[k][3] = v[k][3] * scale * g[k].w; }
        }
        const int B = 4 * (wave + ENG_NW * t) + r;
        q8k_rows_to_lds(v, j, B < nblk, xs_q, xs_gs, xs_d, B);
    }
}

// attention of head h over cached cells by waves 0-3 (256 threads): attn_tail_device.h
template <int DH>
__device__ __forceinline__ void eng_attention(const EngPhase * ph, int h, char * smem, Ctl * c, unsigned & agen, int * err) {
    const AttnTailP a = {ph->aq, ph->akc, ph->avc, ph->apos, ph->aseq, ph->aseq_stride, ph->aout, ph->aH, ph->aHkv, ph->an_ctx, ph->ascale};
    const int lane = (int) __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    attn_tail_head<DH>(a, h, smem, [&]() __attribute__((always_inline)) { wbar(&c->abar, agen, 4, lane, c, err, 6); });
}

// The rows of one mat-vec phase in one workgroup: this wave's prefetched steps out of its LDS slot, everything else HBM -> registers.
template <int TA, int TB, bool PAIR, bool EPI>
__device__ __forceinline__ void phase_rows(const EngPhase * ph, char * smem, const XLds & xs, float * outbuf, int wave, int lane, int b, int G) {
    typedef Item<TA, PAIR, 1, EPI> IA;                        // (wq | wk | wv phases carry the NEOX row mapping: runtime identity for NORM rope)
    typedef Item<TB, PAIR, 1, EPI> IB;
    constexpr int R = IA::R, NM = PAIR ? 2 : 1;
    static_assert(R == 1, "one row per item");
    constexpr int NPRE = (PAIR || TA == PM_Q5_K) ? 1 : 2;   // register sets in flight before the prefetched steps are consumed (as the launches: mmvq.hip)
    GemvP pl = GemvP();                                    // what the row loops read of the argument block
    pl.K = ph->g.K;
    const GemvJob j0 = ph->g.job[0];
    const PfGeo g = pf_geo(j0, TA, PAIR, pl.K, b, G, wave);
    const int rows0 = g.r1 - g.r0;
    // ---- this wave's first register-path steps of job 0 go out now: they travel while the prefetched steps are consumed
    typename IA::Regs ga, gb;
    const int first = wave + PM_GEMV_NW * g.rows_full;     // the item the register path starts with (at chunk steps_part)
    if (!j0.split) {
        int prow = g.r0 + first * R, pc = g.steps_part;
        auto adv = [&]() __attribute__((always_inline)) { if (++pc == g.cpr) { pc = 0; prow += PM_GEMV_NW * R; } };
        IA::issue(ga, pl, j0, prow, g.r1, pc * IA::CH, lane); adv();
        if (NPRE >= 2) { IA::issue(gb, pl, j0, prow, g.r1, pc * IA::CH, lane); adv(); }
    }
    // ---- prefetched steps (the slot is this wave's own: its DMA loads are older than everything this wave has waited for since the prologue)
    float acc0[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) acc0[m] = 0.0f;
    {
        const char * img = smem + ENG_OUT_OFF - (wave + 1) * g.slot_bytes;
        for (int i = 0; i < g.rows_full; ++i) {
            float acc[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m] = 0.0f;
            eat_steps<TA, PAIR>(img, j0.U, xs, g.cpr, lane, acc);
            img += g.cpr * NM * ST<TA>::BYTES;
            float o[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) o[m] = wave_sum(acc[m]);
            if (lane == 0) outbuf[wave + PM_GEMV_NW * i] = PAIR ? silu_f(o[0]) * o[NM - 1] : o[0];
        }
        if (g.steps_part > 0) eat_steps<TA, PAIR>(img, j0.U, xs, g.steps_part, lane, acc0);
    }
    // ---- the rest of job 0, then jobs 1 and 2 (wk / wv next to wq): the mat-vec kernel's row loops
    if (!j0.split) IA::template run_job<false, NPRE>(ga, gb, pl, j0, xs, outbuf, first, rows0, g.r0, g.r1, lane, g.steps_part, g.steps_part > 0 ? acc0 : nullptr);
    else if constexpr (!PAIR) IA::template run_job_split<false>(ga, gb, pl, j0, xs, outbuf, wave, rows0 * g.cpr, g.r0, g.r1, lane);
    if constexpr (EPI) {
        const GemvJob j1 = ph->g.job[1], j2 = ph->g.job[2];
        const int r0_1 = (int) ((long) j1.N * b / G), r1_1 = (int) ((long) j1.N * (b + 1) / G), r0_2 = (int) ((long) j2.N * b / G), r1_2 = (int) ((long) j2.N * (b + 1) / G);
        const int cpr_1 = j1.split ? type_cpr(j1.is_b ? TB : TA, j1.U) : 1, cpr_2 = j2.split ? type_cpr(j2.is_b ? TB : TA, j2.U) : 1;
        const int ni_0 = j0.split ? rows0 * g.cpr : rows0, ni_1 = (r1_1 - r0_1) * cpr_1, ni_2 = (r1_2 - r0_2) * cpr_2;
        const int ob_1 = ni_0, ob_2 = ob_1 + ni_1;
        const int w1 = (wave + PM_GEMV_NW - ni_0 % PM_GEMV_NW) % PM_GEMV_NW;
        const int w2 = (wave + 2 * PM_GEMV_NW - (ni_0 + ni_1) % PM_GEMV_NW) % PM_GEMV_NW;
        typename IB::Regs gB, gB1;
        if (ni_1 > 0) {
            if (j1.split) { if (TA != TB && j1.is_b) IB::template run_job_split<false>(gB, gB1, pl, j1, xs, outbuf + ob_1, w1, ni_1, r0_1, r1_1, lane);
                            else                      IA::template run_job_split<false>(ga, gb, pl, j1, xs, outbuf + ob_1, w1, ni_1, r0_1, r1_1, lane); }
            else          { if (TA != TB && j1.is_b) IB::template run_job<false, 0>(gB, gB1, pl, j1, xs, outbuf + ob_1, w1, ni_1, r0_1, r1_1, lane);
                            else                      IA::template run_job<false, 0>(ga, gb, pl, j1, xs, outbuf + ob_1, w1, ni_1, r0_1, r1_1, lane); }
        }
        if (ni_2 > 0) {
            if (j2.split) { if (TA != TB && j2.is_b) IB::template run_job_split<false>(gB, gB1, pl, j2, xs, outbuf + ob_2, w2, ni_2, r0_2, r1_2, lane);
                            else                      IA::template run_job_split<false>(ga, gb, pl, j2, xs, outbuf + ob_2, w2, ni_2, r0_2, r1_2, lane); }
            else          { if (TA != TB && j2.is_b) IB::template run_job<false, 0>(gB, gB1, pl, j2, xs, outbuf + ob_2, w2, ni_2, r0_2, r1_2, lane);
                            else                      IA::template run_job<false, 0>(ga, gb, pl, j2, xs, outbuf + ob_2, w2, ni_2, r0_2, r1_2, lane); }
        }
    }
}

__global__ __launch_bounds__(ENG_THREADS, 4) void decode_engine_kernel(EngArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ Ctl ctl;
    Ctl * c = &ctl;
    float * outbuf = (float *) (smem + ENG_OUT_OFF);
    const int tid0 = threadIdx.x, wave_k = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    if (tid0 < (int) (sizeof(Ctl) / 4)) ((unsigned *) c)[tid0] = 0;
    __syncthreads();
    if (tid0 == 0 && __hip_atomic_load((PM_G int *) A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) lds_st_asm(&c->giveup, 1);   // an earlier launch gave up: the counters are not trustworthy
    const int b = blockIdx.x, G = gridDim.x;
    const unsigned NGR = (G % 16 == 0 && G / 16 <= ENG_MAXG / 16) ? 16 : 1, GS = G / NGR;
    unsigned agen = 0;                                     // generation of the attention's four-wave barrier
    const EngPhase * phs = uniform_const_ptr(A.ph);
#ifdef ENG_DEBUG
    unsigned long long * tsd = (A.dbg && b == ENG_DEBUG_WG && tid0 == 0) ? (unsigned long long *) A.dbg : nullptr;
#define ENG_STAMP(k) do { if (tsd && pi < 32) tsd[8 * pi + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ENG_STAMP(k) do { } while (0)
#endif
    int pf_done = -1;                                      // the last mat-vec phase whose prefetch has been issued ...
    int pf_w0 = -1;                                        // ... and the one wave 0 still owes: the wave that polls the device-wide barrier waits on vmcnt for every poll -
                                                           // with a prefetch in flight its first poll would return only when that has landed, and the workgroup with it
    for (int pi = 0; pi < A.n_ph; ++pi) {
        const EngPhase * ph = phs + pi;
        // (the lane id is re-derived through an opaque asm in every phase: what the row loops, prologues and epilogues compute from it - lane offsets, row
        //  pointers - would otherwise be hoisted out of the phase loop and kept alive, i.e. spilled, across all instantiations of the phase body)
        int lane_o = (int) __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_o));
        const int lane = lane_o, wave = wave_k, tid = wave * 64 + lane;
        // ---- the launch's first phase has nobody to prefetch for it
        if (pi == 0 && ph->kind == 0) { prefetch_phase(ph, smem, b, G, wave, lane); pf_done = 0; }
        // ---- seam: every workgroup's outputs of the previous phase are in memory
        if (pi > 0) {
            if (wave == 0) {
                if (lane == 0) g_wait(A.ctr, (unsigned) pi, GS, c, A.err);
                __builtin_amdgcn_wave_barrier();
                if (pf_w0 >= 0) { prefetch_phase(phs + pf_w0, smem, b, G, wave, lane); pf_w0 = -1; }
            }
            wg_bar();
        }
        ENG_STAMP(0);
        if (ph->kind == 1) {
            if (b < ph->aH && wave < 4) {
                if (ph->adh == 128) eng_attention<128>(ph, b, smem, c, agen, A.err);
                else eng_attention<64>(ph, b, smem, c, agen, A.err);
            }
        } else {
            const GemvP & p = ph->g;
            const int ta = ph->ta, tb = ph->tb, pair = ph->pair;
            const int K = p.K;
            int8_t * xs_q = (int8_t *) smem; int * xs_gs = (int *) (smem + ((K + 15) & ~15)); float * xs_d = (float *) (xs_gs + K / 16);
            eng_prologue(p, xs_q, xs_gs, xs_d, wave, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this wave's prefetch has landed too: it is older than the activation loads)
            wg_bar();
            ENG_STAMP(1);
            const XLds xs = {xs_q, xs_gs, xs_d, 0};
#define ENG_PH(TA_, TB_, P_, E_) phase_rows<TA_, TB_, P_, E_>(ph, smem, xs, outbuf, wave, lane, b, G)
            if (ph->epi) {
                if (tb == PM_Q4_K) ENG_PH(PM_Q4_K, PM_Q4_K, false, true);
                else if (tb == PM_Q6_K) ENG_PH(PM_Q4_K, PM_Q6_K, false, true);
                else ENG_PH(PM_Q4_K, PM_Q5_K, false, true);
            } else if (pair) ENG_PH(PM_Q4_K, PM_Q4_K, true, false);
            else if (ta == PM_Q4_K) ENG_PH(PM_Q4_K, PM_Q4_K, false, false);
            else ENG_PH(PM_Q6_K, PM_Q6_K, false, false);
#undef ENG_PH
            ENG_STAMP(2);
            // geometry of the jobs' results in outbuf (as phase_rows parks them)
            const int r0_0 = (int) ((long) p.job[0].N * b / G), r1_0 = (int) ((long) p.job[0].N * (b + 1) / G);
            const int r0_1 = (int) ((long) p.job[1].N * b / G), r1_1 = (int) ((long) p.job[1].N * (b + 1) / G);
            const int r0_2 = (int) ((long) p.job[2].N * b / G), r1_2 = (int) ((long) p.job[2].N * (b + 1) / G);
            const int cpr_0 = p.job[0].split ? type_cpr(ta, p.job[0].U) : 1;
            const int cpr_1 = p.job[1].split ? type_cpr(p.job[1].is_b ? tb : ta, p.job[1].U) : 1, cpr_2 = p.job[2].split ? type_cpr(p.job[2].is_b ? tb : ta, p.job[2].U) : 1;
            const int ob_1 = (r1_0 - r0_0) * cpr_0, ob_2 = ob_1 + (r1_1 - r0_1) * cpr_1;
            float ec0 = 1.0f, es0 = 0.0f, ec1 = 1.0f, es1 = 0.0f, ec2 = 1.0f, es2 = 0.0f;
            int epi_slot = 0; long epi_off = 0;
            if (ph->epi) {
                const int seq = p.epi.seq_ptr ? uniform_const_ptr(p.epi.seq_ptr)[0] : 0;
                epi_slot = uniform_const_ptr(p.epi.pos_ptr)[seq];
                epi_off = (long) seq * p.epi.seq_stride;
                qkv_cs(p.job[0], p.epi, r0_0, r1_0, tid, ec0, es0);
                qkv_cs(p.job[1], p.epi, r0_1, r1_1, tid, ec1, es1);
                qkv_cs(p.job[2], p.epi, r0_2, r1_2, tid, ec2, es2);
            }
            wg_bar();                                          // every row result of the workgroup is parked
            ENG_STAMP(3);
            // ---- epilogue: coalesced, write-through
            if (ph->epi) {
                write_out_qkv<true>(p.job[0], p.epi, outbuf, r0_0, r1_0, 0, tid, cpr_0, epi_slot, epi_off, ec0, es0);
                write_out_qkv<true>(p.job[1], p.epi, outbuf, r0_1, r1_1, ob_1, tid, cpr_1, epi_slot, epi_off, ec1, es1);
                write_out_qkv<true>(p.job[2], p.epi, outbuf, r0_2, r1_2, ob_2, tid, cpr_2, epi_slot, epi_off, ec2, es2);
            } else {
                const double ss = write_out<true, 1, true>(p.job[0], outbuf, r0_0, r1_0, 0, tid, 0, cpr_0);
                if (p.ss_out && wave == 0) {               // (rows <= 64 per workgroup: checked by the host)
                    const double ws = wave_sum_f64(ss);
                    if (lane == 0) __hip_atomic_store((PM_G double *) (p.ss_out + b), ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // ---- publish: this wave's stores have left; then, before the wave goes to wait, the first steps of its rows of the next mat-vec phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ENG_STAMP(4);
        {
            int m = pi + 1;
            while (m < A.n_ph && phs[m].kind != 0) ++m;
            if (m < A.n_ph && m > pf_done) {
                if (wave != 0) prefetch_phase(phs + m, smem, b, G, wave, lane); else pf_w0 = m;
                pf_done = m;
            }
        }
        wg_bar();
        ENG_STAMP(5);
        if (wave == 0 && lane == 0) g_arrive(A.ctr, (unsigned) pi, NGR, GS, pi == A.n_ph - 1);
        ENG_STAMP(6);
    }
}

} // namespace

namespace {

// f64 sum of the f32-rounded squares of a row (the rms_norm input of the launch's FIRST phase has no producing phase: embedding row or ring hand-off)
__global__ __launch_bounds__(256) void sumsq_row_kernel(const float * x, int K, double * out) {
    __shared__ double red[4];
    double ss = 0.0;
    for (int i = threadIdx.x; i < K; i += 256) { const float v = x[i]; const float sq = v * v; ss += (double) sq; }
    ss = wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

} // namespace

void pm_launch_sumsq_row(const float * x, int K, double * out, hipStream_t st) { hipLaunchKernelGGL(sumsq_row_kernel, dim3(1), dim3(256), 0, st, x, K, out); }

// ---- host side -----------------------------------------------------------------------------------------------------------------------------------
struct pm_eng_plan {
    std::vector<EngPhase> ph;
    EngPhase * d_ph = nullptr; unsigned * d_ctr = nullptr; int * d_err = nullptr; float * d_dbg = nullptr;
    int grid = 0; bool finished = false;
};

pm_eng_plan * pm_eng_plan_new() { return new pm_eng_plan(); }
void pm_eng_plan_free(pm_eng_plan * p) {
    if (!p) return;
    if (p->d_ph) (void) hipFree(p->d_ph);
    if (p->d_ctr) (void) hipFree(p->d_ctr);
    delete p;
}
int pm_eng_plan_phases(const pm_eng_plan * p) { return p ? (int) p->ph.size() : 0; }

int pm_eng_plan_add_matvec(pm_eng_plan * pl, const pm_gemv_fused & f) {
    if (!pl || pl->finished) return -1;
    EngPhase e = {};
    int ta, tb, grid; bool pair; size_t lds;
    const int rc = gemv_fill(f, 0, e.g, ta, tb, pair, lds, grid);
    if (rc) return rc < 0 ? rc : -rc;
    // the row loops compiled into the engine kernel (phase_rows<>): wq | wk | wv with the rope / KV-store epilogue in the type mixtures of the Q4_K_M and
    // Q6_K files, single Q4_K / Q6_K matrices (wo, ffn_down), Q4_K / Q6_K pairs (ffn_gate | ffn_up). (Q8_0 weights take Q8_0 activations: not yet)
    // (all-Q6_K wq | wk | wv and Q6_K pairs - the Q6_K file type - are written (-DENG_ALL_TYPES) but not compiled in: eight instantiations of the phase body
    //  in one kernel spill; the Q6_K models carry Q8_0 ffn_down rows anyway and stay on the launches)
    if (f.epi) { if (!(ta == PM_Q4_K && (tb == PM_Q4_K || tb == PM_Q6_K || tb == PM_Q5_K))) return -10; }
    else if (f.njobs != 1 || ta != tb || (ta != PM_Q4_K && ta != PM_Q6_K) || (pair && ta != PM_Q4_K)) return -10;
    if (e.g.xmode == 0 || e.g.xmode == 2) return -11;                             // f32 rows only; rms_norm only from producer-side partials
    if (f.dbg_int) return -11;
    const int nblk = f.K / 256;
    if (f.K % 256 || nblk > (e.g.xmode == 3 ? 4 * ENG_NW : 8 * ENG_NW)) return -12;
    if ((size_t) ((f.K + 15) & ~15) + (size_t) (f.K / 16) * 4 + (size_t) ((nblk + 3) & ~3) * 4 + 256 + (size_t) ENG_NW * 2304 > (size_t) ENG_OUT_OFF) return -12;
    if (pl->grid && pl->grid != grid) return -13;
    if (grid > ENG_MAXG) return -13;
    pl->grid = grid;
    // results a workgroup parks before its epilogue
    int nres = 0;
    for (int j = 0; j < 3; ++j) {
        GemvJob & jb = e.g.job[j];
        if (jb.N <= 0) continue;
        const int type = jb.is_b ? tb : ta;
        const int ch = (type == PM_Q4_K || type == PM_Q6_K) ? PM_CH64 : PM_CH32, cpr = (((jb.U + 63) >> 6) + ch - 1) / ch;
        if (jb.split && pair) return -14;
        const int rows = (jb.N + grid - 1) / grid + 1;
        nres += rows * (jb.split ? cpr : 1);
        if (f.ss_out && rows > 64) return -15;
    }
    if (nres > ENG_OUTF) return -15;
    e.kind = 0; e.ta = ta; e.tb = tb; e.pair = pair ? 1 : 0; e.epi = f.epi ? 1 : 0;
    pl->ph.push_back(e);
    return 0;
}

int pm_eng_plan_add_attention(pm_eng_plan * pl, const float * q, void * kc, void * vc, const int32_t * pos0, const int32_t * seq, long seq_stride, float * out,
                              int H, int Hkv, int dh, int n_ctx, float scale, int max_keys) {
    if (!pl || pl->finished) return -1;
    if ((dh != 64 && dh != 128) || n_ctx % 8 || !pos0 || H % Hkv) return -10;
    if (max_keys <= 0 || max_keys > n_ctx) max_keys = n_ctx;
    // LDS of the general path below the prefetch slots (the next wo's rows are landing there): part / pw / reductions (2 KiB + 64) | qs[dh] | part[256] | sc[max_keys + 8]
    if (attn_tail_lds(dh, max_keys) > (size_t) (ENG_OUT_OFF - ENG_NW * ENG_SLOT_MAX)) return -12;
    EngPhase e = {};
    e.kind = 1; e.aq = q; e.akc = (uint16_t *) kc; e.avc = (uint16_t *) vc; e.apos = pos0; e.aseq = seq; e.aseq_stride = seq_stride; e.aout = out;
    e.aH = H; e.aHkv = Hkv; e.adh = dh; e.an_ctx = n_ctx; e.amax_keys = max_keys; e.ascale = scale;
    pl->ph.push_back(e);
    return 0;
}

int pm_eng_plan_finish(pm_eng_plan * pl) {
    if (!pl || pl->ph.empty() || pl->grid <= 0) return -1;
    for (const EngPhase & e : pl->ph) if (e.kind == 1 && e.aH > pl->grid) return -13;
    hipDeviceProp_t pr; int dev = 0; (void) hipGetDevice(&dev);
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess || pl->grid > pr.multiProcessorCount) return -13;        // every workgroup must be resident
    if ((size_t) pr.sharedMemPerBlock < 64 * 1024) return -13;
    const size_t ctr_bytes = (size_t) 32 * 4 * (1 + 2 * (ENG_MAXG / 16)) + 64;
    if (hipMalloc((void **) &pl->d_ph, pl->ph.size() * sizeof(EngPhase)) != hipSuccess || hipMalloc((void **) &pl->d_ctr, ctr_bytes) != hipSuccess) return -3;
    if (hipMemcpy(pl->d_ph, pl->ph.data(), pl->ph.size() * sizeof(EngPhase), hipMemcpyHostToDevice) != hipSuccess) return -3;
    if (hipMemset(pl->d_ctr, 0, ctr_bytes) != hipSuccess) return -3;
    pl->d_err = (int *) ((char *) pl->d_ctr + ctr_bytes - 64);
    if (hipFuncSetAttribute((const void *) decode_engine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ENG_LDS) != hipSuccess) { (void) hipGetLastError(); return -3; }
#ifdef ENG_DEBUG
    if (!pl->d_dbg) { (void) hipMalloc((void **) &pl->d_dbg, 32 * 8 * 8); (void) hipMemset(pl->d_dbg, 0, 32 * 8 * 8); }
#endif
    pl->finished = true;
    return 0;
}

int pm_eng_plan_launch(pm_eng_plan * pl, hipStream_t st) {
    if (!pl || !pl->finished) return -1;
    EngArgs a = {pl->d_ph, (int) pl->ph.size(), pl->d_ctr, pl->d_err, pl->d_dbg};
    hipLaunchKernelGGL(decode_engine_kernel, dim3(pl->grid), dim3(ENG_THREADS), ENG_LDS, st, a);
    return 0;
}

int pm_eng_plan_status(pm_eng_plan * pl) {
    if (!pl || !pl->finished) return -1;
#ifdef ENG_DEBUG
    if (pl->d_dbg) {
        unsigned long long t[32 * 8];
        (void) hipMemcpy(t, pl->d_dbg, sizeof(t), hipMemcpyDeviceToHost);
        auto us = [&](unsigned long long v) { return v ? (double) (v - t[0]) / 100.0 : -1.0; };
        for (int ph = 0; ph < 32 && t[8 * ph]; ++ph)
            fprintf(stderr, "eng phase %2d (wg %d wave 0, us since the first seam-in): seam-in %.2f | prologue done %.2f | own rows done %.2f | all waves %.2f | stores out %.2f | barrier %.2f | arrived %.2f\n", ph, ENG_DEBUG_WG,
                    us(t[8 * ph]), us(t[8 * ph + 1]), us(t[8 * ph + 2]), us(t[8 * ph + 3]), us(t[8 * ph + 4]), us(t[8 * ph + 5]), us(t[8 * ph + 6]));
    }
#endif
    int err = 0;
    if (hipMemcpy(&err, pl->d_err, 4, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    if (err) {
        const size_t ctr_bytes = (size_t) 32 * 4 * (1 + 2 * (ENG_MAXG / 16)) + 64;
        (void) hipMemset(pl->d_ctr, 0, ctr_bytes);
    }
    return err;
}
