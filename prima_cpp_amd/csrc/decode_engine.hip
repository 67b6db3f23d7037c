// decode_engine.hip — the single-token layer stack as ONE persistent launch on a run-ahead LDS-DMA weight loader (round 5).
//
// What it replaces: the five launches per layer of engine.hip's run_layers_fused (mmvq.hip + attn_cached.hip), i.e. the same reference
// nodes - ggml_compute_forward_mul_mat over ggml_vec_dot_q4_K/q5_K/q6_K_q8_K with quantize_row_q8_K and rms_norm fused in front
// (ggml.c:12377, ggml-quants.c:7713 / 8281 / 8918 / 3785, ggml.c:11950), rope + F16 KV store (ggml.c:14143, src/llama.cpp:9673-9718),
// attention over cached cells (ggml.c:12445-12473, :13783-13879), silu * mul, residual adds - with the SAME device arithmetic (QT<>::consume,
// q8k_rows_to_lds, write_out, write_out_qkv, the attn_cached / attn_rope_body bodies): outputs are bit-identical to the five-launch path except
// for ffn_down, whose K = 28672 rows are summed chunk by chunk here (ring capacity) and lane by lane there (summation order, ~1e-7 relative).
//
// Why (profiles/r05_engine_seam_cost.txt, r05_seam_anatomy_sumsq.txt): a 70B layer is 75 us of weight streaming plus five dependent all-to-all
// seams. As kernel boundaries each seam costs ~6.2 us (boundary 2.5 + activation fetch + quantizing prologue + ramp and tail) during which HBM
// idles: 110 us per layer. Inside one launch a seam is a device-wide barrier + the same fetch / quantize (2-5 us), and a DEDICATED loader wave
// keeps streaming the NEXT phase's weights into an LDS ring while the consumers sit in the seam - weights do not depend on activations: the
// skeleton of this structure measured 92 us per layer on the box whose five launches take 110 (MI355X_MICROARCH.md "engine-vs-launches",
// "prefetch-credit").
//
// Structure: one 1024-thread workgroup per CU (all resident: the device-wide barrier needs that), wave 15 = loader, waves 0-14 = consumers.
//   * loader: walks every mat-vec phase of the launch in order; the rows of this workgroup are cut into ITEMS (a whole row of <= 2 steps, or one
//     step of a long / split row; a step = 64 units = what a wave consumes with one instruction stream); an item's bytes are gathered with
//     global_load_lds_dwordx4 ... nt into a ring-resident image [stream pieces of the step, lane-major] (the per-lane source addresses do the
//     row-SoA -> step-major permutation), at most 48 DMA instructions in flight; item offsets and a monotonic `landed` counter live in LDS.
//   * consumers: item n of the launch goes to wave n % 15; a wave waits for landed > n, reads the image with ds_read_b128 (conflict-free), runs the
//     mat-vec's own consume() against the LDS-resident Q8_K activation, parks the row result, retires the item (done[wave]).
//   * seam: results -> epilogue (bias / residual / silu*mul / RoPE + KV store, all write-through sc1 stores) -> consumer barrier -> two-level
//     device-wide arrival -> wait -> consumers fetch the next activation row with sc1 loads and quantize it (rms_norm from the producer-side
//     partial sums of squares: no reduction, no extra barrier).
// Every wait is bounded; a time-out raises the launch's watchdog word and every later wait of that workgroup returns at once.
#define PM_GEMV_BLOCK 960                  // the mat-vec row loops of mmvq_device.h run on the engine's 15 CONSUMER waves (rows are dealt wave, wave + 15, ...)
#include "mmvq_device.h"
#include "attn_device.h"
#include "pm355_engine.h"
#include <vector>
#include <stdio.h>

using namespace pmv;

namespace {

constexpr int ENG_NW = 16, ENG_NL = 1, ENG_NC = ENG_NW - ENG_NL, ENG_THREADS = ENG_NW * 64;      // waves: consumers 0 .. 14, loader 15
static_assert(ENG_NC == PM_GEMV_NW, "consumer waves == the row loops' wave count");
constexpr int ENG_HEAD = 96 * 1024;                        // bytes of a phase's FIRST rows that go through the ring (what the loader has in LDS when the seam ends)
constexpr int ENG_RING = 120 * 1024;                       // bytes of weight images in flight per CU
constexpr int ENG_ACT = 36864;                             // Q8_K activation row: q[K] | group sums[K/16] | d[K/256], K <= 28672 (also the attention scratch)
constexpr int ENG_OUTF = 640;                              // parked results per workgroup (floats)
constexpr int ENG_LDS = ENG_RING + ENG_ACT + ENG_OUTF * 4;
constexpr int ENG_VMAX = 48;                               // DMA instructions in flight (the counter has 6 bits)
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

// LDS control block. Accessed ONLY through LDS-typed pointers (a generic access is FLAT: it waits on vmcnt too and would drain the loader's queue);
// the loader's own accesses are inline asm (a compiler-visible LDS access after global_load_lds gets an s_waitcnt vmcnt(0) in front of it).
struct Ctl { unsigned landed[2], cbar, abar, giveup, pad_[3]; unsigned done[16]; unsigned item_off[64]; unsigned item_U[64]; };      // (item_U right behind item_off: one ds_write2st64 per item)

__device__ __forceinline__ lds_u32 * L(unsigned * p) { return (lds_u32 *) p; }
__device__ __forceinline__ unsigned lds_ld(unsigned * p) { return __hip_atomic_load(L(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned * p, unsigned v) { __hip_atomic_store(L(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st_asm(unsigned * p, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned) (uintptr_t) L(p)), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned lds_ld_asm(unsigned * p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned) (uintptr_t) L(p)) : "memory");
    return v;
}
__device__ __forceinline__ void give_up(Ctl * c, int * err, int code, bool asm_path) {
    if (asm_path) lds_st_asm(&c->giveup, 1); else lds_st(&c->giveup, 1);
    __hip_atomic_store((PM_G int *) err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consumers: wait until *word >= target (one poll per wave instruction; every lane reads the same address)
__device__ __forceinline__ bool spin_ge(unsigned * word, unsigned target, Ctl * c, int * err, int code) {
    int spins = 0;
    while ((int) (lds_ld(word) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (lds_ld(&c->giveup)) return false;
        if (++spins > (1 << 22)) { give_up(c, err, code, false); return false; }
    }
    return true;
}
// barrier of `n` waves on a monotonic LDS counter (gen-th use completes at gen * n arrivals)
__device__ __forceinline__ void wbar(unsigned * ctr, unsigned & gen, int n, int lane, Ctl * c, int * err, int code) {
    ++gen;
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): this wave's LDS writes are done
    if (lane == 0) {
        __hip_atomic_fetch_add(L(ctr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        spin_ge(ctr, gen * (unsigned) n, c, err, code);
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
        if (lds_ld(&c->giveup)) return;
        if (++spins > (1 << 21)) { give_up(c, err, 3, false); return; }
    }
}

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

// items of a job inside one workgroup, as both the loader and the consumers count them
struct JobGeo { int r0, r1, cpr, split, items, steps /*per item*/, bytes /*per item*/, ob /*first result slot*/, nres /*result slots*/; };
__device__ __forceinline__ int type_cpr(int type, int U) { const int upl = (U + 63) >> 6, ch = (type == PM_Q4_K || type == PM_Q6_K) ? PM_CH64 : PM_CH32; return (upl + ch - 1) / ch; }
__device__ __forceinline__ int type_step_bytes(int type) { return type == PM_Q4_K ? ST<PM_Q4_K>::BYTES : type == PM_Q6_K ? ST<PM_Q6_K>::BYTES : ST<PM_Q5_K>::BYTES; }
__device__ __forceinline__ JobGeo job_geo(const GemvJob & jb, int type, int pair, int b, int G, int ob) {
    JobGeo g;
    g.r0 = (int) ((long) jb.N * b / G); g.r1 = (int) ((long) jb.N * (b + 1) / G);
    g.cpr = type_cpr(type, jb.U);
    g.split = jb.split;
    g.items = g.split ? (g.r1 - g.r0) * g.cpr : (g.r1 - g.r0);
    g.steps = g.split ? 1 : g.cpr;
    g.bytes = g.steps * type_step_bytes(type) * (pair ? 2 : 1);
    g.ob = ob; g.nres = g.items;
    return g;
}

// How many of job 0's items take the ring (whole rows; the rest of the phase is read HBM -> registers by the consumers themselves, like the mat-vec
// kernel): measured, LDS-DMA instructions issue at ~150 cycles each into a CU whose consumers are busy - one KiB per 150 cycles is 14 GB/s per
// loader wave, two loader waves reached 20 of the 26.5 GB/s a CU's share of HBM is. The ring therefore carries only what hides the seam.
__device__ __forceinline__ int ring_items(const JobGeo & jg) {
    int n = ENG_HEAD / jg.bytes;
    if (jg.split) n -= n % jg.cpr;
    return n < jg.items ? n : jg.items;
}

// ---- loader wave ------------------------------------------------------------------------------------------------------------------------------
// One wave runs EVERYTHING the weights need - its instruction stream is the engine's bandwidth: a single wave issues one instruction every
// 5-8 cycles, a 2304-byte step has to leave every ~180 cycles, so an item may cost ~40 instructions besides its DMA. Hence: every cursor lives in
// SGPRs (no lane-indexed FIFOs, no per-item loops), the type / pair dispatch is per JOB, and `landed` follows from arithmetic - all items of a
// job carry the same number of DMA instructions, so after s_waitcnt vmcnt(48) everything but the job's newest ceil(48 / k) items has landed.
#ifdef ENG_DEBUG
__device__ unsigned long long g_ld_room = 0, g_ld_issue = 0, g_ld_wait = 0, g_cs_spin = 0, g_cs_eat = 0;      // (debug builds: cycle accounts of workgroup 0 - races between its waves are harmless noise: one wave each writes)
#endif
struct LoaderState {
    unsigned n, cur, U, tail, tailU;                       // items WALKED (both loaders place every item) | ring cursor | the same unwrapped | oldest unretired item (cached), its start
    unsigned m, landed;                                    // OWN items issued | own items published as landed
};
__device__ __forceinline__ void lds_st2_asm(unsigned * p, unsigned a, unsigned b) {       // p[0] = a, p[64] = b (item_off / item_U of one FIFO slot)
    asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:1" :: "v"((unsigned) (uintptr_t) L(p)), "v"(a), "v"(b) : "memory");
}
// the cursor of the next item (a tail of the ring too short for it is skipped: both loaders and nobody else follow this rule)
__device__ __forceinline__ void loader_place(LoaderState & S, unsigned bytes) {
    if (S.cur + bytes > (unsigned) ENG_RING) { S.U += (unsigned) ENG_RING - S.cur; S.cur = 0; }
}
// room for `bytes` at the cursor? (the live window [start of the oldest unretired item, end of this item) may not exceed the ring)
__device__ __forceinline__ bool loader_make_room(const EngArgs & A, Ctl * c, LoaderState & S, unsigned bytes, int lane, int LD) {
    if (S.U + bytes - S.tailU <= (unsigned) ENG_RING) return true;
    int spins = 0;
    for (;;) {
        // refresh the tail: wave w has retired done[w] of its items (w, w + NC, ...): the oldest unretired item of the launch is the minimum
        const unsigned v = lane < ENG_NC ? (unsigned) lane + lds_ld_asm(&c->done[lane < ENG_NC ? lane : 0]) * (unsigned) ENG_NC : 0xFFFFFFFFu;
        unsigned t = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < ENG_NC; ++i) t = min(t, (unsigned) __builtin_amdgcn_readlane((int) v, i));
        S.tail = min(t, S.n);
        S.tailU = S.tail >= S.n ? S.U : (unsigned) __builtin_amdgcn_readfirstlane((int) lds_ld_asm(&c->item_U[S.tail & 63]));
        if (S.U + bytes - S.tailU <= (unsigned) ENG_RING) return true;
        // ring full: nothing can be issued anyway -> drain, publish everything of this loader in flight, wait for a retirement
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (S.landed != S.m) { S.landed = S.m; if (lane == 0) lds_st_asm(&c->landed[LD], S.m); }
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_readfirstlane((int) lds_ld_asm(&c->giveup))) return false;     // (made scalar: a divergent exit would turn every cursor into a VGPR)
        if (++spins > (1 << 22)) { give_up(c, A.err, 1, true); return false; }
    }
}
template <int TYPE, bool PAIR, bool FAST>
__device__ __forceinline__ bool loader_job(const EngArgs & A, Ctl * c, char * ring, LoaderState & S, const GemvJob & jb, int K, int b, int G, int lane, int LD) {
    const JobGeo jg = job_geo(jb, TYPE, PAIR, b, G, 0);
    constexpr int SB = ST<TYPE>::BYTES;
    const unsigned k_item = (unsigned) (ST<TYPE>::NDMA * jg.steps * (PAIR ? 2 : 1));      // DMA instructions per item
    const unsigned lag = ((unsigned) ENG_VMAX + k_item - 1) / k_item;                    // OWN items the newest ENG_VMAX instructions may belong to
    const unsigned job_m0 = S.m;
    const int chunks = jg.split ? jg.cpr : 1, rows = ring_items(jg) / chunks;
    const uint8_t * row_lin = jb.W + (long) jg.r0 * jb.row_stride, * row2 = PAIR ? jb.W2 + (long) jg.r0 * jb.row_stride : nullptr;
    for (int r = 0; r < rows; ++r, row_lin += jb.row_stride, row2 += PAIR ? jb.row_stride : 0) {
        const uint8_t * row = jb.nx_s ? jb.W + (long) job_row(jb, jg.r0 + r) * jb.row_stride : row_lin;      // (NEOX rope: permuted rows)
        for (int cc = 0; cc < chunks; ++cc) {
            loader_place(S, (unsigned) jg.bytes);
            if ((int) (S.n % (unsigned) ENG_NL) == LD) {          // (with more than one loader wave the items alternate; every loader walks - places - every item)
#ifdef ENG_DEBUG
                const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
                if (!loader_make_room(A, c, S, (unsigned) jg.bytes, lane, LD)) return false;
#ifdef ENG_DEBUG
                const unsigned long long t_b = __builtin_amdgcn_s_memtime();
                if (blockIdx.x == 0 && LD == 0) g_ld_room += t_b - t_a;
#endif
                char * dst = ring + S.cur;
                for (int s_ = 0; s_ < jg.steps; ++s_) {
                    dma_step<TYPE, FAST>(dst + s_ * SB, row, K, jb.U, cc + s_, lane);
                    if (PAIR) dma_step<TYPE, FAST>(dst + (jg.steps + s_) * SB, row2, K, jb.U, cc + s_, lane);
                }
                if (lane == 0) lds_st2_asm(&c->item_off[S.n & 63], S.cur, S.U);
                ++S.m;
#ifdef ENG_DEBUG
                const unsigned long long t_c = __builtin_amdgcn_s_memtime();
                if (blockIdx.x == 0 && LD == 0) g_ld_issue += t_c - t_b;
#endif
                asm volatile("s_waitcnt vmcnt(48)" ::: "memory");                         // (ENG_VMAX)
#ifdef ENG_DEBUG
                if (blockIdx.x == 0 && LD == 0) g_ld_wait += __builtin_amdgcn_s_memtime() - t_c;
#endif
                const unsigned in_job = S.m - job_m0;
                const unsigned pub = in_job > lag ? S.m - lag : (in_job * k_item >= (unsigned) ENG_VMAX ? job_m0 : S.landed);
                if (pub != S.landed) { S.landed = pub; if (lane == 0) lds_st_asm(&c->landed[LD], pub); }
            } else if (lane == 0) lds_st2_asm(&c->item_off[S.n & 63], S.cur, S.U);       // (the same values its issuer writes: this wave may need the entry first)
            S.cur += (unsigned) jg.bytes; S.U += (unsigned) jg.bytes; ++S.n;
        }
    }
    return true;
}
__device__ __forceinline__ void loader_wave(const EngArgs & A, Ctl * c, char * ring, int lane, int LD) {
    const int b = blockIdx.x, G = gridDim.x;
    __builtin_amdgcn_s_setprio(3);                         // the youngest waves of their SIMDs would otherwise lose every arbitration to the consumers
    LoaderState S = {0, 0, 0, 0, 0, 0, 0};
#ifdef ENG_DEBUG
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const EngPhase * phs = uniform_const_ptr(A.ph);
    for (int pi = 0; pi < A.n_ph; ++pi) {
        const EngPhase * ph = phs + pi;
        if (ph->kind != 0) continue;
        const int K = ph->g.K, pair = ph->pair, ta = ph->ta, tb = ph->tb;
        for (int j = 0; j < 1; ++j) {                     // (job 0's first rows only: ring_items())
            if (ph->g.job[j].N <= 0) continue;
            const GemvJob & jb = ph->g.job[j];
            const int type = jb.is_b ? tb : ta;
            // the instruction-offset DMA form needs every stream of a row to start at least its piece's offset into the row (K >= 4096) and, for the
            // plain-copy Q5_K image, whole steps
            const bool fast = K >= 4096 && (type != PM_Q5_K || (K / 256) % 16 == 0);
            bool ok;
#define ENG_LJ(T_, P_, F_) loader_job<T_, P_, F_>(A, c, ring, S, jb, K, b, G, lane, LD)
            if (type == PM_Q4_K) ok = pair ? (fast ? ENG_LJ(PM_Q4_K, true, true) : ENG_LJ(PM_Q4_K, true, false)) : (fast ? ENG_LJ(PM_Q4_K, false, true) : ENG_LJ(PM_Q4_K, false, false));
            else if (type == PM_Q6_K) ok = pair ? (fast ? ENG_LJ(PM_Q6_K, true, true) : ENG_LJ(PM_Q6_K, true, false)) : (fast ? ENG_LJ(PM_Q6_K, false, true) : ENG_LJ(PM_Q6_K, false, false));
            else ok = fast ? ENG_LJ(PM_Q5_K, false, true) : ENG_LJ(PM_Q5_K, false, false);
#undef ENG_LJ
            if (!ok) return;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) lds_st_asm(&c->landed[LD], S.m);
#ifdef ENG_DEBUG
    if (A.dbg && b == 0 && lane == 0 && LD == 0) {
        float * Ld = A.dbg + 64 * 16 + 64 * 8 + 64 + 32 * 16; Ld[0] = (float) S.n; Ld[1] = (float) g_ld_room; Ld[2] = (float) g_ld_issue; Ld[3] = (float) (__builtin_amdgcn_s_memrealtime() - dbg_t0) / 100.0f; Ld[4] = (float) g_ld_wait;
        g_ld_room = g_ld_issue = g_ld_wait = 0;
    }
#endif
}

// ---- consumers ---------------------------------------------------------------------------------------------------------------------------------
// one item: `steps` steps of a row (both matrices of a pair), products against the LDS activation row; the wave's sum goes to out[slot]
#ifdef ENG_DEBUG
__device__ float * g_dbg_lane = nullptr;      // (debug builds: per-lane record of the item being consumed, set by the kernel for item 0 of workgroup 0)
#endif
template <int TYPE, bool PAIR>
__device__ __forceinline__ void eat_item(const char * img, int U, const XLds & xs, int c0, int steps, int lane, float * out_slot) {
    typedef QT<TYPE> T;
    constexpr int CH = T::NV == 64 ? PM_CH64 : PM_CH32, NM = PAIR ? 2 : 1;
    float acc[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m] = 0.0f;
    for (int s = 0; s < steps; ++s) {
        typename T::Wr w[NM][CH];
#pragma unroll
        for (int m = 0; m < NM; ++m) LW<TYPE>::get(w[m], img + (m * steps + s) * ST<TYPE>::BYTES, lane, U - 1 - 64 * CH * (c0 + s));
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int uu = lane + 64 * ((c0 + s) * CH + i);
            const bool uv = uu < U;
            const int u = min(uu, U - 1);
            typename T::X x;
            load_x_lds<TYPE>(x, xs, u, 0);
            x.yd = uv ? x.yd : 0.0f;                       // a clamped (out-of-row) unit contributes exactly 0
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                int isum, msum; acc[m] = T::consume(w[m][i], x, u, acc[m], isum, msum);
#ifdef ENG_DEBUG
                if (g_dbg_lane && s == 0 && i == 0 && m == 0) { float * o = g_dbg_lane + 8 * lane; o[0] = (float) isum; o[1] = (float) msum; o[2] = x.yd; o[3] = acc[m]; o[4] = (float) x.q[0]; o[5] = (float) x.gs[0]; o[6] = (float) u; o[7] = (float) uv; }
#endif
            }
        }
    }
    float o[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) o[m] = wave_sum(acc[m]);
    if (lane == 0) *out_slot = PAIR ? silu_f(o[0]) * o[NM - 1] : o[0];
}

// the activation row of a mat-vec phase -> Q8_K in LDS (quantize_row_q8_K_ref bits: q8k_rows_to_lds), after an optional rms_norm whose sum of
// squares comes from the producing phase's partials. 15 waves, 4 blocks per wave and pass, <= 2 passes (<= 1 with norm weights).
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
    for (int t = 0; t < 2; ++t) if (4 * ENG_NC * t < nblk) {
        const int B = min(4 * (wave + ENG_NC * t) + r, nblk - 1);
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
    for (int t = 0; t < 2; ++t) if (4 * (wave + ENG_NC * t) < nblk) {
        float v[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k][0] = f[t][k].x; v[k][1] = f[t][k].y; v[k][2] = f[t][k].z; v[k][3] = f[t][k].w;
            if (norm) { v[k][0] = v[k][0] * scale * g[k].x; v[k][1] = v[k][1] * scale * g[k].y; v[k][2] = v[k][2] * scale * g[k].z; v[k][3] = v[k][3] * scale * g[k].w; }
        }
        const int B = 4 * (wave + ENG_NC * t) + r;
        q8k_rows_to_lds(v, j, B < nblk, xs_q, xs_gs, xs_d, B);
    }
}

// attention of head h over cached cells by waves 0-3 (256 threads): attn_cached.hip's short path (<= 64 cells, one 4-wave barrier) and the
// cached form of attn_rope_body beyond, same arithmetic and rounding points; every load of q / K / V is an agent-scope load (q and this token's
// cell were written by other CUs in this launch)
template <int DH>
__device__ __forceinline__ void eng_attention(const EngPhase * ph, int h, char * smem, Ctl * c, unsigned & agen, int * err) {
    // (the thread id goes through an opaque asm: everything derived from it - dozens of LDS / cache offsets - is then recomputed per phase instead
    //  of being hoisted out of the phase loop and kept alive, i.e. spilled, across the mat-vec phases)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = ph->aH, Hkv = ph->aHkv, n_ctx = ph->an_ctx;
    const int hk = h / (H / Hkv);
    const float scale = ph->ascale;
    const int seq = ph->aseq ? uniform_const_ptr(ph->aseq)[0] : 0;
    const int n_kv = uniform_const_ptr(ph->apos)[seq] + 1;
    const long soff = (long) seq * ph->aseq_stride;
    const __amdgpu_buffer_rsrc_t rk = coh_rsrc(ph->akc + soff), rv = coh_rsrc(ph->avc + soff), rq = coh_rsrc(ph->aq);
    float * part = (float *) smem;                         // [4][64]
    float * pw = part + 256;                               // [4][64]
    float * redf = pw + 256;                               // [8]
    double * redd = (double *) (redf + 8);                 // [4]
    float * body = (float *) (redd + 4);                   // general path: qs[DH] | part[256] | sc[max_keys]
    auto bar = [&]() __attribute__((always_inline)) { wbar(&c->abar, agen, 4, lane, c, err, 6); };
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
        if (lane < DPW) st_act<true>(ph->aout + (long) h * DH + DPW * wave + lane, o);
        return;
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
        st_act<true>(ph->aout + (long) h * DH + tid, acc);
    }
}

// The rows of one mat-vec phase in one workgroup. Returns the number of ring items of the phase (launch-wide item counter).
template <int TA, int TB, bool PAIR, bool EPI>
__device__ __forceinline__ int phase_rows(const EngArgs & A, const EngPhase * ph, Ctl * c, char * ring, const XLds & xs, float * outbuf, unsigned nbase, int wave, int lane, int b, int G,
                                          JobGeo & g0, JobGeo & g1, JobGeo & g2) {
    typedef Item<TA, PAIR, 1> IA;
    typedef Item<TB, PAIR, 1> IB;
    constexpr int R = IA::R;
    static_assert(R == 1, "one row per item");
    constexpr int NPRE = (PAIR || TA == PM_Q5_K) ? 1 : 2;   // register sets in flight before the ring items are consumed (as the launches: mmvq.hip)
    GemvP pl = GemvP();                                    // what the row loops read of the argument block
    pl.K = ph->g.K;
    const GemvJob j0 = ph->g.job[0], j1 = ph->g.job[1], j2 = ph->g.job[2];
    g0 = job_geo(j0, TA, PAIR, b, G, 0);
    g1.ob = g0.nres; if (j1.N > 0) g1 = job_geo(j1, j1.is_b ? TB : TA, false, b, G, g0.nres);
    g2.ob = g1.ob + g1.nres; if (j2.N > 0) g2 = job_geo(j2, j2.is_b ? TB : TA, false, b, G, g1.ob + g1.nres);
    const int P = ring_items(g0), Prow = g0.split ? P / g0.cpr : P, r0r = g0.r0 + Prow;
    // ---- this wave's first register-path steps of job 0 go out now: they travel while the ring items are consumed
    typename IA::Regs ga, gb;
    if (!g0.split) {
        const int cpr0 = (((j0.U + 63) >> 6) + IA::CH - 1) / IA::CH;
        int prow = r0r + wave * R, pc = 0;
        auto adv = [&]() __attribute__((always_inline)) { if (++pc == cpr0) { pc = 0; prow += PM_GEMV_NW * R; } };
        IA::issue(ga, pl, j0, prow, g0.r1, pc * IA::CH, lane); adv();
        if (NPRE >= 2) { IA::issue(gb, pl, j0, prow, g0.r1, pc * IA::CH, lane); adv(); }
    }
    // ---- ring items (launch-wide index n -> wave n % 15), then the rest of job 0 HBM -> registers. An item that has not landed yet (the loader shares the
    // CU's address path with fifteen streaming waves) is left for a second pass behind the register rows: waiting for it would hold back this wave's share
    // of the stream.
    const int ni_0 = g0.split ? (g0.r1 - r0r) * g0.cpr : (g0.r1 - r0r + R - 1) / R;
    unsigned n = nbase + (unsigned) ((wave + ENG_NC - (int) (nbase % ENG_NC)) % ENG_NC);
    auto ring_pass = [&](bool wait) __attribute__((always_inline)) {
        unsigned k_done = lds_ld(&c->done[wave]);
        for (; n < nbase + (unsigned) P; n += ENG_NC) {
            const int id = (int) (n - nbase);
            const int c0 = g0.split ? id % g0.cpr : 0;
            if (!wait) { if ((int) (lds_ld(&c->landed[n % (unsigned) ENG_NL]) - (n / (unsigned) ENG_NL + 1)) < 0) break; }
            else spin_ge(&c->landed[n % (unsigned) ENG_NL], n / (unsigned) ENG_NL + 1, c, A.err, 4);
            const char * img = ring + lds_ld(&c->item_off[n & 63]);
            eat_item<TA, PAIR>(img, j0.U, xs, c0, g0.steps, lane, outbuf + id);
            ++k_done;
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the image has been read (and the result parked)
            if (lane == 0) lds_st(&c->done[wave], k_done);
        }
    };
    ring_pass(false);
    if (!g0.split) IA::template run_job<false, NPRE>(ga, gb, pl, j0, xs, outbuf + Prow, wave, ni_0, r0r, g0.r1, lane);
    else if constexpr (!PAIR) IA::template run_job_split<false>(ga, gb, pl, j0, xs, outbuf + P, wave, ni_0, r0r, g0.r1, lane);
    ring_pass(true);
    if constexpr (EPI) {
        const int w1 = (wave + PM_GEMV_NW - ni_0 % PM_GEMV_NW) % PM_GEMV_NW;
        const int w2 = (wave + 2 * PM_GEMV_NW - (ni_0 + g1.items) % PM_GEMV_NW) % PM_GEMV_NW;
        typename IB::Regs gB, gB1;
        if (g1.items > 0) {
            if (g1.split) { if (TA != TB && j1.is_b) IB::template run_job_split<false>(gB, gB1, pl, j1, xs, outbuf + g1.ob, w1, g1.items, g1.r0, g1.r1, lane);
                            else                      IA::template run_job_split<false>(ga, gb, pl, j1, xs, outbuf + g1.ob, w1, g1.items, g1.r0, g1.r1, lane); }
            else          { if (TA != TB && j1.is_b) IB::template run_job<false, 0>(gB, gB1, pl, j1, xs, outbuf + g1.ob, w1, g1.items, g1.r0, g1.r1, lane);
                            else                      IA::template run_job<false, 0>(ga, gb, pl, j1, xs, outbuf + g1.ob, w1, g1.items, g1.r0, g1.r1, lane); }
        }
        if (g2.items > 0) {
            if (g2.split) { if (TA != TB && j2.is_b) IB::template run_job_split<false>(gB, gB1, pl, j2, xs, outbuf + g2.ob, w2, g2.items, g2.r0, g2.r1, lane);
                            else                      IA::template run_job_split<false>(ga, gb, pl, j2, xs, outbuf + g2.ob, w2, g2.items, g2.r0, g2.r1, lane); }
            else          { if (TA != TB && j2.is_b) IB::template run_job<false, 0>(gB, gB1, pl, j2, xs, outbuf + g2.ob, w2, g2.items, g2.r0, g2.r1, lane);
                            else                      IA::template run_job<false, 0>(ga, gb, pl, j2, xs, outbuf + g2.ob, w2, g2.items, g2.r0, g2.r1, lane); }
        }
    }
    return P;
}

__global__ __launch_bounds__(ENG_THREADS, 4) void decode_engine_kernel(EngArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ Ctl ctl;
    Ctl * c = &ctl;
    char * ring = smem;
    char * acts = smem + ENG_RING;
    float * outbuf = (float *) (smem + ENG_RING + ENG_ACT);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < (int) (sizeof(Ctl) / 4)) ((unsigned *) c)[tid] = 0;
    __syncthreads();                                       // the only workgroup-wide barrier: before the roles split
    if (tid == 0 && __hip_atomic_load((PM_G int *) A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) lds_st(&c->giveup, 1);   // an earlier launch gave up: the counters are not trustworthy
#ifndef ENG_NO_LOADER
    if (wave >= ENG_NC) { loader_wave(A, c, ring, lane, wave - ENG_NC); return; }
#endif

    const int b = blockIdx.x, G = gridDim.x;
    const unsigned NGR = (G % 16 == 0 && G / 16 <= ENG_MAXG / 16) ? 16 : 1, GS = G / NGR;
    unsigned cgen = 0, agen = 0;                           // generations of the consumer / attention barriers
    unsigned nbase = 0;                                    // launch-wide index of the current phase's first item
    const EngPhase * phs = uniform_const_ptr(A.ph);
#ifdef ENG_DEBUG
    unsigned long long * tsd = (A.dbg && b == 0 && tid == 0) ? (unsigned long long *) (A.dbg + 64 * 16 + 64 * 8 + 64) : nullptr;
#define ENG_STAMP(k) do { if (tsd && pi < 32) tsd[8 * pi + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ENG_STAMP(k) do { } while (0)
#endif
    const int wave_k = wave;                               // (scalar: lives in an SGPR across the phases; the lane id is re-derived with mbcnt)
    for (int pi = 0; pi < A.n_ph; ++pi) {
        const EngPhase * ph = phs + pi;
        // (the thread id is re-derived through an opaque asm in every phase: what the row loops, prologues and epilogues compute from it - lane offsets, row
        //  pointers - would otherwise be hoisted out of the phase loop and kept alive, i.e. spilled, across all eight instantiations of the phase body)
        int lane_o = (int) __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_o));
        const int lane = lane_o, wave = wave_k, tid = wave * 64 + lane;
        // ---- seam: every workgroup's outputs of the previous phase are in memory
        if (pi > 0) {
            if (wave == 0 && lane == 0) g_wait(A.ctr, (unsigned) pi, GS, c, A.err);
            wbar(&c->cbar, cgen, ENG_NC, lane, c, A.err, 2);
        }
        ENG_STAMP(0);
        if (ph->kind == 1) {
#ifndef ENG_NO_ATTN
            if (b < ph->aH && wave < 4) {
                if (ph->adh == 128) eng_attention<128>(ph, b, acts, c, agen, A.err);
#ifndef ENG_NO_ATTN64
                else eng_attention<64>(ph, b, acts, c, agen, A.err);
#endif
            }
#endif
        } else {
            const GemvP & p = ph->g;
            const int ta = ph->ta, tb = ph->tb, pair = ph->pair;
            const int K = p.K;
            int8_t * xs_q = (int8_t *) acts; int * xs_gs = (int *) (acts + ((K + 15) & ~15)); float * xs_d = (float *) (xs_gs + K / 16);
#ifndef ENG_NO_PRO
            eng_prologue(p, xs_q, xs_gs, xs_d, wave, lane);
#endif
            wbar(&c->cbar, cgen, ENG_NC, lane, c, A.err, 2);
            ENG_STAMP(1);
            const XLds xs = {xs_q, xs_gs, xs_d, 0};
            // ---- rows: the head of job 0 out of the ring, everything else HBM -> registers (the mat-vec kernel's own row loops)
            JobGeo g0 = JobGeo(), g1 = JobGeo(), g2 = JobGeo();
            int nit = 0;
#define ENG_PH(TA_, TB_, P_, E_) nit = phase_rows<TA_, TB_, P_, E_>(A, ph, c, ring, xs, outbuf, nbase, wave, lane, b, G, g0, g1, g2)
            if (ph->epi) {
                if (ta == PM_Q4_K && tb == PM_Q4_K) ENG_PH(PM_Q4_K, PM_Q4_K, false, true);
                else if (ta == PM_Q4_K && tb == PM_Q6_K) ENG_PH(PM_Q4_K, PM_Q6_K, false, true);
                else if (ta == PM_Q4_K && tb == PM_Q5_K) ENG_PH(PM_Q4_K, PM_Q5_K, false, true);
#ifdef ENG_ALL_TYPES
                else ENG_PH(PM_Q6_K, PM_Q6_K, false, true);
#endif
            } else if (pair) {
                if (ta == PM_Q4_K) ENG_PH(PM_Q4_K, PM_Q4_K, true, false);
#ifdef ENG_ALL_TYPES
                else ENG_PH(PM_Q6_K, PM_Q6_K, true, false);
#endif
            } else {
                if (ta == PM_Q4_K) ENG_PH(PM_Q4_K, PM_Q4_K, false, false); else ENG_PH(PM_Q6_K, PM_Q6_K, false, false);
            }
#undef ENG_PH
            nbase += (unsigned) nit;
            ENG_STAMP(2);
            float ec0 = 1.0f, es0 = 0.0f, ec1 = 1.0f, es1 = 0.0f, ec2 = 1.0f, es2 = 0.0f;
            int epi_slot = 0; long epi_off = 0;
            if (ph->epi) {
                const int seq = p.epi.seq_ptr ? uniform_const_ptr(p.epi.seq_ptr)[0] : 0;
                epi_slot = uniform_const_ptr(p.epi.pos_ptr)[seq];
                epi_off = (long) seq * p.epi.seq_stride;
                qkv_cs(p.job[0], p.epi, g0.r0, g0.r1, tid, ec0, es0);
                qkv_cs(p.job[1], p.epi, g1.r0, g1.r1, tid, ec1, es1);
                qkv_cs(p.job[2], p.epi, g2.r0, g2.r1, tid, ec2, es2);
            }
            wbar(&c->cbar, cgen, ENG_NC, lane, c, A.err, 2);              // every row result of the workgroup is parked
            ENG_STAMP(3);
            // ---- epilogue: coalesced, write-through
            if (ph->epi) {
                write_out_qkv<true>(p.job[0], p.epi, outbuf, g0.r0, g0.r1, g0.ob, tid, g0.split ? g0.cpr : 1, epi_slot, epi_off, ec0, es0);
                write_out_qkv<true>(p.job[1], p.epi, outbuf, g1.r0, g1.r1, g1.ob, tid, g1.split ? g1.cpr : 1, epi_slot, epi_off, ec1, es1);
                write_out_qkv<true>(p.job[2], p.epi, outbuf, g2.r0, g2.r1, g2.ob, tid, g2.split ? g2.cpr : 1, epi_slot, epi_off, ec2, es2);
            } else {
                const double ss = write_out<true, 1>(p.job[0], outbuf, g0.r0, g0.r1, g0.ob, tid, 0, g0.split ? g0.cpr : 1);
                write_out<true, 1>(p.job[1], outbuf, g1.r0, g1.r1, g1.ob, tid, 0, g1.split ? g1.cpr : 1);
                write_out<true, 1>(p.job[2], outbuf, g2.r0, g2.r1, g2.ob, tid, 0, g2.split ? g2.cpr : 1);
                if (p.ss_out && wave == 0) {               // (rows <= 64 per workgroup: checked by the host)
                    const double ws = wave_sum_f64(ss);
                    if (lane == 0) __hip_atomic_store((PM_G double *) (p.ss_out + b), ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // ---- publish: this wave's stores have left, then the workgroup arrives
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wbar(&c->cbar, cgen, ENG_NC, lane, c, A.err, 2);
        ENG_STAMP(4);
        if (wave == 0 && lane == 0) g_arrive(A.ctr, (unsigned) pi, NGR, GS, pi == A.n_ph - 1);
        ENG_STAMP(5);
    }
#ifdef ENG_DEBUG
    if (A.dbg && b == 0 && tid == 0) { float * Ld = A.dbg + 64 * 16 + 64 * 8 + 64 + 32 * 16; Ld[5] = (float) g_cs_spin; Ld[6] = (float) g_cs_eat; g_cs_spin = g_cs_eat = 0; }
#endif
}

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
    if (f.K % 256 || nblk > (e.g.xmode == 3 ? 4 * ENG_NC : 8 * ENG_NC)) return -12;
    if ((size_t) ((f.K + 15) & ~15) + (size_t) (f.K / 16) * 4 + (size_t) ((nblk + 3) & ~3) * 4 > (size_t) ENG_ACT) return -12;
    if (pl->grid && pl->grid != grid) return -13;
    if (grid > ENG_MAXG) return -13;
    pl->grid = grid;
    // items: rows whose image exceeds what 15 waves can hold in the ring next to the run-ahead are consumed step by step
    int nres = 0;
    for (int j = 0; j < 3; ++j) {
        GemvJob & jb = e.g.job[j];
        if (jb.N <= 0) continue;
        const int type = jb.is_b ? tb : ta;
        const int ch = (type == PM_Q4_K || type == PM_Q6_K) ? PM_CH64 : PM_CH32, cpr = (((jb.U + 63) >> 6) + ch - 1) / ch;
        const int sb = type == PM_Q4_K ? ST<PM_Q4_K>::BYTES : type == PM_Q6_K ? ST<PM_Q6_K>::BYTES : ST<PM_Q5_K>::BYTES;
        // (a pair item holds the steps of both matrices: ffn_gate | ffn_up rows of 8192 Q6_K weights - Qwen2.5-72B - are 13 KiB, still one item)
        if (!jb.split && cpr * sb * (pair ? 2 : 1) > (pair ? 16 : 10) * 1024) { if (pair) return -14; jb.split = 1; }
        if (jb.split && cpr == 1) jb.split = 0;
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
    // LDS of the general path inside the activation area: part / pw / reductions (2 KiB + 64) | qs[dh] | part[256] | sc[max_keys + 8]
    if ((size_t) (512 + 8 + 8) * 4 + (size_t) (dh + 256 + ((max_keys + 15) & ~7)) * 4 > (size_t) ENG_ACT) return -12;
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
    if (!pl->d_dbg) { (void) hipMalloc((void **) &pl->d_dbg, (64 * 16 + 64 * 8 + 64 + 32 * 16 + 16) * 4); (void) hipMemset(pl->d_dbg, 0, (64 * 16 + 64 * 8 + 64 + 32 * 16 + 16) * 4); }
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
        std::vector<float> h(64 * 16 + 64 * 8 + 64 + 32 * 16 + 16);
        (void) hipMemcpy(h.data(), pl->d_dbg, h.size() * 4, hipMemcpyDeviceToHost);
        { const float * o = &h[64 * 16 + 64 * 8]; fprintf(stderr, "eng xs_q[0..31]:"); for (int i = 0; i < 32; ++i) fprintf(stderr, " %g", o[i]); fprintf(stderr, "\neng xs_d[0..3]: %g %g %g %g  gs[0..7]:", o[32], o[33], o[34], o[35]); for (int i = 0; i < 8; ++i) fprintf(stderr, " %g", o[36 + i]); fprintf(stderr, "\n"); }
        {
            const unsigned long long * t = (const unsigned long long *) &h[64 * 16 + 64 * 8 + 64];
            for (int ph = 0; ph < 32 && t[8 * ph]; ++ph)
                fprintf(stderr, "eng phase %2d (wg 0, us since launch entry): seam-in %.2f | prologue done %.2f | items done (wave 0) %.2f | all waves %.2f | epilogue + stores %.2f | arrived %.2f\n", ph,
                        (t[8 * ph] - t[0]) / 100.0, (t[8 * ph + 1] - t[0]) / 100.0, (t[8 * ph + 2] - t[0]) / 100.0, (t[8 * ph + 3] - t[0]) / 100.0, (t[8 * ph + 4] - t[0]) / 100.0, (t[8 * ph + 5] - t[0]) / 100.0);
            const float * L = &h[64 * 16 + 64 * 8 + 64 + 32 * 16];
            fprintf(stderr, "eng loader 0 (wg 0): walked %g items, done at %.2f us; shader cycles waiting for ring room %g | issuing %g | in s_waitcnt vmcnt %g\n", L[0], L[3], L[1], L[2], L[4]);
            fprintf(stderr, "eng consumer wave 0 (wg 0): shader cycles waiting for items %g | consuming %g\n", L[5], L[6]);
        }
        for (int l = 0; l < 4; ++l) { const float * o = &h[64 * 16 + 8 * l]; fprintf(stderr, "eng lane %2d: isum=%g msum=%g yd=%g acc=%g xq0=%g gs0=%g u=%g uv=%g\n", l, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]); }
        for (int i = 0; i < 64; ++i) if (h[16 * i + 15] != 0.0f) {
            const float * o = &h[16 * i];
            fprintf(stderr, "eng item n=%g off=%g slot=%g val=%g landed=%g U=%g steps=%g type=%g w0=%08x hdr0=%08x xq0=%g xd0=%g wave=%g c0=%g phase=%g\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7],
                    __builtin_bit_cast(uint32_t, o[8]), __builtin_bit_cast(uint32_t, o[9]), o[10], o[11], o[12], o[13], o[14]);
        }
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
