// engine_probe.hip — MEASUREMENT SKELETON (not on the product path): what would a decode layer cost as ONE persistent launch built
// on a run-ahead LDS-DMA weight loader (MI355X_MICROARCH.md "engine-vs-launches", "prefetch-credit", "ldsdma-fill")?
//
// Why: the 5-launches-per-layer decode path sits at 61 % of the HBM roofline; the earlier persistent kernel (register-held
// pre-issue across a device-wide barrier) lost because a wave that waits at a barrier cannot keep loads in
// flight. The only structure that can is a DEDICATED loader wave that streams the weights of all phases, in order, into an LDS ring
// with global_load_lds (no registers, no dependence on activations) while consumer waves work through landed slots and stall at
// the seams. Before building that engine for real (Q4_K / Q6_K consumers, attention phase, bit-parity) this skeleton measures
// its ceiling with the real byte counts, the real seams and representative consumer work:
//   * one workgroup per CU, NW waves: wave NW-1 = loader, waves 0 .. NW-2 = consumers
//   * loader: for every (layer, phase, chunk) of this CU, in order: wait until ring slot (k mod NS) is free, 16 x
//     global_load_lds_dwordx4 (16 KiB), thinned to `thin` outstanding fills (s_waitcnt vmcnt), publish `landed` in LDS
//   * every consumer wave takes its share (1-KiB pieces w, w + NC, ...) of every fill: ds_read_b128 of the slot + 2 x ds_read_b128 of int8 activations per
//     16 weight bytes, nibble unpack, 8 x v_dot4_i32_i8, a float scale step every 4 KiB - the instruction mix of the Q4_K mat-vec
//   * seam after every phase: write-through stores of the phase's outputs, consumer-only LDS barrier, two-level device-wide
//     barrier (same counters as mmvq_device.h), then every workgroup re-reads the FULL activation vector of the next phase
//     (plain loads, per-(layer, phase) buffers: first touch), abs-max + int8 conversion into LDS, consumer-only LDS barrier
//   * the attention phase has no weights: 64 workgroups hold for `attn_us` microseconds (the latency chain), all others pass
// Every spin is bounded; a time-out raises *err, sets an LDS give-up word and the kernel runs to completion without waiting.
#include "pm355_device.h"
#include "pm355_probe.h"

namespace {

constexpr int FILL = 16384;
constexpr int MAXPH = 8;
typedef __attribute__((address_space(3))) void * lds_ptr;

struct EngP {
    const uint8_t * w; long region_stride; int n_regions;   // weights of layer l: w + (l % n_regions) * region_stride
    int n_layers, nph;
    int chunks[MAXPH];      // 16-KiB fills per CU in phase p (0: no weights)
    long off[MAXPH];        // byte offset of phase p inside a layer region
    int act_n[MAXPH];       // floats of the activation vector every workgroup reads before phase p
    int out_n[MAXPH];       // floats phase p writes (whole chip)
    int attn_ph;            // index of the weight-less attention phase (-1: none)
    float * act;            // [n_layers][nph][act_stride] floats
    long act_stride;
    unsigned * ctr; int * err;
    float attn_us; int ns, nt, thin;
};

// ---- LDS control block (after ring + activation area) ----
struct Ctl { unsigned landed, cbar, giveup, pad; unsigned done[16]; };

// control words are accessed ONLY through LDS-typed pointers (ds_read_b32 / ds_write_b32; a generic volatile access is a FLAT sc0 sc1
// access followed by s_waitcnt vmcnt(0), which would drain the loader's DMA queue at every poll)
typedef __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ lds_u32 * L(unsigned * p) { return (lds_u32 *) p; }
__device__ __forceinline__ unsigned lds_ld(unsigned * p) { return __hip_atomic_load(L(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned * p, unsigned v) { __hip_atomic_store(L(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// the loader's publish: an asm ds_write - for a compiler-visible LDS store after global_load_lds the backend inserts s_waitcnt vmcnt(0)
// (it must assume the DMA writes alias), which would drain the loader's queue after every fill
__device__ __forceinline__ void lds_st_asm(unsigned * p, unsigned v) {
    asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned) (uintptr_t) L(p)), "v"(v) : "memory");
}

// ... and the loader's polls: asm ds_read + its own lgkmcnt wait, for the same reason
__device__ __forceinline__ unsigned lds_ld_asm(unsigned * p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned) (uintptr_t) L(p)) : "memory");
    return v;
}
__device__ __forceinline__ bool loader_spin_until_ge(unsigned * word, unsigned target, unsigned * giveup, int * err, int code) {
    int spins = 0;
    while (lds_ld_asm(word) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (lds_ld_asm(giveup)) return false;
        if (++spins > (1 << 22)) { lds_st_asm(giveup, 1); st_g(err, code); return false; }
    }
    return true;
}

__device__ __forceinline__ bool spin_until_ge(unsigned * word, unsigned target, unsigned * giveup, int * err, int code) {
    int spins = 0;
    while (lds_ld(word) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (lds_ld(giveup)) return false;
        if (++spins > (1 << 22)) { lds_st(giveup, 1); st_g(err, code); return false; }
    }
    return true;
}

// consumer-only barrier: NC waves, monotonic counter (gen-th use completes at gen * NC arrivals)
__device__ __forceinline__ void cbar(Ctl * c, unsigned & gen, int NC, int lane, int * err) {
    ++gen;
    __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this wave's LDS writes are done
    if (lane == 0) {
        __hip_atomic_fetch_add(L(&c->cbar), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        spin_until_ge(&c->cbar, gen * (unsigned) NC, &c->giveup, err, 2);
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void g_arrive(unsigned * ctr, unsigned phase, unsigned ngroups, unsigned gsize, bool last) {
    unsigned * g = ctr + 32 * (1 + blockIdx.x / gsize);
    const unsigned old = __hip_atomic_fetch_add((PM_G unsigned *) g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old + 1) % gsize == 0) {
        const unsigned t = __hip_atomic_fetch_add((PM_G unsigned *) ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == ngroups * (phase + 1)) {
            if (last) {
                __hip_atomic_store((PM_G unsigned *) ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned k = 0; k < ngroups; ++k) {
                    __hip_atomic_store((PM_G unsigned *) (ctr + 32 * (1 + k)), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((PM_G unsigned *) (ctr + 32 * (17 + k)), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                for (unsigned k = 0; k < ngroups; ++k)
                    __hip_atomic_store((PM_G unsigned *) (ctr + 32 * (17 + k)), phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
__device__ __forceinline__ void g_wait(unsigned * ctr, unsigned phase, unsigned gsize, Ctl * c, int * err) {   // phases 0 .. phase-1 complete
    const unsigned * flag = ctr + 32 * (17 + blockIdx.x / gsize);
    int spins = 0;
    while (__hip_atomic_load((const PM_G unsigned *) flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) {
        __builtin_amdgcn_s_sleep(2);
        if (lds_ld(&c->giveup)) return;
        if (++spins > (1 << 21)) { lds_st(&c->giveup, 1); st_g(err, 3); return; }
    }
}

template <int NW, bool NT>
__global__ __launch_bounds__(NW * 64, 1) void engine_probe_kernel(EngP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ Ctl ctl;                                           // static: its address space is known (ds_read / ds_write, never FLAT)
    constexpr int NC = NW - 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NS = p.ns;
    const int ACT_BYTES = NS >= 8 ? 16384 : 32768;               // int8 activation area (power of two; 8 slots leave 16 KiB)
    char * ring = smem;
    char * acts = smem + (size_t) NS * FILL;
    Ctl * c = &ctl;
    if (threadIdx.x < sizeof(Ctl) / 4) ((unsigned *) c)[threadIdx.x] = 0;
    __syncthreads();                                              // the only workgroup-wide barrier: before the roles split


    if (wave == NW - 1) {
        // ---------------- loader ----------------
        unsigned k = 0;                                           // fill index (this CU)
        for (int l = 0; l < p.n_layers; ++l) {
            const uint8_t * reg = p.w + (long) (l % p.n_regions) * p.region_stride;
            for (int ph = 0; ph < p.nph; ++ph) {
                const uint8_t * src0 = reg + p.off[ph] + (long) blockIdx.x * p.chunks[ph] * FILL;
                for (int ch = 0; ch < p.chunks[ph]; ++ch, ++k) {
                    const unsigned slot = k % (unsigned) NS;
                    if (k >= (unsigned) NS && lds_ld_asm(&c->done[slot]) < (k / (unsigned) NS) * (unsigned) NC) {
                        // ring full: nothing can be issued anyway -> drain and publish everything that is in flight (the thinned
                        // publish below would otherwise hold the last fills back until two more have been issued)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) lds_st_asm(&c->landed, k);
                        loader_spin_until_ge(&c->done[slot], (k / (unsigned) NS) * (unsigned) NC, &c->giveup, p.err, 1);   // all NC consumers retired every earlier use of the slot
                    }
                    const uint8_t * src = src0 + (long) ch * FILL + lane * 16;
                    char * dst = ring + (size_t) slot * FILL;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        __builtin_amdgcn_global_load_lds((const PM_G void *) (src + i * 1024), (lds_ptr) (dst + i * 1024), 16, 0, NT ? 2 : 0);
                    }
                    // thinned: at most `thin` fills outstanding; everything older has landed
                    unsigned landed;
                    if (p.thin >= 3)      { asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); landed = k >= 3 ? k - 2 : 0; }
                    else if (p.thin == 2) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); landed = k >= 2 ? k - 1 : 0; }
                    else                  { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); landed = k; }
                    if (lane == 0) lds_st_asm(&c->landed, landed);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) lds_st_asm(&c->landed, k);
        return;
    }

    // ---------------- consumers ----------------
    const unsigned G = gridDim.x, NGR = (G % 16 == 0) ? 16 : 1, GS = G / NGR;
    unsigned gen = 0;                                            // consumer-barrier generation
    unsigned kbase = 0;                                          // first fill index of the current phase
    float facc = 0.0f;
    const int total_ph = p.n_layers * p.nph;
    for (int l = 0; l < p.n_layers; ++l) {
        for (int ph = 0; ph < p.nph; ++ph) {
            const int gph = l * p.nph + ph;
            // ---- seam: outputs of the previous phase of ALL workgroups ----
            if (gph > 0) {
                if (wave == 0 && lane == 0) g_wait(p.ctr, (unsigned) gph, GS, c, p.err);
                cbar(c, gen, NC, lane, p.err);
            }
            // ---- prologue: the whole activation vector -> int8 in LDS ----
            {
                const float * a = p.act + ((long) gph * p.act_stride);
                const int n4 = p.act_n[ph] / 4;
                float amax = 0.0f;
                for (int i = wave * 64 + lane; i < n4; i += NC * 64) {
                    const float4 v = ld_g((const float4 *) a + i);
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                    const float s = 127.0f / (1.0f + amax);
                    const uint32_t q = ((uint32_t) (int) (v.x * s) & 255u) | (((uint32_t) (int) (v.y * s) & 255u) << 8) |
                                       (((uint32_t) (int) (v.z * s) & 255u) << 16) | (((uint32_t) (int) (v.w * s) & 255u) << 24);
                    ((uint32_t *) acts)[i & (ACT_BYTES / 4 - 1)] = q;
                }
                facc += amax * 1e-30f;
                cbar(c, gen, NC, lane, p.err);
            }
            if (ph == p.attn_ph) {
                // ---- attention stand-in: a latency chain on the head workgroups ----
                if (blockIdx.x < 64 && p.attn_us > 0.0f) {
                    const uint64_t t0 = wall_clock64();
                    const uint64_t dt = (uint64_t) (p.attn_us * 100.0f);                 // 100 MHz constant clock
                    while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(4);
                }
            } else {
                // ---- main: every consumer wave takes its share (1-KiB pieces w, w + NC, ...) of EVERY fill of the phase ----
                const int nch = p.chunks[ph];
                for (int ch = 0; ch < nch; ++ch) {
                    const unsigned k = kbase + ch;
                    spin_until_ge(&c->landed, k + 1, &c->giveup, p.err, 4);
                    const char * slot = ring + (size_t) (k % (unsigned) NS) * FILL;
                    int acc = 0;
#pragma unroll
                    for (int i0 = 0; i0 < 16; i0 += NC) {
                        const int i = i0 + wave;
                        if (i < 16) {
                            const u32x4 wv = *(const u32x4 *) (slot + i * 1024 + lane * 16);
                            const int ao = ((ch * 16 + i) * 2048 + lane * 32) & (ACT_BYTES - 1);
                            const u32x4 a0 = *(const u32x4 *) (acts + ao), a1 = *(const u32x4 *) (acts + ao + 16);
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                acc = dot4(wv[d] & 0x0F0F0F0Fu, a0[d], acc);
                                acc = dot4((wv[d] >> 4) & 0x0F0F0F0Fu, a1[d], acc);
                            }
                        }
                    }
                    facc += (float) acc * 0.001f;
                    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's part of the slot has been read
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) __hip_atomic_fetch_add(L(&c->done[k % (unsigned) NS]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                kbase += nch;
            }
            // ---- epilogue: this workgroup's outputs, write-through, into the NEXT phase's activation buffer ----
            if (gph + 1 < total_ph) {
                float * o = p.act + ((long) (gph + 1) * p.act_stride);
                const int per_wg = (p.out_n[ph] + (int) G - 1) / (int) G;
                for (int i = wave * 64 + lane; i < per_wg; i += NC * 64) {
                    const int idx = blockIdx.x * per_wg + i;
                    if (idx < p.out_n[ph]) st_act<true>(o + idx, facc + (float) i);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            cbar(c, gen, NC, lane, p.err);
            if (wave == 0 && lane == 0) g_arrive(p.ctr, (unsigned) gph, NGR, GS, gph == total_ph - 1);
        }
    }
    if (facc == 1.2345e-20f) st_g((float *) p.act, facc);
}

} // namespace

// One persistent launch of n_layers layers of `nph` phases. ctr: 33*128 + 64 zeroed bytes (left zeroed); the watchdog flag is the int
// at byte 33*128. Returns -1 for an unsupported geometry.
int pm_launch_engine_probe(const void * w, long region_stride, int n_regions, int n_layers, int nph, const int * chunks, const int * act_n,
                           const int * out_n, int attn_ph, float attn_us, float * act, long act_stride, void * ctr, int nw, int ns, int nt,
                           int thin, hipStream_t st) {
    if (nph < 1 || nph > MAXPH || ns < 2 || ns > 8 || (nw != 4 && nw != 8 && nw != 16)) return -1;
    EngP p = {};
    p.w = (const uint8_t *) w; p.region_stride = region_stride; p.n_regions = n_regions; p.n_layers = n_layers; p.nph = nph;
    hipDeviceProp_t pr_; int dev_ = 0; (void) hipGetDevice(&dev_);
    const int grid = hipGetDeviceProperties(&pr_, dev_) == hipSuccess ? pr_.multiProcessorCount : 256;
    long off = 0;
    for (int i = 0; i < nph; ++i) {
        p.chunks[i] = chunks[i]; p.act_n[i] = act_n[i]; p.out_n[i] = out_n[i]; p.off[i] = off;
        off += (long) chunks[i] * FILL * grid;
        if (act_n[i] > 28672 || act_n[i] % 4 || act_n[i] > act_stride || out_n[i] > act_stride) return -1;
    }
    if (off > region_stride) return -1;
    p.attn_ph = attn_ph; p.attn_us = attn_us; p.act = act; p.act_stride = act_stride;
    p.ctr = (unsigned *) ctr; p.err = (int *) ((char *) ctr + 33 * 128);
    p.ns = ns; p.nt = nt; p.thin = thin;
    const size_t lds = (size_t) ns * FILL + (ns >= 8 ? 16384 : 32768);
    auto go = [&](auto kern, int threads) {
        (void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, p);
    };
    if (nt) { if (nw == 4) go(engine_probe_kernel<4, true>, 256); else if (nw == 8) go(engine_probe_kernel<8, true>, 512); else go(engine_probe_kernel<16, true>, 1024); }
    else    { if (nw == 4) go(engine_probe_kernel<4, false>, 256); else if (nw == 8) go(engine_probe_kernel<8, false>, 512); else go(engine_probe_kernel<16, false>, 1024); }
    return 0;
}
