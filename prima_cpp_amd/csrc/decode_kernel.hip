// decode_kernel.hip — EXPERIMENT (off by default, PM355_PERSISTENT=1): ONE persistent kernel per decoded token.
//
// Idea: the 5 per-layer launches of the decode path (QKV mat-vec, fused RoPE + KV + attention, wo, gate/up, down;
// engine.hip) each cost ~4.5 us of dispatch ramp + drain on top of their HBM time. Here they are PHASES of one kernel
// with one 1024-thread workgroup pinned on every CU, separated by a SPLIT device-wide barrier: a phase first puts its
// first weight loads in flight (weights depend on nothing), only then waits for the previous phase of all workgroups.
// Phase code = the very same device functions the stand-alone launches run (mmvq_device.h, attn_device.h): results are
// bit-identical to the multi-launch path (tests/test_gpu_engine.py::test_persistent_kernel_matches_launch_path).
//
// Measured on Llama-3-70B Q4_K_M (MI355X): 9.5 ms/token against 8.6 ms for the 5-launch path, so it is NOT the default.
// What was learned on the way (each step measured):
//   * 256 device-scope atomics on one counter serialize at the memory side: 20 us per barrier. Two-level arrival
//     (16 groups of 16) + per-group release flags: 2.1 us per barrier (pm355_probe_grid_barrier).
//   * agent-scope fences (buffer_wbl2 / buffer_inv) from every wave: 200 us per barrier. Instead: write-through (sc1)
//     activation stores, every activation buffer written once per kernel (per-layer scratch), plain cached loads.
//   * device-coherent (sc1) dword LOADS of the activations: 2x slower tokens (256 workgroups x 32 KB uncached).
//   * phase bodies as non-inlined functions: pointers arrive in VGPRs -> FLAT loads and vmcnt(0) lgkmcnt(0) in the row
//     loop, and ~100 callee-saved VGPRs per lane go through scratch per call (~50 MB per phase chip-wide); descriptors
//     are therefore read through uniform constant-address-space pointers and everything is inlined.
//   * what remains is structural: barrier (2 us) + first-touch fetch of the activations written by other XCDs (~1.5 us)
//     + the quantizing prologue per phase is about what a launch boundary costs, and only 2 pre-issued weight steps
//     (24 MB chip-wide, 3.7 us of HBM time) bridge it.
//
// Safety: every workgroup must be resident (grid = number of CUs, one workgroup per CU by its register / thread
// budget). The barrier spin is bounded (watchdog in grid_wait): if a workgroup were never scheduled the kernel still
// terminates and raises the plan's error flag (pm355_model_check) instead of hanging the GPU.
#include "mmvq_device.h"
#include "attn_device.h"
#include "pm355_layer_ops.h"
#include <vector>

using namespace pmv;

namespace {

enum { PH_GEMV = 0, PH_ATTN = 1, PH_NOP = 2 };
constexpr size_t PM_BAR_BYTES = 33 * 128;    // top counter + 16 group counters + 16 release flags, one 128-byte line each
// combos of gemv_body instantiated in the persistent kernel (the Q4_K_M and Q6_K model mixtures)
enum { C_44 = 0, C_45, C_46, C_66, C_55, C_44P, C_66P, C_NONE };

struct Phase {
    int kind, combo, dh, pad_;
    GemvP g;
    AttnP a;
};

// Every phase body is a real (non-inlined) function: each gets its own register allocation under the 128-VGPR budget
// (inlined into one giant switch the allocator spilled 1.1 KB per lane). The LDS objects are declared INSIDE the phase
// functions so that their address space is known (passed as generic pointers every LDS access would be a FLAT access).
template <int TA, int TB, bool PAIR>
__device__ __forceinline__ void phase_gemv(const GemvP * p, unsigned * ctr, int * err, unsigned phase, unsigned ngroups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double nred[PM_GEMV_NW];
    // descriptor and barrier state are read through uniform constant-address-space pointers: s_load into SGPRs at
    // the point of use (a by-value copy kept ~60 SGPRs live across the row loop and spilled them)
    // (arguments arrive in VGPRs: make the barrier state scalar again)
    GridBar bl;
    bl.ctr = uniform_ptr(ctr); bl.err = uniform_ptr(err);
    bl.phase = (unsigned) __builtin_amdgcn_readfirstlane((int) phase);
    bl.ngroups = (unsigned) __builtin_amdgcn_readfirstlane((int) ngroups);
    bl.gsize = gridDim.x / bl.ngroups;
    gemv_body<TA, TB, PAIR, false, true>(*uniform_const_ptr(p), smem, nred, bl);
}
template <int DH>
__device__ __forceinline__ void phase_attn(const AttnP * a, int h) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float redf[8];
    __shared__ double redd[4];
    const AttnP al = *uniform_const_ptr(a);
    attn_rope_body<DH, true>(al, __builtin_amdgcn_readfirstlane(h), smem, redf, redd);
}

__global__ __launch_bounds__(PM_GEMV_BLOCK, 4) void decode_window_kernel(const Phase * phases, int n_phases, unsigned * ctr, int * err) {
    const unsigned G = gridDim.x, NG = (G % 16 == 0) ? 16 : 1;
    for (int ph = 0; ph < n_phases; ++ph) {
        const Phase & P = phases[ph];
        const GridBar bar = {ctr, err, (unsigned) ph, NG, G / NG, ph == n_phases - 1 ? 1u : 0u};
        if (P.kind == PH_GEMV) {
            switch (P.combo) {
                case C_44:  phase_gemv<PM_Q4_K, PM_Q4_K, false>(&P.g, ctr, err, (unsigned) ph, NG); break;
                case C_45:  phase_gemv<PM_Q4_K, PM_Q5_K, false>(&P.g, ctr, err, (unsigned) ph, NG); break;
                case C_46:  phase_gemv<PM_Q4_K, PM_Q6_K, false>(&P.g, ctr, err, (unsigned) ph, NG); break;
                case C_66:  phase_gemv<PM_Q6_K, PM_Q6_K, false>(&P.g, ctr, err, (unsigned) ph, NG); break;
                case C_55:  phase_gemv<PM_Q5_K, PM_Q5_K, false>(&P.g, ctr, err, (unsigned) ph, NG); break;
                case C_44P: phase_gemv<PM_Q4_K, PM_Q4_K, true>(&P.g, ctr, err, (unsigned) ph, NG); break;
                default:    phase_gemv<PM_Q6_K, PM_Q6_K, true>(&P.g, ctr, err, (unsigned) ph, NG); break;
            }
        } else if (P.kind == PH_NOP) {
            grid_wait(bar);
        } else {
            grid_wait(bar);
            if ((int) blockIdx.x < P.a.H) {
                phase_attn<128>(&P.a, blockIdx.x);                  // head_dim 128 only (other sizes keep the 5-launch path)
            }
        }
        grid_arrive(bar);
    }
}

} // namespace

struct pm_decode_plan {
    std::vector<Phase> host;
    Phase * dev = nullptr;
    unsigned * ctr = nullptr;          // barrier counters (top + 16 groups, one 128-B line each), then the watchdog flag
    size_t lds = 0;
    int grid = 0;
    bool attr_set = false;
};

pm_decode_plan * pm_decode_plan_new() {
    pm_decode_plan * pl = new pm_decode_plan();
    pl->grid = pm_device_cus();
    return pl;
}

void pm_decode_plan_free(pm_decode_plan * pl) {
    if (!pl) return;
    if (pl->dev) (void) hipFree(pl->dev);
    if (pl->ctr) (void) hipFree(pl->ctr);
    delete pl;
}

int pm_decode_plan_add_gemv(pm_decode_plan * pl, const pm_gemv_fused & f) {
    Phase ph = {};
    ph.kind = PH_GEMV;
    int ta, tb, grid; bool pair; size_t lds;
    if (f.dbg_int) return -1;
    const int rc = gemv_fill(f, pl->grid, ph.g, ta, tb, pair, lds, grid);
    if (rc) return rc;
    int combo = C_NONE;
    if (pair) combo = (ta == PM_Q4_K) ? C_44P : (ta == PM_Q6_K) ? C_66P : C_NONE;
    else if (ta == PM_Q4_K && tb == PM_Q4_K) combo = C_44;
    else if (ta == PM_Q4_K && tb == PM_Q5_K) combo = C_45;
    else if (ta == PM_Q4_K && tb == PM_Q6_K) combo = C_46;
    else if (ta == PM_Q6_K && tb == PM_Q6_K) combo = C_66;
    else if (ta == PM_Q5_K && tb == PM_Q5_K) combo = C_55;
    if (combo == C_NONE) return -1;
    ph.combo = combo;
    if (lds > pl->lds) pl->lds = lds;
    pl->host.push_back(ph);
    return 0;
}

int pm_decode_plan_add_attn(pm_decode_plan * pl, const float * q, const float * k, const float * v, void * kc, void * vc,
                            const int32_t * pos0, const int32_t * seq, long seq_stride, const float * freq_factors, float * out,
                            int H, int Hkv, int dh, int n_ctx, float scale, const pm_rope_cfg & c) {
    if (dh != 128 || n_ctx % 8 || H > pl->grid) return -1;
    const size_t lds = (size_t) (4 * dh + 256 + n_ctx + 8) * 4;
    if (lds > 150 * 1024) return -1;
    Phase ph = {};
    ph.kind = PH_ATTN; ph.dh = dh;
    RopeP r;
    r.n_dims = c.n_dims; r.mode = c.mode; r.n_ctx_orig = c.n_ctx_orig; r.theta_scale = c.theta_scale;
    r.freq_scale = c.freq_scale; r.ext_factor = c.ext_factor; r.attn_factor = c.attn_factor; r.corr0 = c.corr0; r.corr1 = c.corr1;
    ph.a = AttnP{q, k, v, (uint16_t *) kc, (uint16_t *) vc, pos0, seq, seq_stride, freq_factors, out, H, Hkv, n_ctx, scale, r};
    if (lds > pl->lds) pl->lds = lds;
    pl->host.push_back(ph);
    return 0;
}

int pm_decode_plan_finish(pm_decode_plan * pl) {
    if (pl->host.empty()) return -1;
    if (hipMalloc((void **) &pl->dev, pl->host.size() * sizeof(Phase)) != hipSuccess) return -2;
    if (hipMalloc((void **) &pl->ctr, PM_BAR_BYTES + 64) != hipSuccess) return -2;
    if (hipMemcpy(pl->dev, pl->host.data(), pl->host.size() * sizeof(Phase), hipMemcpyHostToDevice) != hipSuccess) return -2;
    if (hipMemset(pl->ctr, 0, PM_BAR_BYTES + 64) != hipSuccess) return -2;
    if (pl->lds > 48 * 1024) {
        if (hipFuncSetAttribute((const void *) decode_window_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return -2;
    }
    return 0;
}

int pm_decode_plan_launch(pm_decode_plan * pl, hipStream_t st) {
    if (!pl->dev) return -1;
    // (no memset node: the last arriver of the final phase zeroes the barrier state again - grid_arrive; the buffer starts zeroed)
    hipLaunchKernelGGL(decode_window_kernel, dim3(pl->grid), dim3(PM_GEMV_BLOCK), pl->lds, st,
                       (const Phase *) pl->dev, (int) pl->host.size(), pl->ctr, (int *) (pl->ctr + PM_BAR_BYTES / 4));
    return 0;
}

int pm_decode_plan_error(pm_decode_plan * pl) {
    int e = 0;
    if (!pl->ctr) return 0;
    if (hipMemcpy(&e, pl->ctr + PM_BAR_BYTES / 4, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return e;
}

// measurement helper: a plan of n empty phases = n device-wide barriers
int pm_decode_plan_add_nop(pm_decode_plan * pl, int n) {
    Phase ph = {};
    ph.kind = PH_NOP;
    for (int i = 0; i < n; ++i) pl->host.push_back(ph);
    if (pl->lds < 1024) pl->lds = 1024;
    return 0;
}
