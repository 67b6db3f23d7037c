// attn_wo.hip — attention + wo of ONE layer as a single two-phase launch (engine.hip, default single-token decode path).
//
// Why: the fused RoPE + KV store + attention kernel is latency-bound (7.4 us at short contexts on 64 of the 256 CUs, moving ~0
// bytes) and the wo mat-vec that follows is a 9 us launch of which 6.4 us is HBM time. Here both are phases of one kernel:
//   phase 0: workgroups 0 .. H-1 run the attention of their head (attn_device.h, same code as the stand-alone kernel) and publish
//            the head's output write-through; all others fall through
//   phase 1: the wo mat-vec (+ residual) of mmvq_device.h in its persistent-kernel form: weight loads are put in flight FIRST, then
//            the workgroup waits for all heads (split device-wide barrier of mmvq_device.h: two-level arrival, release flags,
//            bounded spin), quantizes the attention output and runs its rows
// so the wo weights travel from HBM while the attention latency chain runs, and one launch boundary disappears.
// 512-thread workgroups (PM_GEMV_BLOCK below): the attention body keeps a K row per thread in registers and does not fit the
// 128-VGPR budget of a 1024-thread workgroup - at 1024 threads the compiler spilled 165 VGPRs (388 B of scratch per lane), and a
// launch that needs scratch costs tens of microseconds of dispatch set-up (measured: +40 us per launch).
// Results are bit-identical to the two-launch path (same device functions, same reduction orders; tests/test_gpu_engine.py).
#define PM_GEMV_BLOCK 512
#include "mmvq_device.h"
#include "attn_device.h"
#include "pm355_layer_ops.h"

using namespace pmv;

namespace {

constexpr size_t PM_BAR_BYTES = 33 * 128;    // top counter + 16 group counters + 16 release flags, one 128-byte line each

template <int T>
__global__ __launch_bounds__(PM_GEMV_BLOCK, 2) void attn_wo_kernel(AttnP a, GemvP g, unsigned * ctr, int * err) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double nred[PM_GEMV_NW];
    __shared__ float redf[8];
    __shared__ double redd[4];
    const unsigned G = gridDim.x, NG = (G % 16 == 0) ? 16 : 1;
    if ((int) blockIdx.x < a.H) attn_rope_body<128, true>(a, blockIdx.x, smem, redf, redd);
    grid_arrive(GridBar{ctr, err, 0u, NG, G / NG, 0u});
    gemv_body<T, T, false, false, true>(g, smem, nred, GridBar{ctr, err, 1u, NG, G / NG, 0u});
    grid_arrive(GridBar{ctr, err, 1u, NG, G / NG, 1u});
}

// measurement: n empty phases = n device-wide barriers (pm355_probe_grid_barrier)
__global__ __launch_bounds__(PM_GEMV_BLOCK, 2) void barrier_probe_kernel(int n, unsigned * ctr, int * err) {
    const unsigned G = gridDim.x, NG = (G % 16 == 0) ? 16 : 1;
    for (int ph = 0; ph < n; ++ph) {
        const GridBar bar = {ctr, err, (unsigned) ph, NG, G / NG, ph == n - 1 ? 1u : 0u};
        grid_wait(bar);
        grid_arrive(bar);
    }
}

} // namespace

// n device-wide barriers in one launch on `ctr` (pm_attn_wo_bar_bytes() of zeroed device memory; left zeroed)
void pm_launch_barrier_probe(int n, void * ctr, hipStream_t st) {
    hipLaunchKernelGGL(barrier_probe_kernel, dim3(pm_device_cus()), dim3(PM_GEMV_BLOCK), 0, st, n, (unsigned *) ctr, (int *) ((char *) ctr + PM_BAR_BYTES));
}

// Launches attention + wo of one layer as one kernel; -1 when this shape / type has no such kernel (caller keeps the two launches).
// `ctr` = PM_ATTN_WO_BAR_BYTES of zero-initialised device memory (the kernel leaves it zeroed), err = watchdog flag (int).
int pm_launch_attn_wo(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * pos0, const int32_t * seq,
                      long seq_stride, const float * freq_factors, float * att, int H, int Hkv, int dh, int n_ctx, float scale,
                      const pm_rope_cfg & c, const pm_gemv_fused & f, void * ctr, hipStream_t st) {
    if (dh != 128 || n_ctx % 8 || f.njobs != 1 || f.job[0].W2 || f.dbg_int || f.norm_w || f.xq) return -1;
    const int grid = pm_device_cus();
    if (H > grid) return -1;
    GemvP g; int ta, tb, gr; bool pair; size_t lds_g;
    if (gemv_fill(f, grid, g, ta, tb, pair, lds_g, gr)) return -1;
    if (ta != tb || pair || (ta != PM_Q4_K && ta != PM_Q6_K)) return -1;
    const size_t lds_a = (size_t) (4 * dh + 256 + n_ctx + 8) * 4;
    const size_t lds = lds_a > lds_g ? lds_a : lds_g;
    if (lds > 150 * 1024) return -1;
    RopeP r;
    r.n_dims = c.n_dims; r.mode = c.mode; r.n_ctx_orig = c.n_ctx_orig; r.theta_scale = c.theta_scale;
    r.freq_scale = c.freq_scale; r.ext_factor = c.ext_factor; r.attn_factor = c.attn_factor; r.corr0 = c.corr0; r.corr1 = c.corr1;
    AttnP a = {q, k, v, (uint16_t *) kc, (uint16_t *) vc, pos0, seq, seq_stride, freq_factors, att, H, Hkv, n_ctx, scale, r, nullptr, nullptr};
    unsigned * cp = (unsigned *) ctr; int * ep = (int *) ((char *) ctr + PM_BAR_BYTES);
    auto go = [&](auto kern) {
        static bool attr[16] = {};
        const int dv = pm_cur_dev();
        if (lds > 48 * 1024 && !attr[dv]) { (void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr[dv] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PM_GEMV_BLOCK), lds, st, a, g, cp, ep);
    };
    if (ta == PM_Q4_K) go(attn_wo_kernel<PM_Q4_K>); else go(attn_wo_kernel<PM_Q6_K>);
    return 0;
}
size_t pm_attn_wo_bar_bytes() { return PM_BAR_BYTES + 64; }

