// mmvq_cols.hip — multi-column quantized GEMV (2..8 token batches: speculative decode, short prompts tails).
//
// Same body as mmvq.hip's single-column kernel (mmvq_device.h gemv_body with NC > 1): the weights are streamed ONCE and every
// decoded unit is dotted with NC activation slices held in LDS. NC accumulator sets do not fit the 128-VGPR budget of the
// 1024-thread workgroup mmvq.hip uses (measured: 36..252 B scratch per lane, a 4-column launch cost 2x a weight-bound pass,
// profiles/r02_cols_probe.txt), so this translation unit compiles the body for 512-thread workgroups, ONE per CU = 2 waves per
// SIMD = the full 256-VGPR budget; the grid stays one workgroup per CU.
// Replaces the batch>1 rows of ggml_mul_mat's vec_dot loop (reference ggml/src/ggml.c ggml_compute_forward_mul_mat, nrc = 1 per column).
#define PM_GEMV_BLOCK 512
#define PM_RCOLS(NC_) ((NC_) <= 4 ? 2 : 1)
#include "mmvq_device.h"

using namespace pmv;

namespace {

template <int T, int NC, bool PAIR>
__global__ __launch_bounds__(PM_GEMV_BLOCK, 2) void gemv_q_cols_kernel(GemvP p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double nred[PM_GEMV_NW];
    // what a multi-column launch never uses is pinned at compile time (as the hot single-token instantiations of mmvq.hip: profiles/r05_kernel_specialization.txt):
    // pre-quantized activation columns, one job, no debug output
    GemvP p = p_in;
    p.xmode = 0; p.dbg = nullptr; p.job[1].N = 0; p.job[2].N = 0; p.job[0].split = 0; p.job[0].is_b = 0; p.job[0].role = 0; p.ss_out = nullptr;
    gemv_body<T, T, PAIR, false, NC>(p, smem, nred);
}

template <int T>
int launch_cols(const GemvP & p, int nc, bool pair, int grid, size_t lds, hipStream_t st) {
    auto go = [&](auto kern) {
        (void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PM_GEMV_BLOCK), lds, st, p);
    };
    if (pair) {                                       // ffn_gate | ffn_up of a 2-token step: silu(gate) * up per column, one pass over both matrices
        if constexpr (T == PM_Q4_K || T == PM_Q6_K) { if (nc == 2) { go(gemv_q_cols_kernel<T, 2, true>); return 0; } }
        return -1;
    }
    if (nc == 8) {
        if constexpr (T == PM_Q5_K) return -1;        // 8 Q5_K accumulator sets spill (168 B / lane): the caller issues 2 x 4
        else go(gemv_q_cols_kernel<T, 8, false>);
    } else if (nc == 4) go(gemv_q_cols_kernel<T, 4, false>);
    else if (nc == 2) go(gemv_q_cols_kernel<T, 2, false>);
    else return -1;
    return 0;
}

}  // namespace

// p: filled by gemv_fill (mmvq.hip) for ONE job, with ncols (the columns of the batch, <= nc: the kernel's column slots) / xq_stride / y_stride set;
// grid = one workgroup per CU
int pm_launch_gemv_cols(int type, const GemvP & p, int nc, bool pair, int grid, size_t lds, hipStream_t st) {
    switch (type) {
        case PM_Q4_K: return launch_cols<PM_Q4_K>(p, nc, pair, grid, lds, st);
        case PM_Q5_K: return launch_cols<PM_Q5_K>(p, nc, pair, grid, lds, st);
        case PM_Q6_K: return launch_cols<PM_Q6_K>(p, nc, pair, grid, lds, st);
        case PM_Q8_0: return launch_cols<PM_Q8_0>(p, nc, pair, grid, lds, st);
    }
    return -1;
}
