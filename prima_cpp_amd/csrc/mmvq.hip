// mmvq.hip — batch-1 quantized mat-vec for gfx950 (the decode hot loop).
//
// Replaces, for tensors in our buffers, the reference's
//   ggml_compute_forward_mul_mat (ggml/src/ggml.c:12377) -> ggml_vec_dot_q4_K_q8_K (ggml-quants.c:7713),
//   ggml_vec_dot_q5_K_q8_K (:8281), ggml_vec_dot_q6_K_q8_K (:8918), ggml_vec_dot_q8_0_q8_0 (:5518),
// and, fused into its prologue, the activation side of that function: quantize_row_q8_K (ggml-quants.c:3785) /
// quantize_row_q8_0 (:848) preceded, when asked, by ggml_compute_forward_rms_norm_f32 (ggml.c:11950) * weight.
// Integer arithmetic is identical to the reference (int8 x int4/5/6/8 products, int32 sums, 6-bit scales/mins,
// group sums of the int8 activations for the min / -32 terms) and is checked bit-for-bit against the oracle through the
// dbg output; the float scale-and-accumulate uses a fixed, deterministic order.
//
// Design: HBM-bound, 4.5-8.5 bits per weight, no reuse of W -> W goes straight from HBM to registers (no LDS staging).
//  * One 1024-thread workgroup (16 waves) per CU. Each workgroup quantizes the activation row ONCE (from f32, fused rms_norm
//    optional, bit-exact with the reference quantizers) into LDS: int8 values, 16-group sums, block scales.
//  * Each WAVE owns whole rows (2 at a time): lanes stride over the row's UNITS (16-48 contiguous weight bytes = 32/64
//    weights), so a wave instruction covers a contiguous 1-3 KB span of the row -> fully coalesced, every byte of every
//    fetched line is used. Q4_K/Q5_K blocks are 9/11 x 16 B; Q6_K/Q8_0 are stored row-SoA (repack.hip), so every field
//    stream is 16-B aligned. 8-20 independent 16-byte non-temporal loads (global_load_dwordx4 nt) are in flight per lane;
//    16 waves per CU hide the HBM latency. There is NO barrier and NO cross-wave reduction in the row loop.
//  * All weight loads are unconditional (clamped addresses) and the loop body contains no other VMEM operation (row
//    results are parked in LDS and written out coalesced afterwards): gfx9 counts loads and stores in ONE vmcnt and a
//    load in a divergent branch or a store in the stream makes the compiler fall back to s_waitcnt vmcnt(0).
//  * v_dot4_i32_i8 for the products against the LDS-resident activation, DPP reduction per row.
//  * Up to 3 weight matrices that share the activation (wq/wk/wv) are served by ONE launch ("jobs"), of at most two
//    different quant types (Q4_K_M: attn_v is Q6_K/Q5_K next to Q4_K q/k); every workgroup takes an equal slice of
//    the rows of every job.
#include "mmvq_device.h"
#include <stdlib.h>

using namespace pmv;

namespace {

// The first things a workgroup does - the activation loads and the pre-issued weight loads of job 0 - need six values of the argument block. They
// travel as LEADING scalar kernel arguments as well, which the dispatcher preloads into SGPRs (-mllvm -amdgpu-kernarg-preload-count, build.py):
// the wave starts issuing its loads without first waiting for an s_load of the kernarg segment (a by-value struct is never preloaded).
// FEAT (mmvq_device.h): PM_FEAT_SS / PM_FEAT_TAIL code lives in instantiations of its own - a launch that uses neither runs the plain kernel
// XM >= 0: the activation mode is a compile-time constant of the instantiation (the hot launches: 1 = f32 row, 2 = rms_norm of an f32 row) - the prologue
// keeps one mode's code instead of four behind scalar branches
// HOT: everything else the decode launches of the Llama-family layer never use is pinned too (no bias, one job outside wq | wk | wv, no residual on
// the pair launch, transposed V cache and - HOT 1: the engine - position-pointer mode / HOT 2: the plug-in - ggml-graph mode in the QKV epilogue):
// launch_types() checks that the argument block really looks like that
template <int TA, int TB, bool PAIR, bool DBG, bool EPI = false, int NPRE = 2, int FEAT = 0, int XM = -1, int HOT = 0>
#ifdef PM_NO_PRELOAD_ARGS
__global__ __launch_bounds__(PM_GEMV_BLOCK, 4) void gemv_q_kernel(GemvP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double nred[PM_GEMV_NW];
#else
__global__ __launch_bounds__(PM_GEMV_BLOCK, 4) void gemv_q_kernel(const float * xf, const float * norm_w, const uint8_t * W0, const uint8_t * W0b, long row_stride0,
                                                                  int K, int xmode, int N0, int U0, GemvP p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double nred[PM_GEMV_NW];
    GemvP p = p_in;
    p.xf = xf; p.norm_w = norm_w; p.K = K; p.xmode = xmode;
    p.job[0].W = W0; p.job[0].W2 = W0b; p.job[0].row_stride = row_stride0; p.job[0].N = N0; p.job[0].U = U0;
#endif
    if constexpr (XM >= 0) p.xmode = XM;
    if constexpr (HOT) {
        p.dbg = nullptr; p.job[0].bias = nullptr;
        if constexpr (EPI) {
            p.job[1].bias = nullptr; p.job[2].bias = nullptr; p.epi.v_rowmajor = 0;
            if constexpr (HOT == 1) p.epi.dyn = nullptr; else __builtin_assume(p.epi.dyn != nullptr);
            p.job[0].role = 1; p.job[1].role = 2; p.job[2].role = 3;                    // wq, wk, wv in this order (no TA-first swap happened)
            p.job[0].is_b = 0; p.job[1].is_b = 0; p.job[2].is_b = TA != TB ? 1 : 0;     // Q4_K_M: wq / wk of the first type, wv of the second (or all of one)
            p.job[0].split = 0; p.job[0].resid = nullptr; p.job[1].resid = nullptr; p.job[2].resid = nullptr;
        } else { p.job[1].N = 0; p.job[2].N = 0; p.job[0].split = 0; p.job[0].is_b = 0; p.job[0].role = 0; }
        if constexpr (PAIR) p.job[0].resid = nullptr;
        if constexpr (!PAIR && !EPI) __builtin_assume(p.job[0].resid != nullptr);     // wo / ffn_down: always + residual
    }
    // (Q5_K: two 32-weight units per lane and step = 24-VGPR sets with the high-bit plane; a second pre-issued set spills)
    gemv_body<TA, TB, PAIR, DBG, 1, EPI, (TA == PM_Q5_K && NPRE == 2) ? 1 : NPRE, FEAT>(p, smem, nred);
}
template <int TA, int TB>
int launch_types(const GemvP & p_in, bool pair, int grid, size_t lds, bool dbg, hipStream_t st, bool epi = false) {
    const GemvP & p = p_in;
    auto go = [&](auto kern) {
        pm_allow_big_lds((const void *) kern, lds);
#ifdef PM_NO_PRELOAD_ARGS
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PM_GEMV_BLOCK), lds, st, p);
#else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PM_GEMV_BLOCK), lds, st, p.xf, p.norm_w, p.job[0].W, p.job[0].W2, p.job[0].row_stride, p.K, p.xmode, p.job[0].N, p.job[0].U, p);
#endif
    };
    const bool ss = p.ss_out != nullptr || p.xmode == 3;          // producer or consumer of per-workgroup partial sums of squares
#if !PM_EXPERIMENTS
    if (ss || (epi && p.epi.att_out != nullptr)) return -8;         // (sum-of-squares partials / attention tail: experiments library only)
#endif
    static const bool xm_spec = [] { const char * e = getenv("PM355_XMODE_SPEC"); return !(e && e[0] == '0'); }();     // (A/B switches)
    static const bool hot_spec = xm_spec && [] { const char * e = getenv("PM355_HOT_SPEC"); return !(e && e[0] == '0'); }();
    const bool one_job = p.job[1].N == 0 && p.job[2].N == 0;
    const bool tail = epi && p.epi.att_out != nullptr; (void) tail;
    const bool neox = epi && (p.job[0].nx_s > 0 || p.job[1].nx_s > 0 || p.job[2].nx_s > 0);
    if (pair) {
        if (TA != TB) return -1;
        // (pair launches: one step of pre-issue - two sets of two matrices next to the activation registers spill)
        if (dbg) { if (ss) return -6; go(gemv_q_kernel<TA, TA, true, true, false, 1>); }
#if PM_EXPERIMENTS
        else if (ss) go(gemv_q_kernel<TA, TA, true, false, false, 1, PM_FEAT_SS>);
#endif
        else if (p.xmode == 2 && hot_spec && one_job && !p.job[0].bias && !p.job[0].resid && !p.job[0].split) go(gemv_q_kernel<TA, TA, true, false, false, 1, 0, 2, 1>);
        else if (p.xmode == 2 && xm_spec) go(gemv_q_kernel<TA, TA, true, false, false, 1, 0, 2>);
        else go(gemv_q_kernel<TA, TA, true, false, false, 1>);
    } else if (epi) {
        if (dbg) return -1;
#if PM_EXPERIMENTS
        if (tail) {
            // the tail is compiled for the wq | wk | wv type mixtures of the Q4_K_M files only (wq Q4_K; wv Q4_K / Q5_K / Q6_K)
            if constexpr (TA == PM_Q4_K) { if (ss) go(gemv_q_kernel<TA, TB, false, false, true, 2, PM_FEAT_SS | PM_FEAT_TAIL>); else go(gemv_q_kernel<TA, TB, false, false, true, 2, PM_FEAT_TAIL>); }
            else return -7;
        } else if (neox && ss) go(gemv_q_kernel<TA, TB, false, false, true, 2, PM_FEAT_NEOX | PM_FEAT_SS>);
        else if (ss) go(gemv_q_kernel<TA, TB, false, false, true, 2, PM_FEAT_SS>);
        else
#endif
        if (neox) go(gemv_q_kernel<TA, TB, false, false, true, 2, PM_FEAT_NEOX>);
        else if (p.xmode == 2 && hot_spec && !p.job[0].bias && !p.job[1].bias && !p.job[2].bias && !p.epi.v_rowmajor &&
                 p.job[0].role == 1 && p.job[1].role == 2 && p.job[2].role == 3 && !p.job[0].is_b && !p.job[1].is_b && p.job[2].is_b == (TA != TB ? 1 : 0) &&
                 !p.job[0].split && !p.job[0].resid && !p.job[1].resid && !p.job[2].resid) { if (p.epi.dyn) go(gemv_q_kernel<TA, TB, false, false, true, 2, 0, 2, 2>); else go(gemv_q_kernel<TA, TB, false, false, true, 2, 0, 2, 1>); }
        else if (p.xmode == 2 && xm_spec) go(gemv_q_kernel<TA, TB, false, false, true, 2, 0, 2>);
        else go(gemv_q_kernel<TA, TB, false, false, true>);
    } else {
        if (dbg) { if (ss) return -6; go(gemv_q_kernel<TA, TB, false, true, false, 1>); }     // (test hook: one pre-issued step - with two, the Q6_K form sat on the 128-VGPR cliff with a spilled register)
#if PM_EXPERIMENTS
        else if (ss) go(gemv_q_kernel<TA, TB, false, false, false, 2, PM_FEAT_SS>);
#endif
        else if (p.xmode == 1 && hot_spec && one_job && !p.job[0].bias && p.job[0].resid && !p.job[0].split && !p.job[0].is_b) go(gemv_q_kernel<TA, TB, false, false, false, 2, 0, 1, 1>);
        else if (p.xmode == 1 && xm_spec) go(gemv_q_kernel<TA, TB, false, false, false, 2, 0, 1>);
        else if (p.xmode == 2 && xm_spec) go(gemv_q_kernel<TA, TB, false, false, false, 2, 0, 2>);
        else go(gemv_q_kernel<TA, TB, false, false>);
    }
    return 0;
}

int nv_of(int type) { return type == PM_Q4_K ? QT<PM_Q4_K>::NV : type == PM_Q6_K ? 64 : 32; }

bool type_ok(int t) { return t == PM_Q4_K || t == PM_Q5_K || t == PM_Q6_K || t == PM_Q8_0; }

} // namespace

static int g_num_cus_dev[16] = {};
static int num_cus() {
    const int dv = pm_cur_dev();
    if (!g_num_cus_dev[dv]) { hipDeviceProp_t pr; g_num_cus_dev[dv] = hipGetDeviceProperties(&pr, dv) == hipSuccess ? pr.multiProcessorCount : 256; }
    return g_num_cus_dev[dv];
}

int pm_gemv_units_per_row(int type, int64_t K) { return (int) (K / nv_of(type)); }

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
        case PM_Q6_K: return pm_q6k_row_stride((size_t) (K / 256));
    }
    return pm_weight_row_bytes(type, K);
}

// Validates a fused job list and fills the kernel argument block. ta <= tb are the (at most two) quant types in the
// canonical order of the instantiated kernels: (Q4_K,Q4_K) (Q5_K,Q5_K) (Q6_K,Q6_K) (Q8_0,Q8_0) (Q4_K,Q6_K) (Q4_K,Q5_K); job 0 is
// always a `ta` job (the kernel pre-issues it). grid_fixed > 0: the caller's grid (attn_wo.hip), else chosen here.
int pmv::gemv_fill(const pm_gemv_fused & a, int grid_fixed, GemvP & p, int & ta_out, int & tb_out, bool & pair_out, size_t & lds_out, int & grid_out) {
    if (a.njobs < 1 || a.njobs > 3) return -4;
    p = GemvP{};
    p.K = a.K; p.xq = (const uint8_t *) a.xq; p.xf = a.xf; p.norm_w = a.norm_w; p.eps = a.eps; p.dbg = a.dbg_int;
    p.xmode = a.xq ? 0 : (a.norm_w ? (a.ss_in ? 3 : 2) : 1);
    if (a.ss_in && (!a.norm_w || a.xq || a.n_ss < 1 || a.n_ss > 256)) return -6;
    if (a.ss_out && (a.njobs != 1 || a.job[0].W2 || a.epi || a.dbg_int)) return -6;
    p.ss_out = a.ss_out; p.ss_in = a.ss_in; p.n_ss = a.n_ss;
    int ta = a.job[0].type, tb = ta;
    const bool pair = a.job[0].W2 != nullptr;
    for (int j = 0; j < a.njobs; ++j) {
        const int t = a.job[j].type;
        if (!type_ok(t)) return -1;
        if (t != ta) { if (tb == ta) tb = t; else if (t != tb) return -1; }
        if ((a.job[j].W2 != nullptr) != pair) return -1;
    }
    // a Q8_0 job needs Q8_0-quantized activations, the K-quants need Q8_K: one prologue serves only one family
    if ((ta == PM_Q8_0) != (tb == PM_Q8_0)) return -1;
    if (ta == PM_Q8_0 ? a.K % 32 : a.K % 256) return -2;
    if (pair && ta != tb) return -1;
    if (ta != tb) {
        if (ta != PM_Q4_K) { const int t = ta; ta = tb; tb = t; }     // canonical order: Q4_K first
        if (ta != PM_Q4_K || (tb != PM_Q6_K && tb != PM_Q5_K)) return -1;
    }
    // one workgroup of 16 waves per CU; every workgroup takes an equal slice of the rows of EVERY job
    int grid = grid_fixed > 0 ? grid_fixed : (1024 / PM_GEMV_BLOCK) * num_cus();
    long tot_rows = 0;
    for (int j = 0; j < a.njobs; ++j) tot_rows += a.job[j].N;
    int rows_cap = PM_MAX_ROWS_PER_WG;
    if (grid_fixed > 0) rows_cap = (int) ((tot_rows + grid - 1) / grid + 3 > PM_MAX_ROWS_PER_WG ? (tot_rows + grid - 1) / grid + 4 : PM_MAX_ROWS_PER_WG);
    else while ((tot_rows + grid - 1) / grid + 3 > PM_MAX_ROWS_PER_WG) grid *= 2;
    for (int j = 0; j < 3; ++j) {
        GemvJob & g = p.job[j];
        if (j >= a.njobs) { g.N = 0; continue; }
        const pm_gemv_job & s = a.job[j];
        g.W = (const uint8_t *) s.W; g.W2 = (const uint8_t *) s.W2; g.y = s.y; g.bias = s.bias; g.resid = s.resid;
        g.N = s.N; g.is_b = (s.type != ta); g.U = a.K / nv_of(s.type);
        g.row_stride = (long) pm_weight_row_stride(s.type, a.K);
        g.role = 0; g.split = 0;
        if (a.epi) {
            // wq | wk | wv with the RoPE + KV-store epilogue: every workgroup's slice must hold whole rotation pairs; a matrix that gives a
            // workgroup fewer rows than it has waves is dealt out step by step (row, chunk) so that all 16 waves share its rows
            if (pair || a.njobs != 3 || s.N % (2 * grid) || a.dbg_int) return -5;
            g.role = j + 1;
            const int upl = (g.U + 63) >> 6, ch = nv_of(s.type) == 64 ? PM_CH64 : PM_CH32, cpr = (upl + ch - 1) / ch;
            g.split = (j > 0 && s.N / grid < PM_GEMV_NW && cpr > 1) ? 1 : 0;
            if (a.epi->neox && j < 2) {
                // NEOX pairs (i, i + n_rot / 2): the slice of every workgroup = two runs of sl / 2 rows, n_rot / 2 apart, inside one head
                const int sl = s.N / grid, dh = a.epi->dh;
                auto pow2 = [](int v) { return v > 0 && !(v & (v - 1)); };
                if (a.epi->n_rot != dh || !pow2(dh) || !pow2(sl) || sl < 2 || sl > dh) return -5;
                int ls = 0, ld = 0;
                while ((1 << ls) < sl) ++ls;
                while ((1 << ld) < dh) ++ld;
                g.nx_s = ls; g.nx_dh = ld; g.nx_hrot = a.epi->n_rot / 2;
            }
        }
    }
    size_t att_lds = 0;
    if (a.epi) {
        const pm_qkv_epi & e = *a.epi;
        if (grid_fixed > 0 || e.dh % 2 || e.n_rot % 2 || e.n_rot > e.dh || a.job[0].N % e.dh || a.job[1].N != e.Hkv * e.dh || a.job[2].N != e.Hkv * e.dh) return -5;
        // (ggml-graph mode has no position pointer: the scalar loads of the un-taken branch may still be issued - give them a valid address)
        p.epi = QkvEpi{e.tab, e.pos ? e.pos : e.dyn, e.seq, e.dyn, e.seq_stride, (uint16_t *) e.kc, (uint16_t *) e.vc, e.Hkv * e.dh, e.dh, e.n_ctx, e.n_rot, e.v_rowmajor, e.neox,
                       nullptr, nullptr, nullptr, nullptr, 0.0f, 0, 0};
        if (e.att_out) {
            // attention in the tail: the workgroups of a KV-head group must be a power-of-two run of the grid, at least one per query head of the group
            const int wgs = e.Hkv > 0 && grid % e.Hkv == 0 ? grid / e.Hkv : 0, nh = e.Hkv > 0 ? e.n_head / e.Hkv : 0;
            const int mk = e.att_max_keys > 0 && e.att_max_keys < e.n_ctx ? e.att_max_keys : e.n_ctx;
            if (!e.pos || e.dyn || !e.att_ticket || e.v_rowmajor || wgs < 1 || (wgs & (wgs - 1)) || nh < 1 || nh > wgs || e.n_head != nh * e.Hkv || a.job[0].N != e.n_head * e.dh ||
                (e.dh != 64 && e.dh != 128) || e.n_ctx % 8 || e.n_ctx < 64) return -7;
            att_lds = 64 + attn_tail_lds(e.dh, mk);
            p.epi.att_out = e.att_out; p.epi.att_q = a.job[0].y; p.epi.att_ticket = e.att_ticket; p.epi.att_err = e.att_err; p.epi.att_scale = e.kq_scale;
            p.epi.att_H = e.n_head; p.epi.att_wgs = wgs;
        }
    }
    if (p.job[0].is_b)                                // the kernel pre-issues job 0 with the TA code path: put a TA job first
        for (int j = 1; j < 3; ++j) if (p.job[j].N > 0 && !p.job[j].is_b) { const GemvJob t = p.job[0]; p.job[0] = p.job[j]; p.job[j] = t; break; }
    const int ablk = ta == PM_Q8_0 ? 32 : 256;
    size_t lds = (size_t) ((a.K + 15) & ~15) + (size_t) (a.K / 16) * 4 + (size_t) ((a.K / ablk + 3) & ~3) * 4 + (size_t) rows_cap * 4;
    if (att_lds > lds) lds = att_lds;
    if (lds > 150 * 1024) return -4;
    ta_out = ta; tb_out = tb; pair_out = pair; lds_out = lds; grid_out = grid;
    return 0;
}

int pm_device_cus() { return num_cus(); }

int pm_gemv_fused_check(const pm_gemv_fused & a) {
    GemvP p; int ta, tb, grid; bool pair; size_t lds;
    return gemv_fill(a, 0, p, ta, tb, pair, lds, grid);
}

int pm_gemv_fused_grid(const pm_gemv_fused & a) {
    GemvP p; int ta, tb, grid; bool pair; size_t lds;
    const int rc = gemv_fill(a, 0, p, ta, tb, pair, lds, grid);
    return rc ? (rc < 0 ? rc : -rc) : grid;
}

// Fused launch: up to 3 matrices sharing one activation row.
int pm_launch_gemv_fused(const pm_gemv_fused & a, hipStream_t st) {
    GemvP p; int ta, tb, grid; bool pair; size_t lds;
    const int rc = gemv_fill(a, 0, p, ta, tb, pair, lds, grid);
    if (rc) return rc;
    p.ts = pm_ts_next_slot();
#define PM_L(TA_, TB_) return launch_types<TA_, TB_>(p, pair, grid, lds, a.dbg_int != nullptr, st, a.epi != nullptr)
    if (ta == PM_Q4_K && tb == PM_Q4_K) PM_L(PM_Q4_K, PM_Q4_K);
    if (ta == PM_Q5_K && tb == PM_Q5_K) PM_L(PM_Q5_K, PM_Q5_K);
    if (ta == PM_Q6_K && tb == PM_Q6_K) PM_L(PM_Q6_K, PM_Q6_K);
    if (ta == PM_Q8_0 && tb == PM_Q8_0) PM_L(PM_Q8_0, PM_Q8_0);
    if (ta == PM_Q4_K && tb == PM_Q6_K) PM_L(PM_Q4_K, PM_Q6_K);
    if (ta == PM_Q4_K && tb == PM_Q5_K) PM_L(PM_Q4_K, PM_Q5_K);
#undef PM_L
    return -1;
}

// Single-matrix entry with ncols pre-quantized activation columns: groups of 8 / 4 / 2 columns share one pass over the
// weights (gemv_q_cols_kernel), a remaining single column - and pair / debug launches - go one launch per column.
int pm_launch_gemv(const pm_gemv_args & a, hipStream_t st) {
    const size_t xrow = a.type == PM_Q8_0 ? pm_q80_row_bytes(a.K) : pm_q8k_row_bytes(a.K);
    int c = 0;
    while (c < a.ncols) {
        pm_gemv_fused f = {};
        f.K = a.K; f.njobs = 1; f.xq = (const uint8_t *) a.xq + (size_t) c * xrow; f.dbg_int = a.dbg_int;
        f.job[0].type = a.type; f.job[0].N = a.N; f.job[0].W = a.W; f.job[0].W2 = a.W2;
        f.job[0].y = a.y + (size_t) c * a.y_stride; f.job[0].bias = a.bias;
        f.job[0].resid = a.resid ? a.resid + (size_t) c * a.y_stride : nullptr;
        const int left = a.ncols - c;
        const bool pair = a.W2 != nullptr;
        // column slots of the launch: 8 / 4 / 2 (3 columns take the 4-slot form with one slot idle: one pass instead of 2 + 1); pair launches: 2
        int nc = (a.dbg_int || left < 2) ? 1 : pair ? ((a.type == PM_Q4_K || a.type == PM_Q6_K) ? 2 : 1) : (left >= 8 && a.type != PM_Q5_K ? 8 : left >= 3 ? 4 : 2);   // (8 Q5_K columns spill even at 256 VGPRs)
        if (nc > 1) {
            GemvP p; int ta, tb, grid; bool pr; size_t lds1;
            int rc = gemv_fill(f, 0, p, ta, tb, pr, lds1, grid);
            if (rc) return rc;
            // LDS: nc activation columns + nc results per row
            const int ablk = ta == PM_Q8_0 ? 32 : 256;
            const size_t col = (size_t) ((a.K + 15) & ~15) + (size_t) (a.K / 16) * 4 + (size_t) ((a.K / ablk + 3) & ~3) * 4;
            const size_t rows = (size_t) ((a.N + grid - 1) / grid + 4);
            while (nc > 1 && nc * col + rows * nc * 4 > 150 * 1024) nc /= 2;
            if (nc > 1) {
                const int served = left < nc ? left : nc;
                p.ncols = served; p.xq_stride = (long) xrow; p.y_stride = (long) a.y_stride;
                const size_t lds = nc * col + rows * nc * 4;
                rc = pm_launch_gemv_cols(ta, p, nc, pair, grid, lds, st);      // mmvq_cols.hip (512-thread workgroups, 256-VGPR budget)
                if (rc) return rc;
                c += served;
                continue;
            }
        }
        const int rc = pm_launch_gemv_fused(f, st);
        if (rc) return rc;
        c += 1;
    }
    return 0;
}
