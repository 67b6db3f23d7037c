// pm355_kernels.h — internal launcher prototypes (host side) for the gfx950 kernels.
#pragma once
// PM_EXPERIMENTS: round 5's three measured-slower forms of the decode layer - producer-side sums of squares, attention in the tail of the QKV launch, the persistent
// engine (decode_engine.hip) - are compiled ONLY into prima_cpp_amd/libprima_mi355_exp.so (build.py build_experiments(), -DPM_EXPERIMENTS=1), which their tests load.
// The product library libprima_mi355.so (what prima.cpp links) carries none of them: the entry points answer PM355_E_UNSUPPORTED there.
#ifndef PM_EXPERIMENTS
#define PM_EXPERIMENTS 0
#endif
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// index of the current HIP device for the per-device one-time state of the launchers (a process may drive up to 16 GPUs
// through the plug-in: hipFuncSetAttribute and the CU count are per device, not per process)
static inline int pm_cur_dev() { int d = 0; return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 16) ? d : 0; }

// one-time hipFuncSetAttribute(MaxDynamicSharedMemorySize) per (kernel, device). Keyed on the kernel POINTER: a `static bool` inside a generic
// lambda is shared by every instantiation that decays to the same function-pointer type, so only the first kernel launched would get it.
static inline void pm_allow_big_lds(const void * kern, size_t lds) {
    if (lds <= 48 * 1024) return;
    struct E { const void * k; int dev; };
    static E done[256]; static int n_done = 0;
    static hipError_t (*set)(const void *, hipFuncAttribute, int) = hipFuncSetAttribute;
    const int dev = pm_cur_dev();
    for (int i = 0; i < n_done && i < 256; ++i) if (done[i].k == kern && done[i].dev == dev) return;
    (void) set(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);       // (idempotent: a lost race sets it twice)
    if (n_done < 256) { done[n_done].k = kern; done[n_done].dev = dev; ++n_done; }
}

// bytes of one quantized ACTIVATION row in the library's internal row-SoA layout (quantize.hip)
static inline size_t pm_q8k_row_bytes(int K) { return (size_t) K + (size_t) (K / 256) * 4 + (size_t) (K / 16) * 2; }
static inline size_t pm_q80_row_bytes(int K) { return (size_t) K + (size_t) (K / 32) * 2; }

// bytes of one WEIGHT row in GGUF order (== ggml_row_size)
size_t pm_weight_row_bytes(int type, int64_t K);
// row STRIDE in HBM: == row bytes except that the trailing fp16 scale stream of the row-SoA layouts
// (Q6_K, Q8_0) is padded to 16 B so every row (and every field stream) starts 16-B aligned for any K
size_t pm_weight_row_stride(int type, int64_t K);
int pm_gemv_units_per_row(int type, int64_t K);
  // units of the mat-vec's per-(row, unit) debug partials
// HBM layout != GGUF block order (repack.hip): Q4_K (12), Q6_K (14), Q8_0 (8)
static inline bool pm_type_is_repacked(int type) { return type == 12 || type == 14 || type == 8; }

// optional second output of the Q8_K quantizers: the activation tables of the small-batch mat-mul (mmq_i8.hip: per 32-row pass, the F16
// 16-value sums in MFMA operand order | the transposed block scales) - that mat-mul then needs no prologue launch. base == null: off
// qbase (optional third table, mmq_i8.hip): the int8 values themselves in the matrix cores' A-operand order, nsb * 8192 bytes per 32 rows -
// [super-block][32-value sub-block s][operand lane = 32 * (k / 16 % 2) + row % 32][16 bytes]: one coalesced load per sub-block instead of one
// cache line per token
struct pm_q8k_tables { uint8_t * base = nullptr; size_t tab_bytes = 0; int nsb = 0; uint8_t * qbase = nullptr; };
void pm_launch_quantize_q8k(const float * x, void * y, int K, int rows, hipStream_t st, pm_q8k_tables tab = {});
void pm_launch_quantize_q80(const float * x, void * y, int K, int rows, hipStream_t st);
// silu(gate) * up -> Q8_0, written into the Q8_0 small-batch mat-mul's activation tables (pm_mmq_i8_q80_tables); then pm_launch_mmq_i8(PM_Q8_0, ..., reuse_prep = 1)
void pm_launch_silu_mul_q80_tab(const float * gate, const float * up, void * tab, size_t qtab_bytes, size_t dtab_bytes, int K, int rows, hipStream_t st);
int pm_mmq_i8_q80_tables(int K, hipStream_t st, void ** tab, size_t * qtab_bytes, size_t * dtab_bytes);
// Q8_K rows of silu(gate) * up (the ffn_down activations of a small batch), no f32 product in HBM
void pm_launch_silu_mul_q8k(const float * gate, const float * up, void * y, int K, int rows, hipStream_t st, pm_q8k_tables tab = {});
void pm_launch_rmsnorm_q8k(const float * x, const float * w, float * ynorm, void * yq, int K, int rows, float eps, hipStream_t st, void * ynorm_f16 = nullptr,
                           pm_q8k_tables tab = {});

// weight repack (row-local, bijective):  GGUF block order <-> HBM row-SoA (Q4_K, Q6_K, Q8_0); identity for the rest
void pm_launch_repack(int type, const void * src, void * dst, int64_t K, int64_t nrows, int to_device_layout, hipStream_t st);

// y[N] = W[N,K] . xq (+bias)(+resid);  optional second matrix W2 (same type/shape): y = silu(W.x) * (W2.x)
struct pm_gemv_args {
    int type; int K; int N;
    const void * W; const void * W2;
    const void * xq;            // quantized activation row(s) (Q8_K row-SoA, or Q8_0 row-SoA for Q8_0 weights)
    int ncols;                  // 1..8 activation columns (rows of xq, stride = row bytes)
    float * y; size_t y_stride; // y[col * y_stride + n]
    const float * bias;         // [N] or null
    const float * resid;        // same indexing as y, or null
    int32_t * dbg_int;          // debug: per (row, unit) {isum, msum} pairs, or null
};
int pm_launch_gemv(const pm_gemv_args & a, hipStream_t st);

// Fused single-token launch: up to 3 matrices (same K) sharing ONE activation row, which is either pre-quantized
// (xq) or given in f32 (xf) and quantized in the kernel prologue, optionally after rms_norm(xf, eps) * norm_w.
struct pm_gemv_job {
    int type; int N;
    const void * W; const void * W2;    // W2 != null: y = silu(W.x) * (W2.x) (all jobs of a launch alike)
    float * y; const float * bias; const float * resid;
};
// RoPE + KV store in the epilogue of a wq | wk | wv launch (jobs in that order). Served for NORM-mode rope when every workgroup's row
// slices start and end on even rows (pm_gemv_fused_check tells); tab = this token's cos / sin table (pm_launch_rope_table)
struct pm_qkv_epi {
    const float * tab; const int32_t * pos, * seq, * dyn; long seq_stride;
    void * kc, * vc; int Hkv, dh, n_ctx, n_rot, v_rowmajor;
    int neox;                           // rope mode 2 (build_qwen2): pairs (i, i + n_rot / 2); needs n_rot == head_dim and power-of-two slices
    // attention in the launch's tail (round 5): att_out != null - the workgroups of one KV-head group (its query rows, its K rows and its V rows all
    // live in the same grid / Hkv workgroups) take a ticket after their RoPE / KV stores; the last n_head / Hkv of them wait for the group to be
    // complete and compute one query head each over the cached cells (attn_tail_device.h: attn_cached.hip's arithmetic, the same bits) - the
    // attention launch and its boundary disappear. att_ticket: Hkv counters (zeroed once, monotonic), att_err: watchdog word (a bounded wait gave up).
    // Position-pointer mode only (pos != null, dyn == null); -7 when the shape is not served (then launch pm_launch_attn_cached as before)
    float * att_out; unsigned * att_ticket; int * att_err; float kq_scale; int n_head, att_max_keys;
};
struct pm_gemv_fused {
    int K; int njobs;
    pm_gemv_job job[3];
    const void * xq;                    // pre-quantized activation row, or null
    const float * xf; const float * norm_w; float eps;
    int32_t * dbg_int;
    const pm_qkv_epi * epi;             // null: plain outputs
    // producer-side sum of squares for the NEXT launch's rms_norm (round 5): ss_out != null (single job, no pair, no epilogue) - every workgroup
    // stores the f64 sum of the f32-rounded squares of the output rows it wrote into ss_out[workgroup] (pm_gemv_fused_grid() of them);
    // ss_in != null (with norm_w): the n_ss <= 256 partials a producer left for THIS launch's input row replace the in-kernel reduction
    double * ss_out; const double * ss_in; int n_ss;
};
int pm_launch_gemv_fused(const pm_gemv_fused & a, hipStream_t st);
int pm_gemv_fused_check(const pm_gemv_fused & a);      // the validation part of pm_launch_gemv_fused only
int pm_gemv_fused_grid(const pm_gemv_fused & a);       // workgroups pm_launch_gemv_fused would launch (= partials written through ss_out), or < 0

// batched (prefill) GEMM on MFMA: Y[T][N] = X[T][K] . W[N][K]^T (+bias[n]) (+resid[t][n]); W quantized (HBM layout), X f32
int pm_launch_gemm_q(int type, const void * W, const float * X, float * Y, int K, int N, int T, const float * bias,
                     const float * resid, hipStream_t st);

int pm_launch_gemm_q_ex(int type, const void * W, const float * X, float * Y, int K, int N, int T, const float * bias,
                        const float * resid, const float * silu_gate, int reuse_x, hipStream_t st);
int pm_launch_gemm_q_h(int type, const void * W, const float * X, const void * x_f16, float * Y, void * y_f16, int K, int N, int T,
                       const float * bias, const float * resid, const float * silu_gate, int reuse_x, hipStream_t st);

// prompt GEMM, third generation (mmq_pf.hip): up to 4 matrices (Q4_K / Q6_K, at most two types per launch) that share the F16 activations xh [T][K]
// go out as ONE launch. Job: Y[t * ldy + n] (f32) or, when Yh != null, Yh[t * ldy + n] (F16) = W . x (+ bias[n]) (+ resid[t * ldy + n]) (* silu(silu_gate[t * ldy + n]));
// ldy == 0 means N. pm_gemm_pf_check: 0 when (type, K, N, T) is served, -1 type, -2 shape
struct pm_gemm_pf_plan_t { int nt, nt_t, splitk, full, grid; double cost; };
void pm_gemm_pf_plan(int tiles, int K, int T, int cus, int force_nt, int force_s, int allow_mixed, pm_gemm_pf_plan_t * out);   // host arithmetic only (mmq_pf.hip)
struct pm_gemm_pf_job { int type; int N; const void * W; float * Y; void * Yh; const float * bias; const float * resid; const float * silu_gate; long ldy; };
int pm_gemm_pf_check(int type, int K, int N, int T);
bool pm_gemm_pf_enabled();      // false under PM355_GEMM_KERNEL=1 / 2 (A/B against the older prompt kernels)
int pm_launch_gemm_pf(const pm_gemm_pf_job * jobs, int njobs, const void * xh, int K, int T, hipStream_t st);
// pair != 0: jobs = {ffn_gate, ffn_up} (same type and N): jobs[1]'s output = silu(W0 . x) * (W1 . x) in one launch, the gate result never leaves the chip
int pm_launch_gemm_pf_ex(const pm_gemm_pf_job * jobs, int njobs, const void * xh, int K, int T, int pair, hipStream_t st);

// up to 4 matrices over the same activations (f32 X converted once, or x_f16 given): ONE launch of mmq_pf.hip when it serves all of them, else one launch each
int pm_launch_gemm_q_multi(const pm_gemm_pf_job * jobs, int njobs, const float * X, const void * x_f16, int K, int T, hipStream_t st);

int pm_launch_gemm_q_pair(const pm_gemm_pf_job * gate_up, const float * X, int K, int T, hipStream_t st);   // f32 X -> F16 scratch -> pair launch

int pm_device_cus();
struct pm_rope_cfg;

// small-batch (1..32 tokens) quantized mat-mul on the integer matrix cores (mmq_i8.hip): Q4_K / Q6_K weights, Q8_K activations
// (xq row-SoA, or x_f32 quantized first into a per-device scratch). 0, or -1 type / -2 shape / -3 device / -4 LDS
int pm_mmq_i8_check(int type, int K, int N, int T);
int pm_mmq_i8_tables(int K, hipStream_t st, pm_q8k_tables * out);      // where a quantizer writes the tables for K (<= 64 rows); then reuse_prep = 1
int pm_launch_mmq_i8_prep(const void * xq, int K, int T, hipStream_t st);      // the activation tables alone (then reuse_prep = 1)
int pm_launch_mmq_i8(int type, const void * W, const void * xq, const float * x_f32, float * Y, int K, int N, int T,
                     const float * bias, const float * resid, int reuse_prep, hipStream_t st);
// 2 or 3 matrices of one type and K that share the activations, as one launch (T <= 16); -5: not served, launch them one by one
int pm_launch_mmq_i8_multi(int type, int njobs, const void * const * W, const int * N, float * const * Y, const float * const * bias, const void * xq,
                           int K, int T, int reuse_prep, hipStream_t st, const pm_qkv_epi * epi = nullptr);
int pm_launch_mmq_i8_dual(int ta, int na, const void * const * Wa, const int * Na, float * const * Ya, const float * const * ba,
                          int tb, const void * Wb, int Nb, float * Yb, const float * bb, const void * xq, int K, int T, int reuse_prep, hipStream_t st,
                          const pm_qkv_epi * epi = nullptr);  // epi: RoPE + KV store in the launch's epilogue (NORM rope, transposed V, T <= 32)

// prompt-sized batches (>= 128 tokens) on the integer matrix cores (mmq_big.hip): Q4_K / Q6_K weights x Q8_K activations, the CPU reference's
// integer arithmetic. xq = T rows of row-SoA Q8_K, tab = the activation tables a quantizer wrote for them (pm_q8k_tables{tab, (K / 256) * 1152, K / 256};
// pm_mmq_big_table_bytes(K, T) bytes). 0, or -1 type / -2 shape / -3 device
int pm_mmq_big_check(int type, int K, int N, int T);
size_t pm_mmq_big_table_bytes(int K, int T);
int pm_launch_mmq_big(int type, const void * W, const void * xq, const void * tab, float * Y, int K, int N, int T,
                      const float * bias, const float * resid, hipStream_t st);
int pm_launch_mmq_big_f32(int type, const void * W, const float * X, float * Y, int K, int N, int T, const float * bias, const float * resid,
                          int reuse_x, hipStream_t st);
