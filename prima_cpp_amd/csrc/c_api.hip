// c_api.hip — extern "C" surface of libprima_mi355.so (see include/prima_mi355.h).
#include "../../include/prima_mi355.h"
#include "pm355_device.h"
#include "pm355_kernels.h"
#include "pm355_engine.h"
#include "pm355_layer_ops.h"
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>

static thread_local char g_err[256] = "";
static int fail(int code, const char * what, hipError_t e = hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
    return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(PM355_E_HIP, #expr, e_); } while (0)
static inline hipStream_t S(pm355_stream_t s) { return (hipStream_t) s; }

extern "C" {

const char * pm355_version(void) { return "prima_mi355 0.1 (gfx950)"; }
const char * pm355_last_error(void) { return g_err; }

int pm355_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
int pm355_set_device(int d) { HIP_TRY(hipSetDevice(d)); return 0; }
int pm355_device_info(int d, char * name, size_t name_len, size_t * free_b, size_t * total_b, int * cus) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, d));
    if (name && name_len) { strncpy(name, p.name, name_len - 1); name[name_len - 1] = 0; }
    if (cus) *cus = p.multiProcessorCount;
    if (free_b || total_b) {
        int cur = 0; hipGetDevice(&cur); hipSetDevice(d);
        size_t f = 0, t = 0; hipError_t e = hipMemGetInfo(&f, &t); hipSetDevice(cur);
        if (e != hipSuccess) return fail(PM355_E_HIP, "hipMemGetInfo", e);
        if (free_b) *free_b = f;
        if (total_b) *total_b = t;
    }
    return 0;
}
// Is `p` page-locked host memory? Asked for every small upload of a token (positions, mask, ...). Only POSITIVE verdicts are cached (per 4 KiB page,
// direct-mapped, flushed whenever this library allocates or frees pinned memory itself - pm355_host_malloc / pm355_host_free, the plug-in's host
// buffer type): a stale "pinned" costs one wait, but a cached "pageable" for a page somebody pins later (hipHostRegister, another library's pinned
// allocator) would let set_tensor return while the DMA still reads the source - pageable pages ask the runtime every time.
namespace { struct PinE { uintptr_t page; }; PinE g_pin[256] = {}; std::atomic<unsigned> g_pin_gen{1}; unsigned g_pin_seen = 0; std::mutex g_pin_mu; }
int pm355_host_is_pinned(const void * p) {
    const uintptr_t page = ((uintptr_t) p >> 12) + 1;                 // (+1: 0 marks an empty slot)
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        const unsigned gen = g_pin_gen.load(std::memory_order_acquire);
        if (gen != g_pin_seen) { memset(g_pin, 0, sizeof(g_pin)); g_pin_seen = gen; }
        if (g_pin[page & 255].page == page) return 1;
    }
    const unsigned gen0 = g_pin_gen.load(std::memory_order_acquire);
    hipPointerAttribute_t a;
    int pinned = 0;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) (void) hipGetLastError();                   // unknown to the runtime: pageable
    else pinned = a.type == hipMemoryTypeHost ? 1 : 0;
    if (pinned) {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (g_pin_gen.load(std::memory_order_acquire) == gen0 && g_pin_seen == gen0) g_pin[page & 255].page = page;   // (an allocation raced the query: do not cache)
    }
    return pinned;
}
int pm355_sync_null_stream(void) { HIP_TRY(hipStreamSynchronize(nullptr)); return 0; }
int pm355_sync(pm355_stream_t s) { HIP_TRY(s ? hipStreamSynchronize(S(s)) : hipDeviceSynchronize()); return 0; }

pm355_stream_t pm355_stream_create(void) { hipStream_t s = nullptr; return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? (pm355_stream_t) s : nullptr; }
void pm355_stream_destroy(pm355_stream_t s) { if (s) (void) hipStreamDestroy(S(s)); }
pm355_event_t pm355_event_create(void) { hipEvent_t e = nullptr; return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? (pm355_event_t) e : nullptr; }
void pm355_event_destroy(pm355_event_t e) { if (e) (void) hipEventDestroy((hipEvent_t) e); }
int pm355_event_record(pm355_event_t e, pm355_stream_t s) { HIP_TRY(hipEventRecord((hipEvent_t) e, S(s))); return 0; }
int pm355_event_wait(pm355_stream_t s, pm355_event_t e) { HIP_TRY(hipStreamWaitEvent(S(s), (hipEvent_t) e, 0)); return 0; }
int pm355_event_sync(pm355_event_t e) { HIP_TRY(hipEventSynchronize((hipEvent_t) e)); return 0; }

void * pm355_malloc(size_t n) { void * p = nullptr; return hipMalloc(&p, n) == hipSuccess ? p : nullptr; }
void   pm355_free(void * p) { if (p) (void) hipFree(p); }
// (the generation moves AFTER the runtime has changed the status of the pages: a query that overlaps the call re-asks the runtime next time)
void * pm355_host_malloc(size_t n) { void * p = nullptr; const bool ok = hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess; g_pin_gen.fetch_add(1, std::memory_order_release); return ok ? p : nullptr; }
void   pm355_host_free(void * p) { if (p) (void) hipHostFree(p); g_pin_gen.fetch_add(1, std::memory_order_release); }
int pm355_memcpy_h2d(void * d, const void * s, size_t n, pm355_stream_t st) { HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, S(st))); return 0; }
int pm355_memcpy_d2h(void * d, const void * s, size_t n, pm355_stream_t st) { HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, S(st))); return 0; }
int pm355_memcpy_d2d(void * d, const void * s, size_t n, pm355_stream_t st) { HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, S(st))); return 0; }
int pm355_memset(void * d, int v, size_t n, pm355_stream_t st) { HIP_TRY(hipMemsetAsync(d, v, n, S(st))); return 0; }

size_t pm355_row_size(int type, int64_t K) { return pm_weight_row_bytes(type, K); }
size_t pm355_row_stride(int type, int64_t K) { return pm_weight_row_stride(type, K); }
size_t pm355_q8_K_row_size(int64_t K) { return pm_q8k_row_bytes((int) K); }
size_t pm355_q8_0_row_size(int64_t K) { return pm_q80_row_bytes((int) K); }

int pm355_repack_rows(int type, const void * src, void * dst, int64_t K, int64_t nrows, int to_dev, pm355_stream_t st) {
    if (pm_weight_row_bytes(type, K) == 0) return fail(PM355_E_UNSUPPORTED, "repack: type");
    pm_launch_repack(type, src, dst, K, nrows, to_dev, S(st));
    HIP_TRY(hipGetLastError());
    return 0;
}

int pm355_quantize_q8_K(const float * x, void * yq, int64_t K, int64_t rows, pm355_stream_t st) {
    if (K <= 0 || K % 256) return fail(PM355_E_SHAPE, "quantize_q8_K: K % 256");
    pm_launch_quantize_q8k(x, yq, (int) K, (int) rows, S(st));
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_quantize_q8_0(const float * x, void * yq, int64_t K, int64_t rows, pm355_stream_t st) {
    if (K <= 0 || K % 32) return fail(PM355_E_SHAPE, "quantize_q8_0: K % 32");
    pm_launch_quantize_q80(x, yq, (int) K, (int) rows, S(st));
    HIP_TRY(hipGetLastError());
    return 0;
}

// row-SoA activation -> reference block layout (test hook; tiny)
__global__ void act_to_blocks_kernel(int act_type, const uint8_t * yq, uint8_t * out, int K, int rows, size_t in_row) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (act_type == PM_Q8_K) {
        const int nblk = K / 256;
        if (i >= (long) rows * nblk) return;
        const int row = (int) (i / nblk), b = (int) (i % nblk);
        const uint8_t * r = yq + (size_t) row * in_row;
        uint8_t * o = out + ((size_t) row * nblk + b) * PM_BS_Q8_K;
        *(float *) o = ((const float *) (r + K))[b];
        for (int j = 0; j < 256; ++j) o[4 + j] = r[b * 256 + j];
        for (int j = 0; j < 16; ++j) ((int16_t *) (o + 260))[j] = ((const int16_t *) (r + K + nblk * 4))[b * 16 + j];
    } else {
        const int nblk = K / 32;
        if (i >= (long) rows * nblk) return;
        const int row = (int) (i / nblk), b = (int) (i % nblk);
        const uint8_t * r = yq + (size_t) row * in_row;
        uint8_t * o = out + ((size_t) row * nblk + b) * PM_BS_Q8_0;
        *(uint16_t *) o = ((const uint16_t *) (r + K))[b];
        for (int j = 0; j < 32; ++j) o[2 + j] = r[b * 32 + j];
    }
}
int pm355_act_to_ggml_blocks(int act_type, const void * yq, void * out, int64_t K, int64_t rows, pm355_stream_t st) {
    if (act_type != PM_Q8_K && act_type != PM_Q8_0) return fail(PM355_E_UNSUPPORTED, "act_to_ggml_blocks: type");
    const long n = rows * (act_type == PM_Q8_K ? K / 256 : K / 32);
    const size_t in_row = act_type == PM_Q8_K ? pm_q8k_row_bytes((int) K) : pm_q80_row_bytes((int) K);
    hipLaunchKernelGGL(act_to_blocks_kernel, dim3((unsigned) ((n + 63) / 64)), dim3(64), 0, S(st),
                       act_type, (const uint8_t *) yq, (uint8_t *) out, (int) K, (int) rows, in_row);
    HIP_TRY(hipGetLastError());
    return 0;
}

int pm355_rms_norm(const float * x, const float * w, float * y, void * yq, int64_t K, int64_t rows, float eps, pm355_stream_t st) {
    if (K <= 0 || K % 256) return fail(PM355_E_SHAPE, "rms_norm: K % 256");
    pm_launch_rmsnorm_q8k(x, w, y, yq, (int) K, (int) rows, eps, S(st));
    HIP_TRY(hipGetLastError());
    return 0;
}

static int gemv_rc(int rc) {
    switch (rc) {
        case 0: return 0;
        case -1: return fail(PM355_E_UNSUPPORTED, "mul_mat_vec_q: weight type");
        case -2: return fail(PM355_E_SHAPE, "mul_mat_vec_q: K not a multiple of the block size");
        case -3: return fail(PM355_E_ALIGN, "mul_mat_vec_q: K not a multiple of 32 for Q8_0");
        case -4: return fail(PM355_E_RANGE, "mul_mat_vec_q: K too large");
        case -8: return fail(PM355_E_UNSUPPORTED, "mul_mat_vec_q: sum-of-squares partials / attention tail are built into libprima_mi355_exp.so only (PM_EXPERIMENTS)");
        default: return fail(PM355_E_HIP, "mul_mat_vec_q: launch", hipGetLastError());
    }
}
int pm355_mul_mat_vec_q(int type, const void * W, const void * W2, int64_t K, int64_t N, const void * xq, int ncols,
                        float * y, int64_t y_stride, const float * bias, const float * resid, pm355_stream_t st) {
    if (ncols < 1 || ncols > 8) return fail(PM355_E_RANGE, "mul_mat_vec_q: ncols must be 1..8");
    pm_gemv_args a = {};
    a.type = type; a.K = (int) K; a.N = (int) N; a.W = W; a.W2 = W2; a.xq = xq; a.ncols = ncols;
    a.y = y; a.y_stride = (size_t) y_stride; a.bias = bias; a.resid = resid; a.dbg_int = nullptr;
    return gemv_rc(pm_launch_gemv(a, S(st)));
}
int pm355_mul_mat_vec_fused(const pm355_matvec_job * jobs, int njobs, int64_t K, const float * x_f32, const float * norm_w,
                            float eps, pm355_stream_t st) {
    if (njobs < 1 || njobs > 3 || !jobs || !x_f32) return fail(PM355_E_RANGE, "mul_mat_vec_fused: 1..3 jobs and an f32 activation");
    pm_gemv_fused f = {};
    f.K = (int) K; f.njobs = njobs; f.xf = x_f32; f.norm_w = norm_w; f.eps = eps;
    for (int j = 0; j < njobs; ++j) {
        f.job[j].type = jobs[j].type; f.job[j].N = (int) jobs[j].N; f.job[j].W = jobs[j].W; f.job[j].W2 = jobs[j].W2;
        f.job[j].y = jobs[j].y; f.job[j].bias = jobs[j].bias; f.job[j].resid = jobs[j].resid;
    }
    (void) hipGetLastError();
    const int rc = gemv_rc(pm_launch_gemv_fused(f, S(st)));
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_mul_mat_vec_fused_ss(const pm355_matvec_job * jobs, int njobs, int64_t K, const float * x_f32, const float * norm_w, float eps,
                               double * sumsq_out, const double * sumsq_in, int n_sumsq_in, pm355_stream_t st) {
    if (njobs < 1 || njobs > 3 || !jobs || !x_f32) return fail(PM355_E_RANGE, "mul_mat_vec_fused_ss: 1..3 jobs and an f32 activation");
    pm_gemv_fused f = {};
    f.K = (int) K; f.njobs = njobs; f.xf = x_f32; f.norm_w = norm_w; f.eps = eps;
    f.ss_out = sumsq_out; f.ss_in = sumsq_in; f.n_ss = n_sumsq_in;
    for (int j = 0; j < njobs; ++j) {
        f.job[j].type = jobs[j].type; f.job[j].N = (int) jobs[j].N; f.job[j].W = jobs[j].W; f.job[j].W2 = jobs[j].W2;
        f.job[j].y = jobs[j].y; f.job[j].bias = jobs[j].bias; f.job[j].resid = jobs[j].resid;
    }
    (void) hipGetLastError();
    const int lrc = pm_launch_gemv_fused(f, S(st));
    if (lrc == -8) return gemv_rc(-8);
    if (lrc == -6) return fail(PM355_E_UNSUPPORTED, "mul_mat_vec_fused_ss: sumsq_out needs ONE plain job; sumsq_in needs norm_w and 1..256 partials");
    const int rc = gemv_rc(lrc);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
namespace {
struct ScatterP { pm355_scatter_seg seg[16]; };
__global__ __launch_bounds__(256) void scatter_bytes_kernel(const uint8_t * __restrict__ src, ScatterP p) {
    const pm355_scatter_seg sg = p.seg[blockIdx.x];
    const uint8_t * s_ = src + sg.src_off;
    uint8_t * d_ = (uint8_t *) sg.dst;
    if ((((uintptr_t) d_ | (uintptr_t) s_ | sg.bytes) & 3) == 0) {
        for (uint32_t i = threadIdx.x; i < sg.bytes / 4; i += 256) ((uint32_t *) d_)[i] = ((const uint32_t *) s_)[i];
    } else {
        for (uint32_t i = threadIdx.x; i < sg.bytes; i += 256) d_[i] = s_[i];
    }
}
} // namespace
int pm355_scatter_bytes(const void * src_dev, const pm355_scatter_seg * segs, int n, pm355_stream_t st) {
    if (!src_dev || !segs || n < 1 || n > 16) return fail(PM355_E_RANGE, "scatter_bytes: 1..16 segments");
    ScatterP p = {};
    for (int i = 0; i < n; ++i) { if (!segs[i].dst) return fail(PM355_E_SHAPE, "scatter_bytes: null destination"); p.seg[i] = segs[i]; }
    (void) hipGetLastError();
    hipLaunchKernelGGL(scatter_bytes_kernel, dim3(n), dim3(256), 0, S(st), (const uint8_t *) src_dev, p);
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_q6k_tail_grouped(void) { return PM_Q6K_SCD ? 1 : 0; }
int pm355_experiments_built(void) { return PM_EXPERIMENTS ? 1 : 0; }
int pm355_mul_mat_vec_fused_grid(const pm355_matvec_job * jobs, int njobs, int64_t K) {
    if (njobs < 1 || njobs > 3 || !jobs) return fail(PM355_E_RANGE, "mul_mat_vec_fused_grid: 1..3 jobs");
    pm_gemv_fused f = {};
    f.K = (int) K; f.njobs = njobs; f.xf = (const float *) 16;
    for (int j = 0; j < njobs; ++j) { f.job[j].type = jobs[j].type; f.job[j].N = (int) jobs[j].N; f.job[j].W = jobs[j].W; f.job[j].W2 = jobs[j].W2; }
    const int g = pm_gemv_fused_grid(f);
    return g > 0 ? g : gemv_rc(g);
}
int pm355_mul_mat_vec_fused_check(const pm355_matvec_job * jobs, int njobs, int64_t K) {
    if (njobs < 1 || njobs > 3 || !jobs) return fail(PM355_E_RANGE, "mul_mat_vec_fused: 1..3 jobs");
    pm_gemv_fused f = {};
    f.K = (int) K; f.njobs = njobs; f.xf = (const float *) 16;
    for (int j = 0; j < njobs; ++j) { f.job[j].type = jobs[j].type; f.job[j].N = (int) jobs[j].N; f.job[j].W = jobs[j].W; f.job[j].W2 = jobs[j].W2; }
    return gemv_rc(pm_gemv_fused_check(f));
}
int pm355_mul_mat_q_mfma(int type, const void * W, int64_t K, int64_t N, const float * x, int64_t n_tokens, float * y,
                         const float * bias, const float * resid, pm355_stream_t st) {
    (void) hipGetLastError();
    const int rc = pm_launch_gemm_q(type, W, x, y, (int) K, (int) N, (int) n_tokens, bias, resid, S(st));
    if (rc == -1) return fail(PM355_E_UNSUPPORTED, "mul_mat_q_mfma: weight type");
    if (rc == -2) return fail(PM355_E_SHAPE, "mul_mat_q_mfma: K % 256 (K % 64 for Q8_0) and N % 4 required");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_mul_mat_q_mfma_multi(const pm355_gemm_job * jobs, int njobs, int64_t K, const float * x, int64_t n_tokens, pm355_stream_t st) {
    (void) hipGetLastError();
    if (!jobs || njobs < 1 || njobs > 4 || !x) return fail(PM355_E_RANGE, "mul_mat_q_mfma_multi: 1..4 jobs");
    pm_gemm_pf_job jb[4];
    for (int j = 0; j < njobs; ++j) jb[j] = {jobs[j].type, (int) jobs[j].N, jobs[j].W, jobs[j].y, nullptr, jobs[j].bias, jobs[j].resid, nullptr, 0};
    const int rc = pm_launch_gemm_q_multi(jb, njobs, x, nullptr, (int) K, (int) n_tokens, S(st));
    if (rc == -1) return fail(PM355_E_UNSUPPORTED, "mul_mat_q_mfma_multi: weight type");
    if (rc == -2) return fail(PM355_E_SHAPE, "mul_mat_q_mfma_multi: K % 256 (K % 64 for Q8_0) and N % 4 required");
    if (rc) return fail(PM355_E_HIP, "mul_mat_q_mfma_multi: scratch allocation");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_mul_mat_q_mfma_pair(int type, const void * W_gate, const void * W_up, int64_t K, int64_t N, const float * x, int64_t n_tokens, float * y, pm355_stream_t st) {
    (void) hipGetLastError();
    if (!W_gate || !W_up || !x || !y) return fail(PM355_E_SHAPE, "mul_mat_q_mfma_pair: null pointer");
    if (pm_gemm_pf_check(type, (int) K, (int) N, (int) n_tokens)) return fail(PM355_E_UNSUPPORTED, "mul_mat_q_mfma_pair: type / shape not served by the prompt kernel");
    const pm_gemm_pf_job jb[2] = {{type, (int) N, W_gate, nullptr, nullptr, nullptr, nullptr, nullptr, 0}, {type, (int) N, W_up, y, nullptr, nullptr, nullptr, nullptr, 0}};
    const int rc = pm_launch_gemm_q_pair(jb, x, (int) K, (int) n_tokens, S(st));
    if (rc) return fail(PM355_E_HIP, "mul_mat_q_mfma_pair: launch");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_mul_mat_q_i8(int type, const void * W, int64_t K, int64_t N, const float * x, int64_t n_tokens, float * y,
                       const float * bias, const float * resid, pm355_stream_t st) {
    if (!W || !x || !y) return fail(PM355_E_SHAPE, "mul_mat_q_i8: null pointer");
    const int rc = pm_launch_mmq_big_f32(type, W, x, y, (int) K, (int) N, (int) n_tokens, bias, resid, 0, S(st));
    if (rc == -1) return fail(PM355_E_UNSUPPORTED, "mul_mat_q_i8: weight type (Q4_K / Q6_K)");
    if (rc == -2) return fail(PM355_E_SHAPE, "mul_mat_q_i8: K % 1024 == 0 required");
    if (rc) return fail(PM355_E_HIP, "mul_mat_q_i8: scratch allocation");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_mul_mat_q_i8_check(int type, int64_t K, int64_t N, int64_t n_tokens) { return pm_mmq_big_check(type, (int) K, (int) N, (int) n_tokens) ? PM355_E_UNSUPPORTED : 0; }
int pm355_mul_mat_q_small(int type, const void * W, int64_t K, int64_t N, const void * xq, const float * x, int64_t n_tokens, float * y,
                          const float * bias, const float * resid, pm355_stream_t st) {
    (void) hipGetLastError();
    if (!xq && !x) return fail(PM355_E_RANGE, "mul_mat_q_small: neither xq nor x given");
    const int rc = pm_launch_mmq_i8(type, W, xq, x, y, (int) K, (int) N, (int) n_tokens, bias, resid, 0, S(st));
    if (rc == -1) return fail(PM355_E_UNSUPPORTED, "mul_mat_q_small: weight type (Q4_K / Q5_K / Q6_K / Q8_0)");
    if (rc == -2) return fail(PM355_E_SHAPE, "mul_mat_q_small: 1 <= n_tokens <= 64, K % 256 == 0 (Q8_0: K % 32 == 0), K >= 512 required");
    if (rc == -4) return fail(PM355_E_RANGE, "mul_mat_q_small: K too large for the LDS tiles");
    if (rc) return fail(PM355_E_HIP, "mul_mat_q_small: scratch allocation");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_mul_mat_q_small_multi(int type, int njobs, const void * const * W, const int64_t * N, float * const * y, const float * const * bias,
                                const void * xq, int64_t K, int64_t n_tokens, pm355_stream_t st) {
    (void) hipGetLastError();
    if (njobs < 2 || njobs > 3 || !W || !N || !y || !xq) return fail(PM355_E_RANGE, "mul_mat_q_small_multi: 2..3 jobs, pre-quantized activations");
    int n32[3];
    for (int j = 0; j < njobs; ++j) n32[j] = (int) N[j];
    const int rc = pm_launch_mmq_i8_multi(type, njobs, W, n32, y, bias, xq, (int) K, (int) n_tokens, 0, S(st));
    if (rc == -5) return fail(PM355_E_UNSUPPORTED, "mul_mat_q_small_multi: Q4_K / Q6_K, n_tokens <= 64, K % 256 == 0 required");
    if (rc) return fail(PM355_E_HIP, "mul_mat_q_small_multi: scratch allocation");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_mul_mat_q_small_mixed(const int * types, int njobs, const void * const * W, const int64_t * N, float * const * y, const float * const * bias,
                                const void * xq, int64_t K, int64_t n_tokens, pm355_stream_t st) {
    (void) hipGetLastError();
    if (njobs < 2 || njobs > 3 || !types || !W || !N || !y || !xq) return fail(PM355_E_RANGE, "mul_mat_q_small_mixed: 2..3 jobs, pre-quantized activations");
    int n32[3];
    for (int j = 0; j < njobs; ++j) n32[j] = (int) N[j];
    for (int j = 1; j < njobs - 1; ++j) if (types[j] != types[0]) return fail(PM355_E_UNSUPPORTED, "mul_mat_q_small_mixed: only the LAST job may be of another type");
    const int na = njobs - 1;
    const int rc = pm_launch_mmq_i8_dual(types[0], na, W, n32, y, bias, types[na], W[na], n32[na], y[na], bias ? bias[na] : nullptr, xq, (int) K, (int) n_tokens, 0, S(st));
    if (rc == -5) return fail(PM355_E_UNSUPPORTED, "mul_mat_q_small_mixed: Q4_K jobs + one Q6_K (n_tokens <= 32) / Q5_K (<= 16) job, K % 256 == 0 required");
    if (rc) return fail(PM355_E_HIP, "mul_mat_q_small_mixed: scratch allocation");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_gemm_plan(int64_t n_row_tiles, int64_t K, int64_t n_tokens, int n_cus, int allow_tail_split, int32_t * out5) {
    if (n_row_tiles < 1 || K < 256 || n_tokens < 1 || n_cus < 8 || !out5) return fail(PM355_E_RANGE, "gemm_plan: n_row_tiles >= 1, K >= 256, n_tokens >= 1, n_cus >= 8");
    pm_gemm_pf_plan_t pl;
    pm_gemm_pf_plan((int) n_row_tiles, (int) K, (int) n_tokens, n_cus, 0, 0, allow_tail_split, &pl);
    out5[0] = pl.nt * 32; out5[1] = pl.nt_t; out5[2] = pl.splitk; out5[3] = pl.full; out5[4] = pl.grid;
    return 0;
}
int pm355_mul_mat_q_small_check(int type, int64_t K, int64_t N, int64_t n_tokens) {
    return pm_mmq_i8_check(type, (int) K, (int) N, (int) n_tokens) == 0 ? 0 : PM355_E_UNSUPPORTED;
}
int pm355_mul_mat_vec_q_dbg(int type, const void * W, int64_t K, int64_t N, const void * xq, float * y,
                            int32_t * ip, int64_t * upr, pm355_stream_t st) {
    if (upr) *upr = pm_gemv_units_per_row(type, K);
    if (!ip && !y) return 0;                          // query of the unit count only
    pm_gemv_args a = {};
    a.type = type; a.K = (int) K; a.N = (int) N; a.W = W; a.xq = xq; a.ncols = 1; a.y = y; a.y_stride = (size_t) N; a.dbg_int = ip;
    return gemv_rc(pm_launch_gemv(a, S(st)));
}

int pm355_get_rows(int type, const void * table, int64_t K, const int32_t * d_tokens, int n_tokens, float * out, pm355_stream_t st) {
    if (!pm_weight_row_bytes(type, K)) return fail(PM355_E_UNSUPPORTED, "get_rows: type");
    pm_launch_embed(type, table, (int) K, d_tokens, n_tokens, out, S(st));
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_rope_kv_store(const float * q, const float * k, const float * v, float * q_out, float * k_out_f32,
                        void * kc, void * vc, const int32_t * d_pos0, const float * ff, int n_tokens, int H, int Hkv, int dh,
                        int n_ctx, const pm355_rope_params * rp, pm355_stream_t st) {
    if (!rp || rp->n_dims % 2 || rp->n_dims > dh || dh % 2) return fail(PM355_E_SHAPE, "rope: n_dims");
    pm_rope_cfg c;
    c.n_dims = rp->n_dims; c.mode = rp->mode; c.n_ctx_orig = rp->n_ctx_orig; c.freq_base = rp->freq_base; c.freq_scale = rp->freq_scale;
    c.ext_factor = rp->ext_factor; c.attn_factor = rp->attn_factor; c.beta_fast = rp->beta_fast; c.beta_slow = rp->beta_slow;
    pm_rope_params(c);
    pm_launch_rope_kv_store(q, k, v, q_out, k_out_f32, kc, vc, d_pos0, nullptr, 0, ff, n_tokens, H, Hkv, dh, n_ctx, c, S(st));
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_attn_decode(const float * q, const void * kc, const void * vc, const int32_t * d_pos0, float * out, int n_tokens,
                      int H, int Hkv, int dh, int n_ctx, float kq_scale, pm355_stream_t st) {
    if (pm_launch_attn_decode(q, kc, vc, d_pos0, nullptr, 0, out, n_tokens, H, Hkv, dh, n_ctx, kq_scale, S(st)))
        return fail(PM355_E_RANGE, "attn_decode: n_ctx/head_dim unsupported (LDS budget 150 KB, multiples of 8)");
    HIP_TRY(hipGetLastError());
    return 0;
}
size_t pm355_attn_split_scratch_floats(int H, int dh, int n_ctx) {       // (covers both long-context forms: attn_split.hip and attn_flash.hip)
    const size_t a = pm_attn_split_scratch_floats(H, dh, n_ctx), b = pm_attn_flash_scratch_floats(H, H, dh, n_ctx);
    return a > b ? a : b;
}
int pm355_attn_decode_split(const float * q_rot, const void * kc, const void * vc, const int32_t * d_pos0, float * out, float * scratch,
                            int H, int Hkv, int dh, int n_ctx, float kq_scale, pm355_stream_t st) {
    if (pm_launch_attn_split(q_rot, nullptr, nullptr, (void *) kc, (void *) vc, d_pos0, nullptr, 0, nullptr, out, scratch, H, Hkv, dh, n_ctx, kq_scale, nullptr, S(st)))
        return fail(PM355_E_RANGE, "attn_decode_split: head_dim 64/128, at most 8 query heads per KV head, n_ctx % 8 == 0, scratch required");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_attn_prefill(const float * q, const void * kc, const void * vc, const int32_t * d_pos0, float * out, int n_tokens,
                       int H, int Hkv, int dh, int n_ctx, float kq_scale, pm355_stream_t st) {
    if (pm_launch_attn_prefill(q, kc, vc, d_pos0, nullptr, 0, out, n_tokens, H, Hkv, dh, n_ctx, kq_scale, S(st)))
        return fail(PM355_E_RANGE, "attn_prefill: head_dim must be 64 or 128 and n_ctx a multiple of 32");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_attn_prefill_masked_ex(const float * q, const void * kc, const void * vc, const void * mask, int64_t mask_stride, float * out,
                                 int n_tokens, int H, int Hkv, int dh, int n_ctx, int n_kv, float kq_scale, int flags, pm355_stream_t st) {
    if (!mask) return fail(PM355_E_SHAPE, "attn_prefill_masked: mask required");
    if (pm_launch_attn_prefill(q, kc, vc, nullptr, nullptr, 0, out, n_tokens, H, Hkv, dh, n_ctx, kq_scale, S(st), (const float *) mask, (long) mask_stride, n_kv,
                               flags & PM355_ATTN_V_ROWMAJOR, flags & PM355_ATTN_MASK_F16))
        return fail(PM355_E_RANGE, "attn_prefill_masked: head_dim 64/128, n_ctx % 32 == 0, n_kv % 4 == 0 <= n_ctx required");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_attn_prefill_masked(const float * q, const void * kc, const void * vc, const float * mask, int64_t mask_stride, float * out,
                              int n_tokens, int H, int Hkv, int dh, int n_ctx, int n_kv, float kq_scale, pm355_stream_t st) {
    return pm355_attn_prefill_masked_ex(q, kc, vc, mask, mask_stride, out, n_tokens, H, Hkv, dh, n_ctx, n_kv, kq_scale, 0, st);
}
int pm355_attn_rope_fused(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * d_pos0,
                          const float * ff, float * out, int H, int Hkv, int dh, int n_ctx, float kq_scale,
                          const pm355_rope_params * rp, pm355_stream_t st) {
    if (!rp || rp->n_dims % 2 || rp->n_dims > dh) return fail(PM355_E_SHAPE, "attn_rope_fused: n_dims");
    pm_rope_cfg c;
    c.n_dims = rp->n_dims; c.mode = rp->mode; c.n_ctx_orig = rp->n_ctx_orig; c.freq_base = rp->freq_base; c.freq_scale = rp->freq_scale;
    c.ext_factor = rp->ext_factor; c.attn_factor = rp->attn_factor; c.beta_fast = rp->beta_fast; c.beta_slow = rp->beta_slow;
    pm_rope_params(c);
    if (pm_launch_attn_rope_fused(q, k, v, kc, vc, d_pos0, nullptr, 0, ff, out, H, Hkv, dh, n_ctx, kq_scale, c, S(st)))
        return fail(PM355_E_UNSUPPORTED, "attn_rope_fused: head_dim must be 64/128/256, n_ctx % 8 == 0 and fit LDS");
    HIP_TRY(hipGetLastError());
    return 0;
}
static pm_rope_cfg rope_cfg_of(const pm355_rope_params * rp) {
    pm_rope_cfg c;
    c.n_dims = rp->n_dims; c.mode = rp->mode; c.n_ctx_orig = rp->n_ctx_orig; c.freq_base = rp->freq_base; c.freq_scale = rp->freq_scale;
    c.ext_factor = rp->ext_factor; c.attn_factor = rp->attn_factor; c.beta_fast = rp->beta_fast; c.beta_slow = rp->beta_slow;
    pm_rope_params(c);
    return c;
}
int pm355_rope_table(const pm355_rope_params * rp, const int32_t * d_pos, const float * ff, float * table, pm355_stream_t st) {
    if (!rp || rp->n_dims % 2 || rp->n_dims > 256 || !d_pos || !table) return fail(PM355_E_SHAPE, "rope_table: n_dims (even, <= 256), d_pos and table required");
    pm_launch_rope_table(rope_cfg_of(rp), d_pos, nullptr, ff, table, S(st));
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_mul_mat_vec_qkv(const pm355_matvec_job * jobs, int64_t K, const float * x_f32, const float * norm_w, float eps,
                          const pm355_qkv_store * s, pm355_stream_t st) {
    return pm355_mul_mat_vec_qkv_ss(jobs, K, x_f32, norm_w, eps, s, nullptr, 0, st);
}
int pm355_mul_mat_vec_qkv_ss(const pm355_matvec_job * jobs, int64_t K, const float * x_f32, const float * norm_w, float eps,
                             const pm355_qkv_store * s, const double * sumsq_in, int n_sumsq_in, pm355_stream_t st) {
    return pm355_mul_mat_vec_qkv_attn(jobs, K, x_f32, norm_w, eps, s, sumsq_in, n_sumsq_in, nullptr, st);
}
int pm355_mul_mat_vec_qkv_attn(const pm355_matvec_job * jobs, int64_t K, const float * x_f32, const float * norm_w, float eps,
                               const pm355_qkv_store * s, const double * sumsq_in, int n_sumsq_in, const pm355_qkv_attn * at, pm355_stream_t st) {
    if (!jobs || !x_f32 || !s || !s->rope_table || !s->k_cache || !s->v_cache || (!s->d_pos && !s->d_cell_nkv)) return fail(PM355_E_SHAPE, "mul_mat_vec_qkv: null pointer");
    pm_gemv_fused f = {};
    f.K = (int) K; f.njobs = 3; f.xf = x_f32; f.norm_w = norm_w; f.eps = eps;
    if (sumsq_in && n_sumsq_in > 0 && norm_w) { f.ss_in = sumsq_in; f.n_ss = n_sumsq_in; }
    for (int j = 0; j < 3; ++j) {
        f.job[j].type = jobs[j].type; f.job[j].N = (int) jobs[j].N; f.job[j].W = jobs[j].W; f.job[j].W2 = nullptr;
        f.job[j].y = jobs[j].y; f.job[j].bias = jobs[j].bias; f.job[j].resid = nullptr;
    }
    pm_qkv_epi e = {s->rope_table, s->d_pos, nullptr, s->d_cell_nkv, 0, s->k_cache, s->v_cache, s->n_head_kv, s->head_dim, s->n_ctx, s->n_rot,
                    s->v_rowmajor, s->rope_neox};
    if (at) {
        if (!at->out || !at->ticket) return fail(PM355_E_SHAPE, "mul_mat_vec_qkv_attn: null pointer");
        e.att_out = at->out; e.att_ticket = at->ticket; e.att_err = at->watchdog; e.kq_scale = at->kq_scale; e.n_head = at->n_head; e.att_max_keys = at->max_keys;
    }
    f.epi = &e;
    (void) hipGetLastError();
    const int rc = pm_launch_gemv_fused(f, S(st));
    if (rc == -8) return gemv_rc(-8);
    if (rc == -7) return fail(PM355_E_UNSUPPORTED, "mul_mat_vec_qkv_attn: the workgroups of a KV-head group must be a power-of-two run of the grid (CUs % n_head_kv == 0), head_dim 64 / 128, position-pointer mode, transposed V cache");
    if (rc == -5) return fail(PM355_E_UNSUPPORTED, "mul_mat_vec_qkv: every workgroup's row slices must hold whole rotation pairs (N % (2 * CUs) == 0), N_k == N_v == n_head_kv * head_dim");
    if (rc) return gemv_rc(rc);
    HIP_TRY(hipGetLastError());
    return 0;
}
#if PM_EXPERIMENTS
int pm355_engine_run(const pm355_engine_phase * phs, int n, pm355_stream_t st) {
    if (!phs || n < 1) return fail(PM355_E_SHAPE, "engine_run: phases");
    pm_eng_plan * pl = pm_eng_plan_new();
    int rc = 0;
    for (int i = 0; i < n && !rc; ++i) {
        const pm355_engine_phase & e = phs[i];
        if (e.kind == 0) {
            if (e.njobs < 1 || e.njobs > 3 || !e.jobs || !e.x_f32) { rc = -100; break; }
            pm_gemv_fused f = {};
            f.K = (int) e.K; f.njobs = e.njobs; f.xf = e.x_f32; f.norm_w = e.norm_w; f.eps = e.eps;
            f.ss_out = e.sumsq_out; f.ss_in = e.sumsq_in; f.n_ss = e.n_sumsq_in;
            for (int j = 0; j < e.njobs; ++j) {
                f.job[j].type = e.jobs[j].type; f.job[j].N = (int) e.jobs[j].N; f.job[j].W = e.jobs[j].W; f.job[j].W2 = e.jobs[j].W2;
                f.job[j].y = e.jobs[j].y; f.job[j].bias = e.jobs[j].bias; f.job[j].resid = e.jobs[j].resid;
            }
            pm_qkv_epi qe = {};
            if (e.qkv) {
                const pm355_qkv_store * s = e.qkv;
                if (s->d_cell_nkv || !s->d_pos || e.njobs != 3) { rc = -100; break; }
                qe = pm_qkv_epi{s->rope_table, s->d_pos, nullptr, nullptr, 0, s->k_cache, s->v_cache, s->n_head_kv, s->head_dim, s->n_ctx, s->n_rot, s->v_rowmajor, s->rope_neox};
                f.epi = &qe;
            }
            rc = pm_eng_plan_add_matvec(pl, f);
        } else if (e.kind == 1) {
            rc = pm_eng_plan_add_attention(pl, e.q_rot, e.k_cache, e.v_cache, e.d_pos, nullptr, 0, e.out, e.n_head, e.n_head_kv, e.head_dim, e.n_ctx, e.kq_scale, e.max_keys);
        } else rc = -100;
    }
    if (!rc) rc = pm_eng_plan_finish(pl);
    if (rc) { pm_eng_plan_free(pl); (void) hipGetLastError(); char msg[96]; snprintf(msg, sizeof(msg), "engine_run: phase list not served (code %d)", rc); return fail(PM355_E_UNSUPPORTED, msg); }
    (void) hipGetLastError();
    pm_eng_plan_launch(pl, S(st));
    const hipError_t le = hipGetLastError();
    const hipError_t se = hipStreamSynchronize(S(st));
    const int w = pm_eng_plan_status(pl);
    pm_eng_plan_free(pl);
    if (le != hipSuccess) return fail(PM355_E_HIP, "engine_run: launch", le);
    if (se != hipSuccess) return fail(PM355_E_HIP, "engine_run: synchronize", se);
    if (w) { char msg[96]; snprintf(msg, sizeof(msg), "engine_run: watchdog code %d", w); return fail(PM355_E_HIP, msg); }
    return 0;
}
#else
int pm355_engine_run(const pm355_engine_phase *, int, pm355_stream_t) { return fail(PM355_E_UNSUPPORTED, "engine_run: the persistent decode engine is built into libprima_mi355_exp.so only (PM_EXPERIMENTS)"); }
#endif
int pm355_mul_mat_vec_qkv_check_ex(const pm355_matvec_job * jobs, int64_t K, int n_head_kv, int head_dim, int n_rot, int rope_neox) {
    if (!jobs) return fail(PM355_E_SHAPE, "mul_mat_vec_qkv_check: jobs");
    pm_gemv_fused f = {};
    f.K = (int) K; f.njobs = 3; f.xf = (const float *) 16;
    for (int j = 0; j < 3; ++j) { f.job[j].type = jobs[j].type; f.job[j].N = (int) jobs[j].N; f.job[j].W = jobs[j].W; f.job[j].bias = jobs[j].bias; }
    const pm_qkv_epi e = {(const float *) 16, (const int32_t *) 16, nullptr, nullptr, 0, (void *) 16, (void *) 16, n_head_kv, head_dim, 8, n_rot, 0, rope_neox ? 1 : 0};
    f.epi = &e;
    const int rc = pm_gemv_fused_check(f);
    return rc == -5 ? PM355_E_UNSUPPORTED : gemv_rc(rc);
}
int pm355_mul_mat_vec_qkv_check(const pm355_matvec_job * jobs, int64_t K, int n_head_kv, int head_dim, int n_rot) {
    return pm355_mul_mat_vec_qkv_check_ex(jobs, K, n_head_kv, head_dim, n_rot, 0);
}
int pm355_attn_cached(const float * q_rot, void * kc, void * vc, const int32_t * d_pos, const int32_t * d_cell_nkv, const void * mask,
                      float * out, int H, int Hkv, int dh, int n_ctx, float kq_scale, int max_keys, int flags, pm355_stream_t st) {
    if (!q_rot || !kc || !vc || !out || (!d_pos && !d_cell_nkv)) return fail(PM355_E_SHAPE, "attn_cached: null pointer");
    if (pm_launch_attn_cached(q_rot, kc, vc, d_pos, nullptr, 0, out, H, Hkv, dh, n_ctx, kq_scale, S(st), d_cell_nkv, mask, max_keys,
                              flags & PM355_ATTN_V_ROWMAJOR, flags & PM355_ATTN_MASK_F16))
        return fail(PM355_E_UNSUPPORTED, "attn_cached: head_dim must be 64/128/256, n_ctx % 8 == 0 and max_keys fit LDS");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_attn_cached_long(const float * q_rot, void * kc, void * vc, const int32_t * d_pos, const int32_t * d_cell_nkv, const void * mask,
                           float * out, float * scratch, int H, int Hkv, int dh, int n_ctx, float kq_scale, int max_cells, int flags, pm355_stream_t st) {
    if (!q_rot || !kc || !vc || !out || !scratch || (!d_pos && !d_cell_nkv)) return fail(PM355_E_SHAPE, "attn_cached_long: null pointer");
    if (flags & ~(PM355_ATTN_MASK_F16 | PM355_ATTN_V_ROWMAJOR | PM355_ATTN_K_Q8_0 | PM355_ATTN_V_Q8_0)) return fail(PM355_E_UNSUPPORTED, "attn_cached_long: unknown flag");
    if (pm_launch_attn_flash_cached(q_rot, kc, vc, d_pos, nullptr, 0, out, scratch, H, Hkv, dh, n_ctx, kq_scale, S(st), d_cell_nkv, mask,
                                    flags & PM355_ATTN_MASK_F16, max_cells, (flags & PM355_ATTN_V_ROWMAJOR) ? 1 : 0,
                                    (flags & PM355_ATTN_K_Q8_0) ? 1 : 0, (flags & PM355_ATTN_V_Q8_0) ? 1 : 0))
        return fail(PM355_E_UNSUPPORTED, "attn_cached_long: head_dim must be 64/128, <= 16 query heads per KV head, n_ctx % 8 == 0; Q8_0 caches need row-major V");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_attn_cached_long_check(int H, int Hkv, int dh, int n_ctx) { return pm_attn_flash_cached_ok(H, Hkv, dh, n_ctx) ? PM355_E_UNSUPPORTED : 0; }
int pm355_attn_token(const pm355_attn_token_args * a, const pm355_rope_params * rp, pm355_stream_t st) {
    if (!a || !rp || rp->n_dims % 2 || rp->n_dims > a->head_dim) return fail(PM355_E_SHAPE, "attn_token: n_dims");
    if (!a->d_pos || !a->d_cell_nkv || !a->q || !a->k || !a->v || !a->k_cache || !a->v_cache || !a->out) return fail(PM355_E_SHAPE, "attn_token: null pointer");
    pm_rope_cfg c;
    c.n_dims = rp->n_dims; c.mode = rp->mode; c.n_ctx_orig = rp->n_ctx_orig; c.freq_base = rp->freq_base; c.freq_scale = rp->freq_scale;
    c.ext_factor = rp->ext_factor; c.attn_factor = rp->attn_factor; c.beta_fast = rp->beta_fast; c.beta_slow = rp->beta_slow;
    pm_rope_params(c);
    (void) hipGetLastError();
    if (a->flags & (PM355_ATTN_K_Q8_0 | PM355_ATTN_V_Q8_0)) {
        if (!(a->flags & PM355_ATTN_V_ROWMAJOR)) return fail(PM355_E_UNSUPPORTED, "attn_token: a quantized KV cache needs row-major V (flash-attention graphs)");
        if (a->split) {
            // long context: RoPE + quantizing KV store (one small launch), then scores and P.V on the matrix cores over the cached Q8_0 / F16 cells
            // (attn_flash_mfma.hip); max_keys = cells the grid is sized for, scratch zeroed once after its allocation
            if (!a->scratch) return fail(PM355_E_SHAPE, "attn_token(q8_0, long): scratch");
            float * q_rot = a->scratch + pm_attn_flash_qrot_offset(a->n_head, a->n_head_kv, a->head_dim, a->n_ctx);
            const int kq = (a->flags & PM355_ATTN_K_Q8_0) ? 1 : 0, vq = (a->flags & PM355_ATTN_V_Q8_0) ? 1 : 0;
            if (pm_launch_q8_token_prep(a->q, a->k, a->v, a->k_cache, a->v_cache, a->d_pos, a->d_cell_nkv, a->freq_factors, q_rot, a->n_head, a->n_head_kv,
                                        a->head_dim, a->n_ctx, c, kq, vq, S(st)) ||
                pm_launch_attn_flash_cached(q_rot, a->k_cache, a->v_cache, nullptr, nullptr, 0, a->out, a->scratch, a->n_head, a->n_head_kv, a->head_dim, a->n_ctx,
                                            a->kq_scale, S(st), a->d_cell_nkv, a->mask, a->flags & PM355_ATTN_MASK_F16, a->max_keys, 1, kq, vq))
                return fail(PM355_E_UNSUPPORTED, "attn_token(q8_0, long): head_dim 64/128, <= 16 query heads per KV head, <= 16 KV heads, n_ctx % 8 == 0");
            HIP_TRY(hipGetLastError());
            return 0;
        }
        if (pm_launch_attn_q8_token(a->q, a->k, a->v, a->k_cache, a->v_cache, a->d_pos, a->d_cell_nkv, a->mask, a->flags & PM355_ATTN_MASK_F16,
                                    a->freq_factors, a->out, a->n_head, a->n_head_kv, a->head_dim, a->n_ctx, a->kq_scale, c,
                                    a->flags & PM355_ATTN_K_Q8_0, a->flags & PM355_ATTN_V_Q8_0, a->max_keys, S(st)))
            return fail(PM355_E_UNSUPPORTED, "attn_token(q8_0): head_dim 64/128 and max_keys must fit LDS");
        HIP_TRY(hipGetLastError());
        return 0;
    }
    static const bool three_launch = [] { const char * e = getenv("PM355_ATTN_FLASH"); return e && e[0] == '0'; }();
    if (a->split && !three_launch) {
        // long context, ONE launch: flash-decoding with an in-launch merge (attn_flash.hip); max_keys (> 0) = cells the grid is sized for.
        // The scratch must have been zeroed once after its allocation (ticket words).
        if (pm_launch_attn_flash(a->q, a->k, a->v, a->k_cache, a->v_cache, a->d_pos, nullptr, 0, a->freq_factors, a->out, a->scratch,
                                 a->n_head, a->n_head_kv, a->head_dim, a->n_ctx, a->kq_scale, c, S(st), a->d_cell_nkv, a->mask,
                                 a->flags & PM355_ATTN_V_ROWMAJOR, a->flags & PM355_ATTN_MASK_F16, a->max_keys))
            return fail(PM355_E_UNSUPPORTED, "attn_token(flash): head_dim 64/128, at most 8 query heads per KV head, n_ctx % 8 == 0, scratch required");
    } else if (a->split) {
        if (pm_launch_attn_split(a->q, a->k, a->v, a->k_cache, a->v_cache, a->d_pos, nullptr, 0, a->freq_factors, a->out, a->scratch,
                                 a->n_head, a->n_head_kv, a->head_dim, a->n_ctx, a->kq_scale, &c, S(st), a->d_cell_nkv, a->mask,
                                 a->flags & PM355_ATTN_V_ROWMAJOR, a->flags & PM355_ATTN_MASK_F16))
            return fail(PM355_E_UNSUPPORTED, "attn_token(split): head_dim 64/128, at most 8 query heads per KV head, n_ctx % 8 == 0, scratch required");
    } else if (pm_launch_attn_rope_fused(a->q, a->k, a->v, a->k_cache, a->v_cache, a->d_pos, nullptr, 0, a->freq_factors, a->out,
                                         a->n_head, a->n_head_kv, a->head_dim, a->n_ctx, a->kq_scale, c, S(st), a->d_cell_nkv, a->mask, a->max_keys,
                                         a->flags & PM355_ATTN_V_ROWMAJOR, a->flags & PM355_ATTN_MASK_F16))
        return fail(PM355_E_UNSUPPORTED, "attn_token: head_dim must be 64/128/256, n_ctx % 8 == 0 and max_keys fit LDS");
    HIP_TRY(hipGetLastError());
    return 0;
}
int pm355_set_i32x2(int32_t * d_p, int32_t a, int32_t b, pm355_stream_t st) { pm_launch_set_i32x2(d_p, a, b, S(st)); HIP_TRY(hipGetLastError()); return 0; }

int pm355_capture_begin(pm355_stream_t st) { HIP_TRY(hipStreamBeginCapture(S(st), hipStreamCaptureModeRelaxed)); return 0; }
pm355_graph_t pm355_capture_end(pm355_stream_t st) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(S(st), &g);
    if (e != hipSuccess || !g) { fail(PM355_E_HIP, "hipStreamEndCapture", e); if (g) (void) hipGraphDestroy(g); return nullptr; }
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    (void) hipGraphDestroy(g);
    if (e != hipSuccess) { fail(PM355_E_HIP, "hipGraphInstantiate", e); return nullptr; }
    return (pm355_graph_t) x;
}
int pm355_graph_launch(pm355_graph_t g, pm355_stream_t st) { HIP_TRY(hipGraphLaunch((hipGraphExec_t) g, S(st))); return 0; }
void pm355_graph_free(pm355_graph_t g) { if (g) (void) hipGraphExecDestroy((hipGraphExec_t) g); }

int pm355_argmax(const float * x, int64_t n, int32_t * d_index, float * d_value, pm355_stream_t st) {
    pm_launch_argmax(x, (int) n, d_index, d_value, S(st)); HIP_TRY(hipGetLastError()); return 0;
}
int pm355_add(const float * a, const float * b, float * y, int64_t n, int64_t nb, pm355_stream_t st) { pm_launch_add(a, b, y, n, nb, S(st)); HIP_TRY(hipGetLastError()); return 0; }
int pm355_mul(const float * a, const float * b, float * y, int64_t n, int64_t nb, pm355_stream_t st) { pm_launch_mul(a, b, y, n, nb, S(st)); HIP_TRY(hipGetLastError()); return 0; }
int pm355_silu_mul(const float * g, const float * u, float * y, int64_t n, pm355_stream_t st) { pm_launch_silu_mul(g, u, y, n, S(st)); HIP_TRY(hipGetLastError()); return 0; }
int pm355_scale(const float * a, float * y, float s, int64_t n, pm355_stream_t st) { pm_launch_scale(a, y, s, n, S(st)); HIP_TRY(hipGetLastError()); return 0; }
int pm355_fill_random_blocks(int type, void * dst, int64_t K, int64_t nrows, uint64_t seed, float scale, pm355_stream_t st) {
    if (!pm_weight_row_bytes(type, K)) return fail(PM355_E_UNSUPPORTED, "fill_random_blocks: type");
    pm_launch_fill_random_blocks(type, dst, K, nrows, seed, scale, S(st)); HIP_TRY(hipGetLastError()); return 0;
}
} // extern "C"
