/* prima_mi355.h — C ABI of libprima_mi355.so: the MI355X (gfx950) implementation of prima.cpp's
 * quantized decode hot path.
 *
 * Two layers, both `extern "C"`, plain pointers and sizes only (no torch / C++ types):
 *
 *  (A) op level  — one entry point per ggml op on the path. These are what the ggml-backend plug-in
 *      (libggml-mi355.so, prima_cpp_amd/csrc/ggml_backend_mi355.cpp, entry ggml_backend_mi355_reg(), see
 *      include/ggml_backend_mi355.h) dispatches to from its graph_compute, i.e. they replace the reference's
 *      ggml_compute_forward_* CPU functions for tensors living in our buffer type. Each cites the
 *      reference function it replaces (paths relative to the reference repo).
 *
 *  (B) engine level — a resident decoder (weights + KV cache in HBM, one hipGraph per layer window)
 *      that replaces the per-token llama_build_graph + ggml_backend_sched + graph_compute round
 *      (src/llama.cpp:18455-18526) for the layer window a rank owns in prima.cpp's piped-ring.
 *
 * All device pointers are HIP device pointers valid on the current device; `stream` is a hipStream_t
 * (pass NULL for the default stream). Every call is asynchronous with respect to the host unless
 * stated otherwise. Return value: 0 on success, negative pm355 error code otherwise (no exceptions
 * cross this boundary; unsupported shapes return PM355_E_UNSUPPORTED and do nothing).
 */
#ifndef PRIMA_MI355_H
#define PRIMA_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM355_API __attribute__((visibility("default")))

/* ggml type ids, identical to enum ggml_type (ggml/include/ggml.h:356-372) */
enum { PM355_TYPE_F32 = 0, PM355_TYPE_F16 = 1, PM355_TYPE_Q8_0 = 8, PM355_TYPE_Q4_K = 12,
       PM355_TYPE_Q5_K = 13, PM355_TYPE_Q6_K = 14 };

enum { PM355_OK = 0, PM355_E_UNSUPPORTED = -1, PM355_E_SHAPE = -2, PM355_E_ALIGN = -3, PM355_E_RANGE = -4,
       PM355_E_HIP = -10, PM355_E_NOMEM = -11 };

typedef void * pm355_stream_t;          /* hipStream_t */

/* ---- library / device --------------------------------------------------------------------------- */
PM355_API const char * pm355_version(void);
PM355_API int    pm355_device_count(void);
PM355_API int    pm355_set_device(int device);
PM355_API int    pm355_device_info(int device, char * name, size_t name_len, size_t * free_bytes, size_t * total_bytes,
                                   int * compute_units);
PM355_API int    pm355_sync(pm355_stream_t stream);
PM355_API const char * pm355_last_error(void);

/* device memory helpers (hipMalloc / hipFree / hipMemcpy[Async]); used by the plug-in's buffer vtable
 * (replaces ggml_backend_cuda_buffer_* of the reference's CUDA plug-in, ggml/src/ggml-cuda.cu:430-560) */
PM355_API void * pm355_malloc(size_t bytes);
PM355_API void   pm355_free(void * dptr);
PM355_API void * pm355_host_malloc(size_t bytes);                 /* pinned */
PM355_API void   pm355_host_free(void * hptr);
PM355_API int    pm355_memcpy_h2d(void * dst, const void * src, size_t bytes, pm355_stream_t stream);
PM355_API int    pm355_memcpy_d2h(void * dst, const void * src, size_t bytes, pm355_stream_t stream);
PM355_API int    pm355_memcpy_d2d(void * dst, const void * src, size_t bytes, pm355_stream_t stream);
PM355_API int    pm355_memset(void * dst, int value, size_t bytes, pm355_stream_t stream);

/* ---- weight layout ------------------------------------------------------------------------------ */
/* bytes of one row of K weights == ggml_row_size(type, K) (ggml/src/ggml.c:3579) */
PM355_API size_t pm355_row_size(int type, int64_t K);
/* Row-local re-ordering GGUF block order <-> HBM order (identity for F32/F16/Q4_K/Q5_K, row-SoA for
 * Q6_K/Q8_0; see prima_cpp_amd/csrc/repack.hip). src and dst are DEVICE pointers and must not overlap.
 * to_device_layout=1 is what set_tensor does after the H2D copy; 0 is what get_tensor does before D2H. */
PM355_API int    pm355_repack_rows(int type, const void * src, void * dst, int64_t K, int64_t nrows,
                                   int to_device_layout, pm355_stream_t stream);

/* ---- (A) ops ------------------------------------------------------------------------------------ */
/* bytes of one quantized activation row (library-internal row-SoA layout) */
PM355_API size_t pm355_q8_K_row_size(int64_t K);
PM355_API size_t pm355_q8_0_row_size(int64_t K);

/* replaces quantize_row_q8_K (ggml/src/ggml-quants.c:3785-3835): x f32 [rows][K] -> Q8_K rows */
PM355_API int pm355_quantize_q8_K(const float * x, void * yq, int64_t K, int64_t rows, pm355_stream_t stream);
/* replaces quantize_row_q8_0 (ggml/src/ggml-quants.c:848-873) */
PM355_API int pm355_quantize_q8_0(const float * x, void * yq, int64_t K, int64_t rows, pm355_stream_t stream);
/* copy a quantized activation row back in the REFERENCE's block layout (block_q8_K / block_q8_0), for tests */
PM355_API int pm355_act_to_ggml_blocks(int act_type /*15=Q8_K, 8=Q8_0*/, const void * yq, void * blocks_out,
                                       int64_t K, int64_t rows, pm355_stream_t stream);

/* replaces ggml_compute_forward_rms_norm_f32 (ggml/src/ggml.c:11950) [+ ggml_compute_forward_mul_f32 (:10077)
 * when w != NULL]. y_f32 and/or yq (Q8_K) may be NULL. One pass over x. */
PM355_API int pm355_rms_norm(const float * x, const float * w, float * y_f32, void * yq, int64_t K, int64_t rows,
                             float eps, pm355_stream_t stream);

/* replaces ggml_compute_forward_mul_mat (ggml/src/ggml.c:12377) for quantized W (HBM layout, N rows of K) and
 * ncols <= 8 activation columns ALREADY quantized to vec_dot_type (Q8_K, or Q8_0 for Q8_0 weights):
 *   y[c*y_stride + n] = dot(W[n,:], x_c) (+ bias[n]) (+ resid[c*y_stride + n])
 * W2 != NULL: y = silu(W.x) * (W2.x)  (llm_build_ffn LLM_FFN_SILU/LLM_FFN_PAR, src/llama.cpp:9804-9929). */
PM355_API int pm355_mul_mat_vec_q(int type, const void * W, const void * W2, int64_t K, int64_t N,
                                  const void * xq, int ncols, float * y, int64_t y_stride,
                                  const float * bias, const float * resid, pm355_stream_t stream);
/* test hook: additionally writes, per (row, unit), the exact int32 pair {sum scale*q_w*q_a, sum min*bsum} */
PM355_API int pm355_mul_mat_vec_q_dbg(int type, const void * W, int64_t K, int64_t N, const void * xq, float * y,
                                      int32_t * int_partials, int64_t * units_per_row, pm355_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
