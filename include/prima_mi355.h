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
/* waits for the work enqueued on the NULL stream only (small synchronous uploads), not for the other streams of the device */
PM355_API int    pm355_sync_null_stream(void);
/* streams / events for the plug-in's ggml_backend_i (a ggml backend IS a stream) and ggml_backend_event */
typedef void * pm355_event_t;            /* hipEvent_t */
PM355_API pm355_stream_t pm355_stream_create(void);
PM355_API void   pm355_stream_destroy(pm355_stream_t stream);
PM355_API pm355_event_t  pm355_event_create(void);
PM355_API void   pm355_event_destroy(pm355_event_t ev);
PM355_API int    pm355_event_record(pm355_event_t ev, pm355_stream_t stream);
PM355_API int    pm355_event_wait(pm355_stream_t stream, pm355_event_t ev);      /* stream waits for ev */
PM355_API int    pm355_event_sync(pm355_event_t ev);                              /* host waits for ev */
PM355_API const char * pm355_last_error(void);

/* hipGraph capture of everything launched on `stream` between begin and end (relaxed mode; nothing executes while capturing):
 * what the plug-in's graph_compute uses to replay a token's launch sequence (cf. the reference CUDA plug-in's CUDA-graph path,
 * ggml/src/ggml-cuda.cu:2513-2780). end returns an instantiated executable graph or NULL. */
typedef void * pm355_graph_t;            /* hipGraphExec_t */
PM355_API int    pm355_capture_begin(pm355_stream_t stream);
PM355_API pm355_graph_t pm355_capture_end(pm355_stream_t stream);
PM355_API int    pm355_graph_launch(pm355_graph_t g, pm355_stream_t stream);
PM355_API void   pm355_graph_free(pm355_graph_t g);

/* device memory helpers (hipMalloc / hipFree / hipMemcpy[Async]); used by the plug-in's buffer vtable
 * (replaces ggml_backend_cuda_buffer_* of the reference's CUDA plug-in, ggml/src/ggml-cuda.cu:430-560) */
PM355_API void * pm355_malloc(size_t bytes);
PM355_API void   pm355_free(void * dptr);
PM355_API void * pm355_host_malloc(size_t bytes);                 /* pinned */
PM355_API void   pm355_host_free(void * hptr);
PM355_API int    pm355_memcpy_h2d(void * dst, const void * src, size_t bytes, pm355_stream_t stream);
PM355_API int    pm355_memcpy_d2h(void * dst, const void * src, size_t bytes, pm355_stream_t stream);
/* 1 when `p` is page-locked host memory known to the HIP runtime (hipHostMalloc / hipHostRegister): an asynchronous copy FROM it really is
 * asynchronous (the DMA reads it after the call returned), unlike a copy from pageable memory, which is staged before the call returns */
PM355_API int    pm355_host_is_pinned(const void * p);
PM355_API int    pm355_memcpy_d2d(void * dst, const void * src, size_t bytes, pm355_stream_t stream);
PM355_API int    pm355_memset(void * dst, int value, size_t bytes, pm355_stream_t stream);
/* One launch that copies up to 16 byte ranges out of ONE device staging block to their destinations: the per-token graph inputs of llama_set_inputs
 * (src/llama.cpp:17276-17420: token / embedding, positions, KQ mask row, output ids - four buffer set_tensor calls per token) travel as one pinned block + one
 * hipMemcpyAsync + this, instead of four pageable copies of ~19 us each. */
typedef struct pm355_scatter_seg { void * dst; uint32_t src_off; uint32_t bytes; } pm355_scatter_seg;
PM355_API int    pm355_scatter_bytes(const void * src_dev, const pm355_scatter_seg * segs, int n_segs, pm355_stream_t stream);

/* Asynchronous weight staging from host DRAM (upload.hip): pageable / mmap'd GGUF bytes -> ring of pinned chunks (filled by
 * `copy_threads` copier threads; -1 = PM355_UPLOAD_THREADS or 4) -> hipMemcpyAsync on a private stream -> (row-SoA types) repack into
 * the HBM layout, the three stages overlapped. Replaces the synchronous per-tensor ggml_backend_tensor_set(cur, mmap pointer, ...)
 * upload of the reference's loader (src/llama.cpp:5580-5600 -> ggml-cuda.cu:503-511); its own pinned async path (:5432-5520) is
 * disabled in this fork. pm355_upload returns once the last chunk is enqueued (the host bytes have been copied out);
 * pm355_uploader_sync waits for the device side. repack=1: `type` matrices of K-weight rows, nbytes = whole rows, dev = first
 * destination row in the HBM layout (pm355_row_stride apart); repack=0 or a type without re-ordering: plain bytes. */
typedef struct pm355_uploader pm355_uploader;
PM355_API pm355_uploader * pm355_uploader_new(size_t chunk_bytes /* 0 = 32 MiB */, int copy_threads);
PM355_API void   pm355_uploader_free(pm355_uploader * u);
PM355_API int    pm355_upload(pm355_uploader * u, int type, int64_t K, const void * host, void * dev, size_t nbytes, int repack);
PM355_API int    pm355_uploader_sync(pm355_uploader * u);
PM355_API uint64_t pm355_uploader_bytes(const pm355_uploader * u);

/* ---- weight layout ------------------------------------------------------------------------------ */
/* bytes of one row of K weights == ggml_row_size(type, K) (ggml/src/ggml.c:3579) */
PM355_API size_t pm355_row_size(int type, int64_t K);
/* row stride of the HBM layout: == pm355_row_size except Q6_K / Q8_0 with K % 2048 != 0 (row-SoA scale stream padded
 * to 16 B). A tensor of N rows occupies N * pm355_row_stride bytes in HBM (the plug-in's get_alloc_size). */
PM355_API size_t pm355_row_stride(int type, int64_t K);
/* Row-local re-ordering GGUF block order <-> HBM order (identity for F32/F16/Q4_K/Q5_K, row-SoA for
 * Q6_K/Q8_0; see prima_cpp_amd/csrc/repack.hip). src and dst are DEVICE pointers and must not overlap; the GGUF-order
 * side has rows pm355_row_size apart, the HBM side pm355_row_stride apart.
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
/* Single-token fusion used by the engine (and by the plug-in's graph pattern matcher): up to 3 weight matrices of the
 * same K that share ONE f32 activation row x (e.g. wq/wk/wv; at most two distinct quant types among Q4_K/Q5_K/Q6_K, or
 * all Q8_0) in ONE launch. The activation is quantized to the reference's vec_dot_type inside the kernel, bit-exactly
 * (quantize_row_q8_K / quantize_row_q8_0), after an optional fused rms_norm(x, eps) * norm_w
 * (ggml_compute_forward_rms_norm_f32 + mul). Per job: y = W.x (+bias) (+resid), or silu(W.x) * (W2.x) when W2 != NULL. */
typedef struct {
    int32_t type; int32_t pad_; int64_t N;
    const void * W; const void * W2;
    float * y; const float * bias; const float * resid;
} pm355_matvec_job;
PM355_API int pm355_mul_mat_vec_fused(const pm355_matvec_job * jobs, int njobs, int64_t K, const float * x_f32,
                                      const float * norm_w, float eps, pm355_stream_t stream);
/* 1 when this build stores the tail of a Q6_K row per group of 8 blocks - scales[8][16] | d[8] - (round 5, csrc/pm355_device.h), 0 for the
 * round-1..4 form scales[nb][16] | d[nb]; the streams in front of it (la | lb | qh) are the same. Tests build the expected HBM image from it. */
PM355_API int pm355_q6k_tail_grouped(void);
/* 1 in prima_cpp_amd/libprima_mi355_exp.so (-DPM_EXPERIMENTS=1): round 5's measured-slower forms of the decode layer - producer-side sums of squares
 * (pm355_mul_mat_vec_fused_ss with sumsq arguments), attention in the tail of the QKV launch (pm355_mul_mat_vec_qkv_attn), the persistent engine
 * (pm355_engine_run; PM355_SS / PM355_ATTN_TAIL / PM355_ENGINE for the resident window) - exist only there; in the product library (0) those entry
 * points return PM355_E_UNSUPPORTED and the switches do nothing. */
PM355_API int pm355_experiments_built(void);
/* Producer-side sum of squares (round 5). The rms_norm in front of wq | wk | wv and of ffn_gate | ffn_up reads a row that the preceding wo / ffn_down
 * launch has just written: sumsq_out != NULL (ONE job, W2 == NULL) makes every workgroup of that launch store the f64 sum of the f32-rounded squares
 * of the output rows it wrote - the terms of ggml_compute_forward_rms_norm_f32's `sum += (ggml_float)(x[i] * x[i])`, ggml.c:11975-11980 - into
 * sumsq_out[workgroup], pm355_mul_mat_vec_fused_grid() partials in all; sumsq_in != NULL (with norm_w) makes the consuming launch add those
 * n_sumsq_in (<= 256) partials instead of reducing the row in its prologue. Same Q8_K blocks as the plain form (tests: bit for bit). */
PM355_API int pm355_mul_mat_vec_fused_ss(const pm355_matvec_job * jobs, int njobs, int64_t K, const float * x_f32, const float * norm_w, float eps,
                                         double * sumsq_out, const double * sumsq_in, int n_sumsq_in, pm355_stream_t stream);
/* workgroups (= sumsq_out partials) of the launch pm355_mul_mat_vec_fused(_ss) issues for this job list, or a negative PM355_E_* */
PM355_API int pm355_mul_mat_vec_fused_grid(const pm355_matvec_job * jobs, int njobs, int64_t K);
/* 0 when pm355_mul_mat_vec_fused can serve this job list in ONE launch (type mix, K, LDS), else the error it would return */
PM355_API int pm355_mul_mat_vec_fused_check(const pm355_matvec_job * jobs, int njobs, int64_t K);
/* Batched (prefill) path, n_tokens >= 16: Y[t][n] = sum_k W[n][k] x[t][k] (+bias[n]) (+resid[t][n]) on the MFMA matrix cores
 * (v_mfma_f32_32x32x16_f16, weights dequantized on the fly into LDS, f32 accumulate; prima_cpp_amd/csrc/mmq.hip). Stands
 * in for the reference CUDA plug-in's mul_mat_q / dequantize+cuBLAS large-batch path (ggml-cuda/mmq.cuh:2583). */
PM355_API int pm355_mul_mat_q_mfma(int type, const void * W, int64_t K, int64_t N, const float * x, int64_t n_tokens, float * y,
                                   const float * bias, const float * resid, pm355_stream_t stream);
/* Several matrices that share the activations (wq | wk | wv: the three MUL_MATs of build_llama, src/llama.cpp:11065-11092; ffn_gate | ffn_up, llm_build_ffn
 * :9804) as ONE launch of the prompt GEMM (prima_cpp_amd/csrc/mmq_pf.hip), each job with its own quant type (at most two types per launch:
 * Llama-3-70B Q4_K_M has Q4_K wq / wk next to Q5_K or Q6_K wv, src/llama.cpp:19360-19373) and its own output y[n_tokens][N], bias, residual. Jobs the
 * kernel does not serve (Q8_0, K % 256 != 0) are launched one by one. njobs <= 4. */
typedef struct pm355_gemm_job { int32_t type; int32_t N; const void * W; float * y; const float * bias; const float * resid; } pm355_gemm_job;
PM355_API int pm355_mul_mat_q_mfma_multi(const pm355_gemm_job * jobs, int njobs, int64_t K, const float * x, int64_t n_tokens, pm355_stream_t stream);
/* ffn_gate | ffn_up of a prompt batch as one launch of PAIR tiles: y[t][n] = silu(Wgate . x) * (Wup . x) (llm_build_ffn LLM_FFN_SILU + LLM_FFN_PAR,
 * src/llama.cpp:9804-9890: MUL_MAT, MUL_MAT, SILU, MUL) - the gate result crosses from the waves that computed it to the waves that hold the matching
 * ffn_up rows inside the workgroup. Same type and N for both matrices; PM355_E_UNSUPPORTED when the prompt kernel does not serve the shape. */
PM355_API int pm355_mul_mat_q_mfma_pair(int type, const void * W_gate, const void * W_up, int64_t K, int64_t N, const float * x, int64_t n_tokens, float * y,
                                        pm355_stream_t stream);
/* Prompt-sized batches on the INTEGER matrix cores (v_mfma_i32_32x32x32_i8, prima_cpp_amd/csrc/mmq_big.hip): x is quantized to Q8_K
 * (quantize_row_q8_K, ggml-quants.c:3785) and multiplied with the CPU reference's own integer arithmetic (ggml_vec_dot_q4_K_q8_K /
 * ggml_vec_dot_q6_K_q8_K, ggml-quants.c:7713 / :8918): exact int32 sub-block sums, one f32 multiply-add per 256-weight super-block - the
 * design of the reference's CUDA plug-in for these batches (ggml-cuda/mmq.cuh:2583, quantize.cu:41-126). Q4_K / Q6_K weights, K % 1024 == 0;
 * any n_tokens (tiles of 128 tokens: meant for >= 128). pm355_mul_mat_q_i8_check: 0 when the shape is served. */
PM355_API int pm355_mul_mat_q_i8(int type, const void * W, int64_t K, int64_t N, const float * x, int64_t n_tokens, float * y,
                                 const float * bias, const float * resid, pm355_stream_t stream);
PM355_API int pm355_mul_mat_q_i8_check(int type, int64_t K, int64_t N, int64_t n_tokens);
/* Small batches, 1 <= n_tokens <= 64 (speculative decoding, parallel sequences, short prompts): the weights are streamed ONCE per 32
 * tokens and the products run on the integer matrix cores (v_mfma_i32_32x32x16_i8, prima_cpp_amd/csrc/mmq_i8.hip) with the
 * reference's own integer arithmetic - activations quantized to Q8_K, exact int32 block sums, one f32 multiply-add per 256-weight
 * super-block (ggml_vec_dot_q4_K_q8_K / ggml_vec_dot_q6_K_q8_K, ggml/src/ggml-quants.c; CUDA plug-in: ggml-cuda/mmq.cuh:2583).
 * Q4_K / Q5_K / Q6_K weights, and Q8_0 weights with Q8_0 activations (ggml_vec_dot_q8_0_q8_0, ggml-quants.c:5518: one 32-k matrix
 * instruction per block, K % 32 == 0). xq = n_tokens rows already quantized by pm355_quantize_q8_K (Q8_0: pm355_quantize_q8_0), or NULL
 * and x = f32 [n_tokens][K]. y / resid: [n_tokens][N]. pm355_mul_mat_q_small_check: 0 when the shape is served. */
PM355_API int pm355_mul_mat_q_small(int type, const void * W, int64_t K, int64_t N, const void * xq, const float * x, int64_t n_tokens,
                                    float * y, const float * bias, const float * resid, pm355_stream_t stream);
PM355_API int pm355_mul_mat_q_small_check(int type, int64_t K, int64_t N, int64_t n_tokens);
/* 2 or 3 matrices of ONE type and K that share the activations (wq | wk | wv, ffn_gate | ffn_up of build_llama, src/llama.cpp:11085-11160) in
 * one launch per 32 tokens, n_tokens <= 64, xq = rows quantized by pm355_quantize_q8_K; y[j]: [n_tokens][N[j]]; bias[j] may be NULL (bias may be NULL). */
PM355_API int pm355_mul_mat_q_small_multi(int type, int njobs, const void * const * W, const int64_t * N, float * const * y,
                                          const float * const * bias, const void * xq, int64_t K, int64_t n_tokens, pm355_stream_t stream);
/* The same with the LAST job of another K-quant type - wq | wk (Q4_K) and wv (Q6_K: n_tokens <= 32; Q5_K: <= 16), the layer composition of the Q4_K_M files
 * (llama_tensor_get_type, src/llama.cpp:19447) - as ONE grid whose workgroups are divided by weight bytes. PM355_E_UNSUPPORTED: launch the jobs separately. */
PM355_API int pm355_mul_mat_q_small_mixed(const int * types, int njobs, const void * const * W, const int64_t * N, float * const * y,
                                          const float * const * bias, const void * xq, int64_t K, int64_t n_tokens, pm355_stream_t stream);
/* test hook: additionally writes, per (row, unit), the exact int32 pair {sum scale*q_w*q_a, sum min*bsum} */
PM355_API int pm355_mul_mat_vec_q_dbg(int type, const void * W, int64_t K, int64_t N, const void * xq, float * y,
                                      int32_t * int_partials, int64_t * units_per_row, pm355_stream_t stream);

/* remaining per-layer ops (prima_cpp_amd/csrc/layer_ops.hip); positions are read from DEVICE memory (pos0[0] =
 * position of token 0 of the batch) so captured graphs can be replayed */
typedef struct {
    int32_t n_dims, mode /* 0 = NORM (llama), 2 = NEOX (qwen2) */, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
} pm355_rope_params;                     /* == ggml_rope_ext arguments (ggml/include/ggml.h:1497) */

/* replaces ggml_compute_forward_get_rows_{q,f32} (ggml/src/ggml.c:13288, :13414): out[t] = dequant(table[tokens[t]]) */
PM355_API int pm355_get_rows(int type, const void * table, int64_t K, const int32_t * d_tokens, int n_tokens,
                             float * out, pm355_stream_t stream);
/* replaces ggml_compute_forward_rope_f32 (ggml/src/ggml.c:14143) on Q and K + llm_build_kv_store
 * (src/llama.cpp:9673-9718): K -> F16 cache rows [n_ctx][Hkv*dh], V -> F16 transposed cache [Hkv*dh][n_ctx].
 * q_out may alias q. k_out_f32 optional (rotated K in f32, for tests / node equivalence). */
PM355_API int pm355_rope_kv_store(const float * q, const float * k, const float * v, float * q_out, float * k_out_f32,
                                  void * k_cache, void * v_cache, const int32_t * d_pos0, const float * freq_factors,
                                  int n_tokens, int n_head, int n_head_kv, int head_dim, int n_ctx,
                                  const pm355_rope_params * rp, pm355_stream_t stream);
/* replaces llm_build_kqv's MUL_MAT(k,q) -> SOFT_MAX_EXT -> MUL_MAT(v,kq) chain (src/llama.cpp:10062-10148) for
 * causal single-sequence decode; out: [n_tokens][n_head*head_dim] f32 */
PM355_API int pm355_attn_decode(const float * q, const void * k_cache, const void * v_cache, const int32_t * d_pos0,
                                float * out, int n_tokens, int n_head, int n_head_kv, int head_dim, int n_ctx,
                                float kq_scale, pm355_stream_t stream);
/* single-token variant of pm355_attn_decode for LONG contexts: the keys are split over n_ctx/256 x n_head_kv workgroups,
 * each serving the whole group of query heads of its KV head (prima_cpp_amd/csrc/attn_split.hip; the engine uses it beyond
 * PM355_ATTN_SPLIT_MIN = 320 positions). q_rot = the token's rotated queries (pm355_rope_kv_store output), the caches
 * already contain the token. scratch: pm355_attn_split_scratch_floats() floats of device memory. Position d_pos0[0] = index
 * of the token (it attends keys 0 .. d_pos0[0]). */
PM355_API size_t pm355_attn_split_scratch_floats(int n_head, int head_dim, int n_ctx);
PM355_API int pm355_attn_decode_split(const float * q_rot, const void * k_cache, const void * v_cache, const int32_t * d_pos0,
                                      float * out, float * scratch, int n_head, int n_head_kv, int head_dim, int n_ctx,
                                      float kq_scale, pm355_stream_t stream);
/* the same chain for a multi-token (prefill) batch on the MFMA matrix cores: causal, two passes over the keys so that p is
 * rounded to F16 after the division by the full row sum like the reference (prima_cpp_amd/csrc/attn_prefill.hip).
 * head_dim 64 or 128, n_ctx % 32 == 0. Same arguments and result layout as pm355_attn_decode. */
PM355_API int pm355_attn_prefill(const float * q, const void * k_cache, const void * v_cache, const int32_t * d_pos0,
                                 float * out, int n_tokens, int n_head, int n_head_kv, int head_dim, int n_ctx,
                                 float kq_scale, pm355_stream_t stream);
/* ggml-graph form of pm355_attn_prefill: the node chain MUL_MAT(k, q) SOFT_MAX(kq, KQ_mask, scale) MUL_MAT(v, kq) PERMUTE CONT of
 * llm_build_kqv (src/llama.cpp:10062-10148) for a multi-token batch. q = the ROTATED queries [n_tokens][n_head*head_dim] f32, the caches
 * already hold the batch; every query attends cells [0, n_kv) with its row of the additive F32 KQ_mask [n_kv per row, mask_stride floats
 * apart] (any mask the reference builds, not just causal). */
PM355_API int pm355_attn_prefill_masked(const float * q, const void * k_cache, const void * v_cache, const float * mask, int64_t mask_stride,
                                        float * out, int n_tokens, int n_head, int n_head_kv, int head_dim, int n_ctx, int n_kv,
                                        float kq_scale, pm355_stream_t stream);
/* the same for flash-attention graphs (FLASH_ATTN_EXT over a multi-token batch, llm_build_kqv src/llama.cpp:10075-10095):
 * flags = PM355_ATTN_V_ROWMAJOR (V cache [n_ctx][n_head_kv*head_dim]) | PM355_ATTN_MASK_F16 (mask rows are F16, mask_stride in elements) */
PM355_API int pm355_attn_prefill_masked_ex(const float * q, const void * k_cache, const void * v_cache, const void * mask, int64_t mask_stride,
                                           float * out, int n_tokens, int n_head, int n_head_kv, int head_dim, int n_ctx, int n_kv,
                                           float kq_scale, int flags, pm355_stream_t stream);
/* single-token fusion of the two entries above (rope on q,k + KV store + attention in ONE launch; what the engine
 * uses at decode). q/k/v are the raw projections of ONE token; the caches receive the new K row / V column. */
PM355_API int pm355_attn_rope_fused(const float * q, const float * k, const float * v, void * k_cache, void * v_cache,
                                    const int32_t * d_pos0, const float * freq_factors, float * out,
                                    int n_head, int n_head_kv, int head_dim, int n_ctx, float kq_scale,
                                    const pm355_rope_params * rp, pm355_stream_t stream);
/* ggml-graph form of pm355_attn_rope_fused / the split path, what the plug-in's graph_compute lowers the node chain
 *   ROPE(Qcur) ROPE(Kcur) CPY(K -> k_cache_view) CPY(V^T -> v_cache_view) MUL_MAT(k, q) SOFT_MAX(kq, KQ_mask, scale) MUL_MAT(v, kq) CONT
 * of llm_build_kv (src/llama.cpp:10167-10205) to for ONE token: the RoPE position comes from the graph's inp_pos tensor
 * (d_pos[0]); the cache cell the token is stored in and the number of cells attended (kv_self.head / kv_self.n,
 * src/llama.cpp:18433-18453) are read from DEVICE memory d_cell_nkv[0..1] so that a captured hipGraph can be replayed for the
 * next token; mask = row 0 of the F32 KQ_mask [n_kv] (0 / -inf, llama_set_inputs src/llama.cpp:17379-17420) or NULL.
 * split = 0: one workgroup per query head (max_keys bounds the cells attended, LDS); split = 1: keys split over workgroups, one launch
 * (flash-decoding with an in-launch merge, attn_flash.hip; max_keys = cells the grid covers, 0 = n_ctx; scratch =
 * pm355_attn_split_scratch_floats() floats, ZEROED once after allocation; PM355_ATTN_FLASH=0 selects the three-launch form). */
typedef struct {
    const float * q, * k, * v;           /* raw projections of the token: [n_head*head_dim], [n_head_kv*head_dim] x2 */
    void * k_cache, * v_cache;           /* F16 K rows [n_ctx][n_head_kv*head_dim]; F16 V transposed [n_head_kv*head_dim][n_ctx], or
                                          * (PM355_ATTN_V_ROWMAJOR) rows like K */
    const int32_t * d_pos;               /* device: RoPE position of the token */
    const int32_t * d_cell_nkv;          /* device int32[2]: {cache cell, cells attended} */
    const void * mask;                   /* device f32 (F16 with PM355_ATTN_MASK_F16) [cells attended] or NULL */
    const float * freq_factors;          /* rope_freqs [head_dim/2] or NULL */
    float * out;                         /* [n_head*head_dim] f32 */
    float * scratch;                     /* split only */
    int32_t n_head, n_head_kv, head_dim, n_ctx, split, max_keys;
    float kq_scale; int32_t flags;       /* PM355_ATTN_* */
} pm355_attn_token_args;
/* flash-attention graphs (llm_build_kv with cparams.flash_attn, src/llama.cpp:9705, :10075-10095: CPY(V -> row of v_cache),
 * FLASH_ATTN_EXT(q, k, v, F16 mask)): the V cache is row-major and the mask is F16; the kernel then uses the rounding points of
 * ggml_compute_forward_flash_attn_ext_f16 (q -> F16, probabilities stay f32) with an f32 accumulator. */
#define PM355_ATTN_V_ROWMAJOR 1
#define PM355_ATTN_MASK_F16   2
/* quantized KV cache (`-ctk q8_0` / `-ctv q8_0`, with flash attention): k_cache / v_cache hold native Q8_0 blocks (34 bytes per 32
 * values), rows like K; the token's K / V are quantized on store (quantize_row_q8_0_ref) and the query is quantized to Q8_0 for the
 * K.q products (ggml_vec_dot_q8_0_q8_0) as the reference CPU path does. One workgroup per head (no split path): max_keys bounds the
 * cells attended. Requires PM355_ATTN_V_ROWMAJOR. */
#define PM355_ATTN_K_Q8_0     4
#define PM355_ATTN_V_Q8_0     8
PM355_API int pm355_attn_token(const pm355_attn_token_args * a, const pm355_rope_params * rp, pm355_stream_t stream);
/* Single-token decode with the RoPE and the KV store in the EPILOGUE of the wq | wk | wv launch (NORM-mode rope - build_llama - and, with
 * rope_neox set, NEOX pairs (i, i + n_rot / 2) - build_qwen2, ggml.c:14238-14253 - where n_rot == head_dim and the row slices are powers of two):
 *   pm355_rope_table       this token's cos / sin per rotation pair, table[2 i] / table[2 i + 1], built with the reference's running
 *                          product theta_i = theta_{i-1} * theta_scale (ggml_rope_cache_init, ggml.c:14117-14131; rope_yarn :14094) - one
 *                          launch per token serves every layer
 *   pm355_mul_mat_vec_qkv  jobs = {wq, wk, wv}: rms_norm + Q8_K quantize + the three mat-vecs (+bias) like pm355_mul_mat_vec_fused, then
 *                          ROPE on adjacent row pairs (ggml.c:14224-14237), q rounded to F16 (the conversion MUL_MAT applies to src1 of
 *                          K.q, ggml.c:12445-12473) -> jobs[0].y as f32, k -> F16 row of cache cell d_cell_nkv[0] (or d_pos[0] when
 *                          d_cell_nkv == NULL), v -> F16 V cache (llm_build_kv_store, src/llama.cpp:9688-9716). jobs[1].y / jobs[2].y unused.
 *   pm355_attn_cached      MUL_MAT(k, q) -> SOFT_MAX_EXT -> MUL_MAT(v, kq) over cells [0, d_cell_nkv[1]) (or [0, d_pos[0]]) that are all in
 *                          the cache; same rounding points and mask / flags as pm355_attn_token's one-workgroup-per-head path. */
typedef struct {
    const float * rope_table; const int32_t * d_pos; const int32_t * d_cell_nkv; void * k_cache; void * v_cache;
    int32_t n_head_kv, head_dim, n_ctx, n_rot /* rope n_dims */, v_rowmajor, rope_neox /* rope mode & 2 */;
} pm355_qkv_store;
PM355_API int pm355_rope_table(const pm355_rope_params * rp, const int32_t * d_pos, const float * freq_factors, float * table, pm355_stream_t stream);
PM355_API int pm355_mul_mat_vec_qkv(const pm355_matvec_job * jobs, int64_t K, const float * x_f32, const float * norm_w, float eps,
                                    const pm355_qkv_store * s, pm355_stream_t stream);
/* the same with the rms_norm's sum of squares taken from n_sumsq_in producer-side partials (pm355_mul_mat_vec_fused_ss) */
PM355_API int pm355_mul_mat_vec_qkv_ss(const pm355_matvec_job * jobs, int64_t K, const float * x_f32, const float * norm_w, float eps,
                                       const pm355_qkv_store * s, const double * sumsq_in, int n_sumsq_in, pm355_stream_t stream);
/* The same launch with the attention over the cached cells in its TAIL (round 5; replaces the pm355_attn_cached launch that would follow - the
 * reference's MUL_MAT(k, q) -> SOFT_MAX_EXT -> MUL_MAT(v, kq) nodes, ggml.c:12445-12473 / :13783-13879 - same arithmetic, same bits): the workgroups
 * that hold the rows of one KV-head group take a ticket after their stores; the last n_head / n_head_kv of them wait for their group only and compute
 * one query head each over cells [0, d_pos[0]]. ticket: n_head_kv uint32 counters, zeroed once (monotonic across launches of one stream);
 * watchdog: optional int32, set non-zero if a bounded wait gave up. Needs d_pos (d_cell_nkv == NULL), the transposed V cache, no mask.
 * PM355_E_UNSUPPORTED when the grid does not split into power-of-two runs per KV head - then launch pm355_attn_cached as before. */
typedef struct { float * out; uint32_t * ticket; int32_t * watchdog; float kq_scale; int32_t n_head, max_keys; } pm355_qkv_attn;
PM355_API int pm355_mul_mat_vec_qkv_attn(const pm355_matvec_job * jobs, int64_t K, const float * x_f32, const float * norm_w, float eps,
                                         const pm355_qkv_store * s, const double * sumsq_in, int n_sumsq_in, const pm355_qkv_attn * attn, pm355_stream_t stream);
/* The persistent decode engine (round 5, prima_cpp_amd/csrc/decode_engine.hip): a list of PHASES - the launches above, in order - executed as ONE
 * launch: one workgroup per CU, a loader wave that streams every phase's weights into an LDS ring ahead of the consumers, a device-wide barrier
 * and write-through hand-offs between phases. pm355_model_step uses it for the whole layer stack of a single-token step; this entry runs an
 * arbitrary phase list (tests: every phase kind against the launch it stands for, bit for bit; ffn_down-sized rows to summation order).
 *   kind 0: pm355_mul_mat_vec_fused_ss (qkv == NULL) or pm355_mul_mat_vec_qkv_ss (qkv != NULL; d_cell_nkv must be NULL: engine mode); an rms_norm
 *           (norm_w != NULL) needs sumsq_in - the engine never reduces a row itself
 *   kind 1: pm355_attn_cached over cells [0, d_pos[0]] (no mask), transposed V cache
 * Returns 0, PM355_E_UNSUPPORTED when a phase is not served (types Q4_K / Q5_K / Q6_K, K % 256, rows per workgroup), PM355_E_HIP when the launch's
 * watchdog fired (a bounded wait gave up). Synchronizes the stream. */
typedef struct {
    int32_t kind, njobs; int64_t K;
    const pm355_matvec_job * jobs; const float * x_f32; const float * norm_w; float eps; int32_t n_sumsq_in;
    double * sumsq_out; const double * sumsq_in;
    const pm355_qkv_store * qkv;
    const float * q_rot; void * k_cache; void * v_cache; const int32_t * d_pos; float * out;
    int32_t n_head, n_head_kv, head_dim, n_ctx, max_keys; float kq_scale;
} pm355_engine_phase;
PM355_API int pm355_engine_run(const pm355_engine_phase * phases, int n_phases, pm355_stream_t stream);
/* 0 when pm355_mul_mat_vec_qkv can serve this wq | wk | wv list (types, K, every workgroup's row slices hold whole rotation pairs) */
PM355_API int pm355_mul_mat_vec_qkv_check(const pm355_matvec_job * jobs, int64_t K, int n_head_kv, int head_dim, int n_rot);
PM355_API int pm355_mul_mat_vec_qkv_check_ex(const pm355_matvec_job * jobs, int64_t K, int n_head_kv, int head_dim, int n_rot, int rope_neox);
PM355_API int pm355_attn_cached(const float * q_rot, void * k_cache, void * v_cache, const int32_t * d_pos, const int32_t * d_cell_nkv,
                                const void * mask, float * out, int n_head, int n_head_kv, int head_dim, int n_ctx, float kq_scale,
                                int max_keys, int flags, pm355_stream_t stream);
/* The same (llm_build_kqv's MUL_MAT(k, q) -> SOFT_MAX_EXT -> MUL_MAT(v, kq), src/llama.cpp:10032-10095; CUDA plug-in: fattn-vec-f16.cuh) over LONG
 * contexts (cells attended >= the caller's split threshold): the GQA group's scores and the P.V product on the matrix
 * cores, keys split over workgroups, partials merged in the launch (attn_flash_mfma.hip). F16 caches: transposed V (default graphs) or, with
 * PM355_ATTN_V_ROWMAJOR, the row-major V of the flash-attention graphs (the kernel transposes a wave's tile through LDS: ds_read_b64_tr_b16);
 * head_dim 64 / 128, <= 16 query heads per KV head. scratch = pm355_attn_split_scratch_floats() floats, zeroed once by the
 * caller; max_cells sizes the grid (>= cells attended; 0 = n_ctx). flags: PM355_ATTN_MASK_F16 | PM355_ATTN_V_ROWMAJOR. */
PM355_API int pm355_attn_cached_long(const float * q_rot, void * k_cache, void * v_cache, const int32_t * d_pos, const int32_t * d_cell_nkv,
                                     const void * mask, float * out, float * scratch, int n_head, int n_head_kv, int head_dim, int n_ctx,
                                     float kq_scale, int max_cells, int flags, pm355_stream_t stream);
/* 0 when pm355_attn_cached_long serves the shape (head_dim 64 / 128, <= 16 KV heads, <= 16 query heads per KV head, n_ctx % 8 == 0); callers
 * keep pm355_attn_token's split form otherwise (e.g. MHA models with 32 / 40 KV heads). No device work. */
PM355_API int pm355_attn_cached_long_check(int n_head, int n_head_kv, int head_dim, int n_ctx);
/* p[0] = a, p[1] = b on the stream (values travel as kernel arguments: no host buffer lifetime to manage) */
PM355_API int pm355_set_i32x2(int32_t * d_p, int32_t a, int32_t b, pm355_stream_t stream);
/* greedy sampler (src/llama-sampling.cpp:390-397): index of the first maximum */
PM355_API int pm355_argmax(const float * x, int64_t n, int32_t * d_index, float * d_value, pm355_stream_t stream);
/* ggml_compute_forward_add_f32 / mul_f32 (row-broadcast of b over a), silu(*u), scale */
PM355_API int pm355_add(const float * a, const float * b, float * y, int64_t n, int64_t nb, pm355_stream_t stream);
PM355_API int pm355_mul(const float * a, const float * b, float * y, int64_t n, int64_t nb, pm355_stream_t stream);
PM355_API int pm355_silu_mul(const float * g, const float * u, float * y, int64_t n, pm355_stream_t stream);
PM355_API int pm355_scale(const float * a, float * y, float s, int64_t n, pm355_stream_t stream);
/* synthetic weights generated directly in HBM (bench): random VALID blocks of `type`, |w| ~ scale */
PM355_API int pm355_fill_random_blocks(int type, void * dst, int64_t K, int64_t nrows, uint64_t seed, float scale,
                                       pm355_stream_t stream);

/* ---- (A') stride-aware node-equivalent ops (prima_cpp_amd/csrc/ggml_ops.hip) ------------------------------------------
 * What the plug-in falls back to for a graph node that is not covered by a fused fast path: arbitrary views / permutes /
 * broadcasts exactly as ggml describes them (ne[] elements, nb[] BYTE strides, struct ggml_tensor ggml/include/ggml.h:576). */
typedef struct { void * data; int32_t type; int32_t pad_; int64_t ne[4]; size_t nb[4]; } pm355_tensor;

/* ggml_compute_forward_dup (ggml.c:8238/:8509): F32/F16 -> F32/F16, same element count, any strides */
PM355_API int pm355_op_cpy(const pm355_tensor * src, const pm355_tensor * dst, pm355_stream_t stream);
/* op 0 = add (ggml.c:9002), 1 = mul (ggml.c:10077); b is broadcast over a */
PM355_API int pm355_op_binary(int op, const pm355_tensor * a, const pm355_tensor * b, const pm355_tensor * dst, pm355_stream_t stream);
/* op 0 = scale by param (ggml.c:11262), 1 = silu (ggml.c:11581) */
PM355_API int pm355_op_unary(int op, const pm355_tensor * a, const pm355_tensor * dst, float param, pm355_stream_t stream);
PM355_API int pm355_op_rms_norm(const pm355_tensor * a, const pm355_tensor * dst, float eps, pm355_stream_t stream);
/* ggml_compute_forward_soft_max_f32 (ggml.c:13783): mask F32/F16 [ne0, >= ne1] or NULL */
PM355_API int pm355_op_soft_max(const pm355_tensor * a, const pm355_tensor * mask, const pm355_tensor * dst, float scale,
                                float max_bias, pm355_stream_t stream);
/* ggml_compute_forward_flash_attn_ext_f16 (ggml.c:15538): q F32 [D, N, H], k / v F16 [D, n_kv, Hkv] (V not transposed), mask F16
 * [n_kv, >= N] or NULL, dst F32 [D, H, N]; scale, ALiBi max_bias, logit softcap as in ggml_flash_attn_ext (ggml.h:1757) */
PM355_API int pm355_op_flash_attn_ext(const pm355_tensor * q, const pm355_tensor * k, const pm355_tensor * v, const pm355_tensor * mask,
                                      const pm355_tensor * dst, float scale, float max_bias, float logit_softcap, pm355_stream_t stream);
PM355_API int pm355_op_rope(const pm355_tensor * a, const int32_t * d_pos, const float * freq_factors, const pm355_tensor * dst,
                            const pm355_rope_params * rp, pm355_stream_t stream);
/* MUL_MAT with F16 / F32 src0 (the K.q and V.p products of llm_build_kqv): src1 F32 rounded to F16 when src0 is F16; or with a Q8_0 src0 in
 * ggml's native 34-byte blocks (a view of a `-ctk q8_0` cache without flash attention): src1 rows quantized to Q8_0 (quantize_row_q8_0_ref,
 * ggml-quants.c:848), integer dots x d_k x d_q (ggml_vec_dot_q8_0_q8_0, :5518) */
PM355_API int pm355_op_mul_mat_f(const pm355_tensor * a, const pm355_tensor * b, const pm355_tensor * dst, pm355_stream_t stream);
PM355_API int pm355_op_get_rows_f32(const pm355_tensor * a, const int32_t * d_idx, int64_t n_idx, const pm355_tensor * dst,
                                    pm355_stream_t stream);

/* ---- (B) engine: resident decoder for one layer window --------------------------------------------- */
typedef struct pm355_model pm355_model;

typedef struct {                          /* llm_load_hparams (src/llama.cpp:5823) subset */
    int32_t arch;                         /* 0 = llama (rope NORM, optional rope_freqs), 1 = qwen2 (rope NEOX, qkv bias) */
    int32_t n_layer, n_embd, n_head, n_head_kv, head_dim, n_ff, n_vocab, n_ctx, n_ctx_orig;
    float   rms_eps, rope_freq_base, rope_freq_scale;
    int32_t pad_;
} pm355_hparams;

enum pm355_tensor_kind { PM355_T_ATTN_NORM = 0, PM355_T_WQ, PM355_T_WK, PM355_T_WV, PM355_T_WO, PM355_T_FFN_NORM,
                         PM355_T_FFN_GATE, PM355_T_FFN_UP, PM355_T_FFN_DOWN, PM355_T_BQ, PM355_T_BK, PM355_T_BV,
                         PM355_T_TOK_EMBD, PM355_T_OUT_NORM, PM355_T_OUTPUT, PM355_T_ROPE_FREQS, PM355_T_COUNT };
enum { PM355_HAS_EMBD = 1, PM355_HAS_HEAD = 2 };

/* A model holds the tensors of layers [layer_lo, layer_hi) — the window prima.cpp's piped-ring assigns to this
 * rank (this_layer_is_mine, src/llama.cpp:3838-3852) — plus tok_embd (HAS_EMBD) and output_norm/output (HAS_HEAD). */
PM355_API pm355_model * pm355_model_new(const pm355_hparams * hp, int layer_lo, int layer_hi, int flags);
PM355_API void pm355_model_free(pm355_model * m);
PM355_API const char * pm355_model_error(pm355_model * m);
/* ggml_backend_tensor_set semantics with async staging: host bytes in GGUF block order are streamed through
 * pinned double buffers (hipMemcpyAsync) and re-ordered into the HBM layout on the device
 * (replaces load_all_data's synchronous upload, src/llama.cpp:5418-5640). layer = -1 for per-model tensors. */
PM355_API int pm355_model_set_tensor(pm355_model * m, int kind, int layer, int type, const void * host_data, size_t nbytes);
/* WINDOW STREAMING for windows larger than HBM (call before the first layer tensor is set): the layer tensors are kept in pinned
 * host memory (already in the HBM layout) and cycled through n_slots device-side layer slots by a copy stream (hipMemcpyAsync),
 * one layer ahead per free slot, while the compute stream works on the current layer - replaces prima.cpp's mmap prefetch /
 * release of the next layer window (manage_graph_tensors + posix_madvise, src/llama.cpp:18152-18218, :18566-18575).
 * n_slots = 0: everything resident (default). Token embedding, head and KV caches always stay resident. */
PM355_API int pm355_model_set_streaming(pm355_model * m, int n_slots);
PM355_API uint64_t pm355_model_streamed_bytes(const pm355_model * m);   /* host -> device bytes streamed so far */
/* synthetic tensor generated in HBM (no host traffic) */
PM355_API int pm355_model_fill_tensor(pm355_model * m, int kind, int layer, int type, uint64_t seed, float scale);
/* allocate KV cache (F16, zero-cleared: llama_kv_cache_init src/llama.cpp:3889-3992) + scratch for max_tokens per call */
PM355_API int pm355_model_finalize(pm355_model * m, int max_tokens);
/* same with n_seq independent sequences (one KV slab + one position counter each): the sequences that are in flight
 * around the piped ring at the same time */
PM355_API int pm355_model_finalize_seqs(pm355_model * m, int max_tokens, int n_seq);
PM355_API size_t pm355_model_weight_bytes(const pm355_model * m);     /* matmul weight bytes read per token */
PM355_API size_t pm355_model_kv_bytes_per_pos(const pm355_model * m); /* KV bytes read per cached position */
PM355_API int pm355_model_kv_clear(pm355_model * m, pm355_stream_t stream);
PM355_API void * pm355_model_kv_ptr(pm355_model * m, int layer, int which /*0=K,1=V*/);
PM355_API void * pm355_model_tensor_ptr(pm355_model * m, int kind, int layer, int * type_out);  /* HBM-layout device pointer */
/* One pass of the window over n_tokens tokens at positions pos0.. (causal, single sequence).
 *   input : d_tokens (int32, needs HAS_EMBD) or d_x_in (f32 [n_tokens][n_embd], the activation handed over by the
 *           previous rank: llama_recv_tensors src/llama.cpp:18054)
 *   output: d_x_out (f32 [n_tokens][n_embd], residual stream after layer_hi-1; may be NULL), and with HAS_HEAD
 *           d_logits (f32 [n_vocab], LAST token) / d_argmax (int32) when non-NULL.
 * Equivalent of the per-sub-graph body of llama_decode_internal's ring loop (src/llama.cpp:18503-18564). */
PM355_API int pm355_model_decode(pm355_model * m, const int32_t * d_tokens, const float * d_x_in, int n_tokens, int pos0,
                                 float * d_x_out, float * d_logits, int32_t * d_argmax, pm355_stream_t stream);
/* the same for sequence `seq` (its KV slab and position counter) of a window finalized with n_seq > 1; `seq` stays the current sequence */
PM355_API int pm355_model_decode_seq(pm355_model * m, int seq, const int32_t * d_tokens, const float * d_x_in, int n_tokens, int pos0,
                                     float * d_x_out, float * d_logits, int32_t * d_argmax, pm355_stream_t stream);
PM355_API int pm355_model_n_embd(const pm355_model * m);
PM355_API int pm355_model_n_seq(const pm355_model * m);          /* KV slabs the window was finalized for (pm355_model_finalize_seqs) */
/* Launch geometry of the prompt GEMM (pm355_mul_mat_q_mfma and friends) for n_row_tiles tiles of 256 weight rows (pair launches: 128 rows of each matrix) x
 * n_tokens x K on a device of n_cus CUs in 8 XCDs - host arithmetic only, no device needed (the reference's counterpart is the stream-k / tile choice of
 * ggml-cuda/mmq.cuh:2583-2700). out5 = {tokens per tile (64 / 128 / 256), token tiles, K slices of a split tile, whole-tile slots per XCD in front of the split
 * ones (0: every tile is split, or none when slices == 1), workgroups launched}. allow_tail_split = 0: the uniform split only. */
PM355_API int pm355_gemm_plan(int64_t n_row_tiles, int64_t K, int64_t n_tokens, int n_cus, int allow_tail_split, int32_t * out5);
/* Largest batch the library routes to the integer small-batch mat-mul (pm355_mul_mat_q_small: the CPU backend's Q8_K arithmetic, ggml/src/ggml.c:12377); larger
 * batches take the F16 prompt GEMM (pm355_mul_mat_q_mfma). 32 unless PM355_MMQ_MAX_TOKENS = 16 .. 64 says otherwise. */
PM355_API int pm355_small_batch_max_tokens(void);
/* Device-resident greedy loop (needs HAS_EMBD|HAS_HEAD and the whole model in one window): starting from the token
 * in d_tokens_io[0] at position pos0, generate n_steps tokens; step i reads d_tokens_io[i], writes d_tokens_io[i+1].
 * One captured hipGraph per step, replayed; no host synchronisation inside. */
/* 0, or an error once a persistent decode kernel's device-wide barrier timed out (results since then are invalid; set
 * PM355_PERSISTENT=0 to use the 5-launches-per-layer path). Synchronizes the device. */
PM355_API int pm355_model_check(pm355_model * m);
PM355_API int pm355_model_generate(pm355_model * m, int32_t * d_tokens_io, int pos0, int n_steps, int use_graph,
                                   pm355_stream_t stream);
/* single-token step with the position held in device memory (graph-replayable building block of the piped ring):
 * x_in/x_out as above, position = internal device counter set by pm355_model_set_pos and advanced by `advance`. */
PM355_API int pm355_model_set_pos(pm355_model * m, int pos, pm355_stream_t stream);       /* sequence 0, made current */
PM355_API int pm355_model_set_seq_pos(pm355_model * m, int seq, int pos, pm355_stream_t stream);
PM355_API int pm355_model_set_seq(pm355_model * m, int seq, pm355_stream_t stream);       /* choose the current sequence */
/* result_norm + lm_head (+ greedy argmax) on one hidden row: build_llama's last sub-graph (src/llama.cpp:11191-11215),
 * which rank 0 runs on the activation returned by the last rank of the ring */
PM355_API int pm355_model_head(pm355_model * m, const float * d_x_row, float * d_logits, int32_t * d_argmax,
                               pm355_stream_t stream);
/* step with ring controls: after the window, position[current seq] += advance and current seq = (seq + rotate) % n_seq.
 * head_first != 0 (rank 0 of the ring): first apply the head to d_x_in (the last rank's activation) writing d_argmax /
 * d_logits, then embed the token at d_token (may alias d_argmax) and run the window into d_x_out. */
PM355_API int pm355_model_step_ex(pm355_model * m, const int32_t * d_token, const float * d_x_in, float * d_x_out,
                                  float * d_logits, int32_t * d_argmax, int advance, int rotate_seq, int head_first,
                                  int use_graph, pm355_stream_t stream);
PM355_API int pm355_model_step(pm355_model * m, const int32_t * d_token, const float * d_x_in, float * d_x_out,
                               float * d_logits, int32_t * d_argmax, int advance, int use_graph, pm355_stream_t stream);

/* ---- (C) piped-ring transport over RCCL (prima_cpp_amd/csrc/ring.hip) ---------------------------------------------------
 * Replaces llama_send_tensors / llama_recv_tensors (src/llama.cpp:18031-18077, ZeroMQ) and the D2H / H2D bounce of the ring loop
 * (src/llama.cpp:18503-18564): one rank per GPU, activations handed neighbour to neighbour with ncclSend / ncclRecv on a dedicated
 * communication stream, event hand-off to the compute stream, no host wait per micro-step. RCCL is dlopen()ed on first use. */
typedef struct pm355_ring pm355_ring;
PM355_API const char * pm355_ring_error(void);
/* rank 0: 128-byte RCCL unique id (ncclGetUniqueId) to be distributed to the other ranks by the launcher */
PM355_API int pm355_ring_unique_id(void * id128);
/* ncclCommInitRank on the current device + communication stream + events; NULL on failure (pm355_ring_error) */
PM355_API pm355_ring * pm355_ring_init(const void * id128, int rank, int world);
PM355_API void pm355_ring_free(pm355_ring * r);
PM355_API int pm355_ring_rank(const pm355_ring * r);
PM355_API int pm355_ring_world(const pm355_ring * r);
/* ncclGroupStart; ncclSend(send -> next rank); ncclRecv(recv <- previous rank); ncclGroupEnd on the communication stream, after
 * everything enqueued so far on compute_stream. send / recv: n f32 each, either may be NULL. Asynchronous. */
PM355_API int pm355_ring_exchange(pm355_ring * r, const float * send, float * recv, int64_t n, pm355_stream_t compute_stream);
/* compute_stream waits ON THE DEVICE for the last exchange (its input has arrived, the previous send buffer is free again) */
PM355_API int pm355_ring_wait(pm355_ring * r, pm355_stream_t compute_stream);
/* one micro-step of a rank = body of the reference's ring loop: pm355_ring_wait -> pm355_model_step_ex -> pm355_ring_exchange
 * {send x_out to the next rank when do_send, receive the NEXT micro-step's input into recv_next when non-NULL} */
PM355_API int pm355_ring_step(pm355_ring * r, pm355_model * m, const int32_t * d_token, const float * x_in, float * x_out,
                              float * d_logits, int32_t * d_argmax, int advance, int rotate, int head_first, int use_graph,
                              int do_send, float * recv_next, int64_t n_embd, pm355_stream_t compute_stream);
/* A ring whose exchanges go through the CALLER's transport instead of RCCL (tests: two ranks on one GPU over gloo; any other fabric):
 * exchange(user, send, n_send, recv, n_recv, stream) moves n_send f32 from device memory `send` to the next rank and n_recv f32 from
 * the previous rank into device memory `recv` (either may be NULL / 0); wait(user, stream) returns once the last exchange is complete
 * (for the compute stream). Everything above the transport - pm355_ring_step_tokens, pm355_ring_prefill, pm355_ring_single_token -
 * is the same code for both. */
typedef int (*pm355_ring_exchange_fn)(void * user, const float * send, int64_t n_send, float * recv, int64_t n_recv, pm355_stream_t stream);
typedef int (*pm355_ring_wait_fn)(void * user, pm355_stream_t stream);
PM355_API pm355_ring * pm355_ring_init_cb(int rank, int world, pm355_ring_exchange_fn exchange, pm355_ring_wait_fn wait, void * user);
/* the general exchange: different element counts in the two directions (the last rank returns one row, a window hands on a ubatch) */
PM355_API int pm355_ring_exchange2(pm355_ring * r, const float * send, int64_t n_send, float * recv, int64_t n_recv, pm355_stream_t compute_stream);
/* Multi-token micro-step: the window of sequence `seq` over n_tokens tokens at positions pos0.. (tokens on the first rank, x_in
 * [n_tokens][n_embd] elsewhere - the reference hands the whole ubatch [n_embd, n_tokens] per hop, llama_send_tensors src/llama.cpp:18031-18052),
 * then the exchange {send n_send f32 from send_ptr (inside x_out) when non-NULL, receive n_recv f32 into recv_next when non-NULL} */
PM355_API int pm355_ring_step_tokens(pm355_ring * r, pm355_model * m, int seq, const int32_t * d_tokens, const float * x_in, float * x_out,
                                     int n_tokens, int pos0, const float * send_ptr, int64_t n_send, float * recv_next, int64_t n_recv,
                                     pm355_stream_t compute_stream);
/* Prompt processing pipelined over the ranks (what prima.cpp does one ubatch at a time with every other rank idle, src/llama.cpp:18503-18564):
 * the n_seq prompts of n_prompt tokens are cut into chunks of `ubatch` tokens; chunk g runs on rank r at pipeline step g + r while chunk g + 1
 * runs on rank r - 1; a window hands its [n_tokens][n_embd] output to the next rank; the last rank returns the LAST row of a prompt's final
 * chunk to rank 0 (final_rows[seq][n_embd], rank 0 only) for the head. d_tokens: [n_seq][n_prompt] int32 on rank 0 (NULL elsewhere).
 * Every rank calls it with the same n_seq / n_prompt / ubatch; the window must be finalized with max_tokens >= ubatch, n_seq >= n_seq. Positions
 * of the sequences are left at n_prompt. Asynchronous on compute_stream (pm355_ring_wait before reading final_rows). */
PM355_API int pm355_ring_prefill(pm355_ring * r, pm355_model * m, int n_seq, const int32_t * d_tokens, int n_prompt, int ubatch,
                                 float * final_rows, pm355_stream_t compute_stream);
/* ONE sequence in flight, the reference's own mode (rank 0 blocked until the token has been round the ring, src/llama.cpp:18509): one token of
 * sequence `seq` per call on every rank. Rank 0: embeds *d_token, runs its window, sends, receives the last rank's row, runs the head and writes
 * the next token to *d_token (and the logits to d_logits when non-NULL); other ranks: receive, window, send. */
PM355_API int pm355_ring_single_token(pm355_ring * r, pm355_model * m, int seq, int32_t * d_token, float * d_logits, pm355_stream_t compute_stream);
/* The staggered multi-sequence decode loop in C (what produces the N-GPU aggregate of bench.py --gpus N): `world` sequences in flight one rank
 * apart - at micro-step m rank r works on sequence (m - r) mod world - where the reference keeps one batch in flight and rank 0 blocks until the
 * token has been round the ring (llama_decode_internal, src/llama.cpp:18503-18564). The window must be finalized with n_seq == world. Per
 * micro-step: wait (device-side) for the previous exchange -> window step (rank 0: head on the returned activation -> argmax -> embed -> window,
 * one captured graph) -> grouped exchange. Nothing blocks the host. forced: host array [n_micro] for rank 0 (token to feed instead of the head's
 * argmax, < 0 = none; required for the first `world` micro-steps after a reset), else NULL. d_tokens_out: device int32 [n_micro] on rank 0
 * (token fed at each micro-step), or NULL. reset != 0 restarts the schedule at micro-step 0 AND sets the model's sequence counter back to sequence 0
 * (pm355_model_set_seq(m, 0): the micro-step index selects the KV slab on every rank); positions are the caller's (pm355_model_set_seq_pos). Finish with
 * pm355_ring_wait.
 * Round 6 - two sequences per rank: a window finalized with n_seq == 2 * world runs the two-deep schedule: rank r works on sequence (m - 2 r) mod (2 world)
 * at micro-step m, the row it sends after step m is consumed by its successor at step m + 2, so every hop travels under the other round's compute instead
 * of in front of it (the exchange of step m signals its own event, the window of step m + 2 waits for that one only); the first 2 * world micro-steps of
 * rank 0 need forced tokens. */
PM355_API int pm355_ring_decode_staggered(pm355_ring * ring, pm355_model * m, int n_micro, const int32_t * forced, int32_t * d_tokens_out, int reset,
                                          int use_graph, pm355_stream_t compute_stream);
PM355_API const float * pm355_ring_decode_last_output(const pm355_ring * ring);
/* a ring of world size 1 without a communicator (the schedules above on a single window) */
PM355_API pm355_ring * pm355_ring_init_local(void);

#ifdef __cplusplus
}
#endif
#endif
