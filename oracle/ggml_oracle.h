// oracle/ggml_oracle.h — CPU restatement of the reference's quantized decode hot path.
//
// TEST INFRASTRUCTURE ONLY (see ggml_oracle.c header). Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load liboracle.so; the product (prima_cpp_amd/) never does.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

// ggml type ids (reference: ggml/include/ggml.h:356-372)
enum { ORC_F32 = 0, ORC_F16 = 1, ORC_Q8_0 = 8, ORC_Q4_K = 12, ORC_Q5_K = 13, ORC_Q6_K = 14, ORC_Q8_K = 15 };

uint16_t orc_f32_to_f16(float f);
float    orc_f16_to_f32(uint16_t h);

int64_t orc_row_size(int type, int64_t k);       // bytes of one row of k elements
int     orc_vec_dot_type(int type);              // activation format a weight type is dotted with

void  orc_quantize_row_q8_K(const float * x, void * y, int64_t k);
void  orc_quantize_row_q8_0(const float * x, void * y, int64_t k);
void  orc_dequantize_row(int type, const void * x, float * y, int64_t k);

float orc_vec_dot(int type, int64_t n, const void * w, const void * a);
// exact integer partials per 256-super-block (K-quants) / per 32-block (Q8_0):
//   isum[b] = sum_j scale_j * sum_l q_w * q_a   (what multiplies d_w*d_a)
//   msum[b] = sum_j min_j * bsum_j              (what multiplies dmin_w*d_a; 0 for Q6_K/Q8_0)
void  orc_vec_dot_int_partials(int type, int64_t n, const void * w, const void * a, int32_t * isum, int32_t * msum);

void  orc_mul_mat(int type, const void * W, int64_t K, int64_t N, const float * x, int64_t ncols, float * out);
void  orc_rms_norm(const float * x, const float * w, int64_t n, int64_t rows, float eps, float * out);
void  orc_rope(const float * x, int64_t d, int64_t heads, int64_t ntok, const int32_t * pos,
               const float * freq_factors, int n_dims, int mode, int n_ctx_orig,
               float freq_base, float freq_scale, float ext_factor, float attn_factor,
               float beta_fast, float beta_slow, float * out);
void  orc_soft_max_ext(const float * x, const float * mask, int64_t nc, int64_t nr, int64_t heads,
                       float scale, float max_bias, float * out);
void  orc_silu_mul(const float * g, const float * u, int64_t n, float * out);

typedef struct { int32_t type; int32_t pad_; const void * data; } orc_tensor_t;

// Same field order as ref_model_desc (oracle/ref_ops.c) and pm355_model_desc (include/prima_mi355.h).
typedef struct {
    int32_t arch;            // 0 = llama, 1 = qwen2
    int32_t n_layer, n_embd, n_head, n_head_kv, head_dim, n_ff, n_vocab, n_ctx, n_ctx_orig;
    float   rms_eps, rope_freq_base, rope_freq_scale;
    int32_t pad_;
    const orc_tensor_t * attn_norm, * wq, * wk, * wv, * wo, * ffn_norm, * ffn_gate, * ffn_up, * ffn_down;
    const orc_tensor_t * bq, * bk, * bv;
    orc_tensor_t tok_embd, out_norm, output;
    const float * rope_freqs;
} orc_model_desc;

void * orc_model_new(const orc_model_desc * d);
void   orc_model_free(void * m);
void   orc_model_kv_clear(void * m);
int    orc_model_eval(void * m, const int32_t * tokens, const float * embd_in, int n_tokens, int pos0,
                      int layer_lo, int layer_hi, int with_head,
                      float * hidden_out, float * logits_out, int n_threads);
const void * orc_model_kv_ptr(void * m, int il, int which);

#ifdef __cplusplus
}
#endif
