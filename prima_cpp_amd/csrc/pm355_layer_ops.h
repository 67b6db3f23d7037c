// pm355_layer_ops.h — host launchers of layer_ops.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/prima_mi355.h"

struct pm_rope_cfg {
    int n_dims, mode /*0 = NORM, 2 = NEOX*/, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
    float theta_scale, corr0, corr1;     // filled by pm_rope_params()
};
void pm_rope_params(pm_rope_cfg & c);

void pm_launch_embed(int type, const void * table, int K, const int32_t * tokens, int n_tok, float * out, hipStream_t st);
void pm_launch_rope_kv_store(const float * q, const float * k, const float * v, float * q_out, float * k_out_f32,
                             void * kc, void * vc, const int32_t * pos0, const int32_t * seq, long seq_stride,
                             const float * freq_factors,
                             int n_tok, int H, int Hkv, int dh, int n_ctx, const pm_rope_cfg & c, hipStream_t st,
                             int round_q = 0);       // q_out rounded to F16 (what MUL_MAT(k, q) does to its src1): the form pm_launch_attn_cached reads
int  pm_launch_attn_decode(const float * q, const void * kc, const void * vc, const int32_t * pos0,
                           const int32_t * seq, long seq_stride, float * out,
                           int n_tok, int H, int Hkv, int dh, int n_ctx, float scale, hipStream_t st);
// long-context single-token attention, keys split over workgroups (attn_split.hip). rope == nullptr: q = ROTATED queries and the
// caches are already updated; else q / k / v are the raw projections and rope + KV store happen inside the first kernel
size_t pm_attn_split_scratch_floats(int H, int dh, int n_ctx);
int  pm_launch_attn_split(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * pos0, const int32_t * seq,
                          long seq_stride, const float * freq_factors, float * out, float * scratch, int H, int Hkv, int dh, int n_ctx,
                          float scale, const pm_rope_cfg * rope, hipStream_t st, const int32_t * dyn = nullptr, const void * mask = nullptr,
                          int v_rowmajor = 0, int mask_f16 = 0);
// long-context single-token attention in ONE launch (flash-decoding with an in-launch merge, attn_flash.hip); q / k / v raw projections
size_t pm_attn_flash_scratch_floats(int H, int Hkv, int dh, int n_ctx);
int  pm_attn_flash_cached_ok(int H, int Hkv, int dh, int n_ctx);      // 0: served by the matrix-core long-context kernel
int  pm_launch_attn_flash_cached(const float * q_rot, void * kc, void * vc, const int32_t * pos0, const int32_t * seq, long seq_stride, float * out,
                                 float * scratch, int H, int Hkv, int dh, int n_ctx, float scale, hipStream_t st, const int32_t * dyn,
                                 const void * mask, int mask_f16, int max_cells, int v_rowmajor = 0, int k_q8 = 0, int v_q8 = 0);
// where the long-context Q8_0 path parks the token's rotated (and, for a Q8_0 K cache, Q8_0-quantized) query rows inside the attention scratch
size_t pm_attn_flash_qrot_offset(int H, int Hkv, int dh, int n_ctx);
int  pm_launch_attn_flash(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * pos0, const int32_t * seq,
                          long seq_stride, const float * freq_factors, float * out, float * scratch, int H, int Hkv, int dh, int n_ctx,
                          float scale, const pm_rope_cfg & c, hipStream_t st, const int32_t * dyn = nullptr, const void * mask = nullptr,
                          int v_rowmajor = 0, int mask_f16 = 0, int max_cells = 0);
// causal multi-token attention on MFMA (attn_prefill.hip); -1: unsupported shape (head_dim 64/128, n_ctx % 32 == 0)
int  pm_launch_attn_prefill(const float * q, const void * kc, const void * vc, const int32_t * pos0, const int32_t * seq,
                            long seq_stride, float * out, int n_tok, int H, int Hkv, int dh, int n_ctx, float scale, hipStream_t st,
                            const float * mask = nullptr, long mask_stride = 0, int n_kv = 0,   // mask != null: ggml-graph mode (cells [0, n_kv) + additive mask row per token)
                            int v_rowmajor = 0, int mask_f16 = 0,                              // flash-attention graphs: row-major V cache, F16 mask rows
                            void * out_f16 = nullptr);                                         // result as F16 (activations of the wo GEMM) instead of `out`
int  pm_launch_attn_rope_fused(const float * q, const float * k, const float * v, void * kc, void * vc,
                                const int32_t * pos0, const int32_t * seq, long seq_stride, const float * freq_factors,
                                float * out, int H, int Hkv, int dh, int n_ctx, float scale, const pm_rope_cfg & c, hipStream_t st,
                                const int32_t * dyn = nullptr, const void * mask = nullptr, int max_keys = 0, int v_rowmajor = 0, int mask_f16 = 0);
// decode path with the RoPE + KV-store epilogue in the wq | wk | wv launch (mmvq_device.h QkvEpi): the per-token cos / sin table
// (tab[2 i], tab[2 i + 1] for rotation pair i; position = pos[seq ? *seq : 0]) and the attention over cells that are all cached
// (q = rotated, F16-rounded; attn_cached.hip). Remaining arguments as pm_launch_attn_rope_fused.
void pm_launch_rope_table(const pm_rope_cfg & c, const int32_t * pos, const int32_t * seq, const float * freq_factors, float * tab, hipStream_t st,
                          int n_tok = 1);    // n_tok > 1: tables of the tokens pos .. pos + n_tok - 1, n_dims floats each
int  pm_launch_attn_cached(const float * q, void * kc, void * vc, const int32_t * pos0, const int32_t * seq, long seq_stride, float * out,
                           int H, int Hkv, int dh, int n_ctx, float scale, hipStream_t st, const int32_t * dyn = nullptr,
                           const void * mask = nullptr, int max_keys = 0, int v_rowmajor = 0, int mask_f16 = 0,
                           int n_tok = 1);          // n_tok > 1 (engine mode): token t of a small batch attends cells [0, pos0 + t], q / out rows t
// ggml-graph mode of the two launchers above: dyn = device int32[2] {cache cell the token is stored in, cells attended}, the RoPE
// position is pos0[0], mask = additive KQ-mask row [cells attended] (f32, or F16 with mask_f16) or null; v_rowmajor = the V cache is
// [n_ctx][n_embd_v_gqa] (flash-attention graphs) instead of transposed, with flash-attention rounding points; max_keys (> 0) sizes the fused kernel's LDS score
// buffer instead of n_ctx (the caller guarantees cells attended <= max_keys)
void pm_launch_set_i32x2(int32_t * p, int a, int b, hipStream_t st);
// quantized (Q8_0) KV cache, native 34-byte blocks (attn_q8.hip)
int  pm_launch_cpy_f32_q8_0(const pm355_tensor * src, void * dst, hipStream_t st);
int  pm_launch_flash_attn_ext_q8(const pm355_tensor * q, const pm355_tensor * k, const pm355_tensor * v, const pm355_tensor * mask,
                                 const pm355_tensor * dst, float scale, float softcap, hipStream_t st);
int  pm_launch_q8_token_prep(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * pos, const int32_t * dyn, const float * ff,
                             float * q_rot, int H, int Hkv, int dh, int n_ctx, const pm_rope_cfg & c, int k_q8, int v_q8, hipStream_t st);
int  pm_launch_attn_q8_token(const float * q, const float * k, const float * v, void * kc, void * vc, const int32_t * pos, const int32_t * dyn,
                             const void * mask, int mask_f16, const float * ff, float * out, int H, int Hkv, int dh, int n_ctx, float scale,
                             const pm_rope_cfg & c, int k_q8, int v_q8, int max_keys, hipStream_t st);
void pm_launch_argmax(const float * x, int n, int32_t * idx, float * val, hipStream_t st);
void pm_launch_add(const float * a, const float * b, float * y, long n, long nb, hipStream_t st);
void pm_launch_mul(const float * a, const float * b, float * y, long n, long nb, hipStream_t st);
void pm_launch_silu_mul(const float * g, const float * u, float * y, long n, hipStream_t st);
void pm_launch_scale(const float * a, float * y, float s, long n, hipStream_t st);
void pm_launch_set_i32(int32_t * p, int v, hipStream_t st);
void pm_launch_advance(int32_t * pos, int32_t * ctl, int by, int rotate, hipStream_t st);
void pm_launch_fill_random_blocks(int type, void * dst, int64_t K, int64_t nrows, uint64_t seed, float scale, hipStream_t st);
void pm_launch_fill_random_f32(float * dst, int64_t n, uint64_t seed, float mean, float amp, hipStream_t st);
