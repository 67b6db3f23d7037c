// oracle/ref_ops.c — OUR harness around the UNMODIFIED reference (compiled together with
// /root/reference/ggml/src/{ggml.c,ggml-quants.c,ggml-alloc.c,ggml-aarch64.c,ggml-backend.cpp}
// by oracle/Makefile into oracle/_ref/libggml_ref_<flavour>.so).
//
// TEST INFRASTRUCTURE ONLY. It exposes flat C entry points (ctypes-friendly) that drive the
// reference's own CPU ops through the public ggml graph API, so that
//   (1) oracle/ggml_oracle.c (the restatement) can be pinned against the real thing, and
//   (2) bench.py's cpu_baseline leg can time the reference on the GPU box's host cores.
// Nothing here is linked into the product library.
//
// The node sequences below follow the reference's graph builders:
//   build_llama      src/llama.cpp:11000-11215      build_qwen2   src/llama.cpp:12736-12890
//   llm_build_norm   src/llama.cpp:9772             llm_build_ffn src/llama.cpp:9804
//   llm_build_kv_store src/llama.cpp:9673           llm_build_kqv src/llama.cpp:10032
#include "ggml.h"
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define REF_API __attribute__((visibility("default")))

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static struct ggml_context * make_ctx(size_t bytes) {
    struct ggml_init_params ip = { bytes, NULL, false };
    return ggml_init(ip);
}

// ------------------------------------------------------------------------------------------
// single ops
// ------------------------------------------------------------------------------------------

// dst[N, ncols] = W[K, N] (type) x X[K, ncols] (f32)       ggml_compute_forward_mul_mat, ggml.c:12377
REF_API int ref_mul_mat(int type, const void * w, int64_t K, int64_t N,
                        const float * x, int64_t ncols, float * out, int n_threads) {
    size_t wbytes = ggml_row_size((enum ggml_type) type, K) * N;
    struct ggml_context * ctx = make_ctx(wbytes + (K + N) * ncols * 4 + (64u << 20) +
                                         ggml_row_size(GGML_TYPE_Q8_K, K) * ncols * 2 + K * ncols * 8);
    if (!ctx) return -1;
    struct ggml_tensor * W = ggml_new_tensor_2d(ctx, (enum ggml_type) type, K, N);
    struct ggml_tensor * X = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, K, ncols);
    memcpy(W->data, w, wbytes);
    memcpy(X->data, x, (size_t) K * ncols * 4);
    struct ggml_tensor * Y = ggml_mul_mat(ctx, W, X);
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, Y);
    ggml_graph_compute_with_ctx(ctx, gf, n_threads);
    memcpy(out, Y->data, (size_t) N * ncols * 4);
    ggml_free(ctx);
    return 0;
}

// y = rms_norm(x) [* w]                                  ggml_compute_forward_rms_norm_f32, ggml.c:11950
REF_API int ref_rms_norm(const float * x, const float * w, int64_t n, int64_t rows, float eps, float * out) {
    struct ggml_context * ctx = make_ctx((size_t) n * rows * 16 + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, n, rows);
    memcpy(X->data, x, (size_t) n * rows * 4);
    struct ggml_tensor * Y = ggml_rms_norm(ctx, X, eps);
    if (w) {
        struct ggml_tensor * Wt = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, n);
        memcpy(Wt->data, w, (size_t) n * 4);
        Y = ggml_mul(ctx, Y, Wt);
    }
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, Y);
    ggml_graph_compute_with_ctx(ctx, gf, 1);
    memcpy(out, Y->data, (size_t) n * rows * 4);
    ggml_free(ctx);
    return 0;
}

// x: [d, heads, ntok] f32                                ggml_compute_forward_rope_f32, ggml.c:14143
REF_API int ref_rope(const float * x, int64_t d, int64_t heads, int64_t ntok, const int32_t * pos,
                     const float * freq_factors, int n_dims, int mode, int n_ctx_orig,
                     float freq_base, float freq_scale, float ext_factor, float attn_factor,
                     float beta_fast, float beta_slow, float * out) {
    struct ggml_context * ctx = make_ctx((size_t) d * heads * ntok * 16 + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, d, heads, ntok);
    memcpy(X->data, x, (size_t) d * heads * ntok * 4);
    struct ggml_tensor * P = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, ntok);
    memcpy(P->data, pos, (size_t) ntok * 4);
    struct ggml_tensor * F = NULL;
    if (freq_factors) {
        F = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, n_dims / 2);
        memcpy(F->data, freq_factors, (size_t) (n_dims / 2) * 4);
    }
    struct ggml_tensor * Y = ggml_rope_ext(ctx, X, P, F, n_dims, mode, n_ctx_orig, freq_base, freq_scale,
                                           ext_factor, attn_factor, beta_fast, beta_slow);
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, Y);
    ggml_graph_compute_with_ctx(ctx, gf, 1);
    memcpy(out, Y->data, (size_t) d * heads * ntok * 4);
    ggml_free(ctx);
    return 0;
}

// x: [nc, nr, heads] f32, mask: [nc, nr] f32 or NULL     ggml_compute_forward_soft_max_f32, ggml.c:13783
REF_API int ref_soft_max_ext(const float * x, const float * mask, int64_t nc, int64_t nr, int64_t heads,
                             float scale, float max_bias, float * out) {
    struct ggml_context * ctx = make_ctx((size_t) nc * nr * heads * 16 + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, nc, nr, heads);
    memcpy(X->data, x, (size_t) nc * nr * heads * 4);
    struct ggml_tensor * M = NULL;
    if (mask) {
        M = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, nc, nr);
        memcpy(M->data, mask, (size_t) nc * nr * 4);
    }
    struct ggml_tensor * Y = ggml_soft_max_ext(ctx, X, M, scale, max_bias);
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, Y);
    ggml_graph_compute_with_ctx(ctx, gf, 1);
    memcpy(out, Y->data, (size_t) nc * nr * heads * 4);
    ggml_free(ctx);
    return 0;
}

// out = silu(g) * u                                       ggml_compute_forward_silu_f32, ggml.c:11581
REF_API int ref_silu_mul(const float * g, const float * u, int64_t n, float * out) {
    struct ggml_context * ctx = make_ctx((size_t) n * 32 + (16u << 20));
    struct ggml_tensor * G = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, n);
    memcpy(G->data, g, (size_t) n * 4);
    struct ggml_tensor * Y = ggml_silu(ctx, G);
    if (u) {
        struct ggml_tensor * U = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, n);
        memcpy(U->data, u, (size_t) n * 4);
        Y = ggml_mul(ctx, Y, U);
    }
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, Y);
    ggml_graph_compute_with_ctx(ctx, gf, 1);
    memcpy(out, Y->data, (size_t) n * 4);
    ggml_free(ctx);
    return 0;
}

// ------------------------------------------------------------------------------------------
// whole decoder stack (Llama / Qwen2), stateful, for logits/token goldens
// ------------------------------------------------------------------------------------------

typedef struct { int32_t type; int32_t pad_; const void * data; } ref_tensor_t;

// Same field order as pm355_model_desc (include/prima_mi355.h) and orc_model_desc.
typedef struct {
    int32_t arch;            // 0 = llama (rope NORM, rope_freqs optional), 1 = qwen2 (rope NEOX, qkv bias)
    int32_t n_layer, n_embd, n_head, n_head_kv, head_dim, n_ff, n_vocab, n_ctx, n_ctx_orig;
    float   rms_eps, rope_freq_base, rope_freq_scale;
    int32_t pad_;
    const ref_tensor_t * attn_norm, * wq, * wk, * wv, * wo, * ffn_norm, * ffn_gate, * ffn_up, * ffn_down;
    const ref_tensor_t * bq, * bk, * bv;     // NULL for llama
    ref_tensor_t tok_embd, out_norm, output;
    const float * rope_freqs;                // NULL or [head_dim/2]
} ref_model_desc;

typedef struct {
    ref_model_desc d;
    struct ggml_context * wctx;      // weights + kv
    struct ggml_tensor ** t[12];     // per-layer tensors in desc order
    struct ggml_tensor * tok_embd, * out_norm, * output, * rope_freqs;
    struct ggml_tensor ** k_l, ** v_l;
} ref_model;

static struct ggml_tensor * load_t(struct ggml_context * ctx, const ref_tensor_t * rt, int64_t ne0, int64_t ne1) {
    if (!rt || !rt->data) return NULL;
    struct ggml_tensor * t = ne1 > 0 ? ggml_new_tensor_2d(ctx, (enum ggml_type) rt->type, ne0, ne1)
                                     : ggml_new_tensor_1d(ctx, (enum ggml_type) rt->type, ne0);
    memcpy(t->data, rt->data, ggml_nbytes(t));
    return t;
}

REF_API void * ref_model_new(const ref_model_desc * d) {
    ref_model * m = calloc(1, sizeof(*m));
    m->d = *d;
    const int64_t E = d->n_embd, F = d->n_ff, dh = d->head_dim;
    const int64_t Eq = dh * d->n_head, Ekv = dh * d->n_head_kv;
    size_t bytes = 64u << 20;
    for (int l = 0; l < d->n_layer; ++l) {
        bytes += ggml_row_size(d->wq[l].type, E) * Eq + ggml_row_size(d->wk[l].type, E) * Ekv +
                 ggml_row_size(d->wv[l].type, E) * Ekv + ggml_row_size(d->wo[l].type, Eq) * E +
                 ggml_row_size(d->ffn_gate[l].type, E) * F + ggml_row_size(d->ffn_up[l].type, E) * F +
                 ggml_row_size(d->ffn_down[l].type, F) * E + 4 * (2 * E + Eq + 2 * Ekv) +
                 2 * 2 * Ekv * (size_t) d->n_ctx + 8192;
    }
    bytes += ggml_row_size(d->tok_embd.type, E) * d->n_vocab + ggml_row_size(d->output.type, E) * d->n_vocab + 8 * E;
    m->wctx = make_ctx(bytes);
    if (!m->wctx) { free(m); return NULL; }
    for (int i = 0; i < 12; ++i) m->t[i] = calloc(d->n_layer, sizeof(void *));
    m->k_l = calloc(d->n_layer, sizeof(void *));
    m->v_l = calloc(d->n_layer, sizeof(void *));
    for (int l = 0; l < d->n_layer; ++l) {
        m->t[0][l]  = load_t(m->wctx, &d->attn_norm[l], E, 0);
        m->t[1][l]  = load_t(m->wctx, &d->wq[l], E, Eq);
        m->t[2][l]  = load_t(m->wctx, &d->wk[l], E, Ekv);
        m->t[3][l]  = load_t(m->wctx, &d->wv[l], E, Ekv);
        m->t[4][l]  = load_t(m->wctx, &d->wo[l], Eq, E);
        m->t[5][l]  = load_t(m->wctx, &d->ffn_norm[l], E, 0);
        m->t[6][l]  = load_t(m->wctx, &d->ffn_gate[l], E, F);
        m->t[7][l]  = load_t(m->wctx, &d->ffn_up[l], E, F);
        m->t[8][l]  = load_t(m->wctx, &d->ffn_down[l], F, E);
        m->t[9][l]  = d->bq ? load_t(m->wctx, &d->bq[l], Eq, 0) : NULL;
        m->t[10][l] = d->bk ? load_t(m->wctx, &d->bk[l], Ekv, 0) : NULL;
        m->t[11][l] = d->bv ? load_t(m->wctx, &d->bv[l], Ekv, 0) : NULL;
        // llama_kv_cache_init (src/llama.cpp:3889): F16, one 1-D tensor per layer, zero-cleared
        m->k_l[l] = ggml_new_tensor_1d(m->wctx, GGML_TYPE_F16, Ekv * d->n_ctx);
        m->v_l[l] = ggml_new_tensor_1d(m->wctx, GGML_TYPE_F16, Ekv * d->n_ctx);
        memset(m->k_l[l]->data, 0, ggml_nbytes(m->k_l[l]));
        memset(m->v_l[l]->data, 0, ggml_nbytes(m->v_l[l]));
    }
    m->tok_embd = load_t(m->wctx, &d->tok_embd, E, d->n_vocab);
    m->out_norm = load_t(m->wctx, &d->out_norm, E, 0);
    m->output   = load_t(m->wctx, &d->output, E, d->n_vocab);
    if (d->rope_freqs) {
        m->rope_freqs = ggml_new_tensor_1d(m->wctx, GGML_TYPE_F32, dh / 2);
        memcpy(m->rope_freqs->data, d->rope_freqs, (size_t) (dh / 2) * 4);
    }
    return m;
}

REF_API void ref_model_free(void * vm) {
    ref_model * m = vm;
    if (!m) return;
    ggml_free(m->wctx);
    for (int i = 0; i < 12; ++i) free(m->t[i]);
    free(m->k_l); free(m->v_l); free(m);
}

REF_API void ref_model_kv_clear(void * vm) {
    ref_model * m = vm;
    for (int l = 0; l < m->d.n_layer; ++l) {
        memset(m->k_l[l]->data, 0, ggml_nbytes(m->k_l[l]));
        memset(m->v_l[l]->data, 0, ggml_nbytes(m->v_l[l]));
    }
}

// Evaluate n_tokens tokens at positions pos0..pos0+n_tokens-1 (single sequence, causal).
// If embd_in != NULL it replaces the token-embedding lookup ([n_embd, n_tokens] f32).
// layer_lo/layer_hi select a layer window [lo, hi) (piped-ring sub-graph); with_head adds
// result_norm + lm_head on the LAST token.  hidden_out (optional) = residual stream after layer hi-1,
// [n_embd, n_tokens].  logits_out (optional) = [n_vocab] for the last token.
REF_API int ref_model_eval(void * vm, const int32_t * tokens, const float * embd_in, int n_tokens, int pos0,
                           int layer_lo, int layer_hi, int with_head,
                           float * hidden_out, float * logits_out, int n_threads) {
    ref_model * m = vm;
    const ref_model_desc * d = &m->d;
    const int64_t E = d->n_embd, dh = d->head_dim, H = d->n_head, Hkv = d->n_head_kv;
    const int64_t Ekv = dh * Hkv, n_ctx = d->n_ctx;
    const int64_t n_kv = ((pos0 + n_tokens + 31) / 32) * 32 < n_ctx ? ((pos0 + n_tokens + 31) / 32) * 32 : n_ctx;
    const int rope_mode = d->arch == 1 ? 2 /*GGML_ROPE_TYPE_NEOX*/ : 0;
    const float kq_scale = 1.0f / sqrtf((float) dh);

    size_t cbytes = (size_t) 256u << 20;
    cbytes += (size_t) n_tokens * (E * 64 + d->n_ff * 16 + H * n_kv * 16) + (size_t) d->n_vocab * 8;
    struct ggml_context * ctx = make_ctx(cbytes);
    if (!ctx) return -1;
    struct ggml_cgraph * gf = ggml_new_graph_custom(ctx, 8192, false);

    struct ggml_tensor * inpL;
    if (embd_in) {
        inpL = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, E, n_tokens);
        memcpy(inpL->data, embd_in, (size_t) E * n_tokens * 4);
    } else {
        struct ggml_tensor * tok = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n_tokens);
        memcpy(tok->data, tokens, (size_t) n_tokens * 4);
        inpL = ggml_get_rows(ctx, m->tok_embd, tok);                    // llm_build_inp_embd :9640
    }
    struct ggml_tensor * inp_pos = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n_tokens);
    for (int i = 0; i < n_tokens; ++i) ((int32_t *) inp_pos->data)[i] = pos0 + i;
    // KQ_mask: [n_kv, n_tokens] f32, 0 where key pos <= query pos else -INF (llama_set_inputs :17379)
    struct ggml_tensor * kq_mask = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, n_kv, n_tokens);
    for (int j = 0; j < n_tokens; ++j)
        for (int64_t i = 0; i < n_kv; ++i)
            ((float *) kq_mask->data)[j * n_kv + i] = (i <= pos0 + j) ? 0.0f : -INFINITY;

    struct ggml_tensor * cur;
    for (int il = layer_lo; il < layer_hi; ++il) {
        struct ggml_tensor * inpSA = inpL;
        cur = ggml_mul(ctx, ggml_rms_norm(ctx, inpL, d->rms_eps), m->t[0][il]);
        struct ggml_tensor * Qcur = ggml_mul_mat(ctx, m->t[1][il], cur);
        if (m->t[9][il])  Qcur = ggml_add(ctx, Qcur, m->t[9][il]);
        struct ggml_tensor * Kcur = ggml_mul_mat(ctx, m->t[2][il], cur);
        if (m->t[10][il]) Kcur = ggml_add(ctx, Kcur, m->t[10][il]);
        struct ggml_tensor * Vcur = ggml_mul_mat(ctx, m->t[3][il], cur);
        if (m->t[11][il]) Vcur = ggml_add(ctx, Vcur, m->t[11][il]);
        Qcur = ggml_rope_ext(ctx, ggml_reshape_3d(ctx, Qcur, dh, H, n_tokens), inp_pos, m->rope_freqs,
                             (int) dh, rope_mode, d->n_ctx_orig, d->rope_freq_base, d->rope_freq_scale,
                             0.0f, 1.0f, 32.0f, 1.0f);
        Kcur = ggml_rope_ext(ctx, ggml_reshape_3d(ctx, Kcur, dh, Hkv, n_tokens), inp_pos, m->rope_freqs,
                             (int) dh, rope_mode, d->n_ctx_orig, d->rope_freq_base, d->rope_freq_scale,
                             0.0f, 1.0f, 32.0f, 1.0f);
        // llm_build_kv_store
        ggml_build_forward_expand(gf, Qcur);
        ggml_build_forward_expand(gf, Kcur);
        ggml_build_forward_expand(gf, Vcur);
        struct ggml_tensor * k_view = ggml_view_1d(ctx, m->k_l[il], n_tokens * Ekv,
                                                   ggml_row_size(GGML_TYPE_F16, Ekv) * pos0);
        ggml_build_forward_expand(gf, ggml_cpy(ctx, Kcur, k_view));
        struct ggml_tensor * v_view = ggml_view_2d(ctx, m->v_l[il], n_tokens, Ekv,
                                                   n_ctx * ggml_element_size(m->v_l[il]),
                                                   pos0 * ggml_element_size(m->v_l[il]));
        ggml_build_forward_expand(gf, ggml_cpy(ctx, ggml_transpose(ctx, Vcur), v_view));
        // llm_build_kqv
        struct ggml_tensor * q = ggml_permute(ctx, Qcur, 0, 2, 1, 3);
        struct ggml_tensor * k = ggml_view_3d(ctx, m->k_l[il], dh, n_kv, Hkv,
                                              ggml_row_size(GGML_TYPE_F16, Ekv), ggml_row_size(GGML_TYPE_F16, dh), 0);
        struct ggml_tensor * kq = ggml_mul_mat(ctx, k, q);
        kq = ggml_soft_max_ext(ctx, kq, kq_mask, kq_scale, 0.0f);
        struct ggml_tensor * v = ggml_view_3d(ctx, m->v_l[il], n_kv, dh, Hkv,
                                              ggml_element_size(m->v_l[il]) * n_ctx,
                                              ggml_element_size(m->v_l[il]) * n_ctx * dh, 0);
        struct ggml_tensor * kqv = ggml_mul_mat(ctx, v, kq);
        cur = ggml_cont_2d(ctx, ggml_permute(ctx, kqv, 0, 2, 1, 3), dh * H, n_tokens);
        ggml_build_forward_expand(gf, cur);
        cur = ggml_mul_mat(ctx, m->t[4][il], cur);
        struct ggml_tensor * ffn_inp = ggml_add(ctx, cur, inpSA);
        cur = ggml_mul(ctx, ggml_rms_norm(ctx, ffn_inp, d->rms_eps), m->t[5][il]);
        // llm_build_ffn (LLM_FFN_SILU, LLM_FFN_PAR): up, gate, silu(gate), gate*up, down
        struct ggml_tensor * up   = ggml_mul_mat(ctx, m->t[7][il], cur);
        struct ggml_tensor * gate = ggml_mul_mat(ctx, m->t[6][il], cur);
        gate = ggml_silu(ctx, gate);
        cur = ggml_mul(ctx, gate, up);
        cur = ggml_mul_mat(ctx, m->t[8][il], cur);
        cur = ggml_add(ctx, cur, ffn_inp);
        inpL = cur;
    }
    ggml_build_forward_expand(gf, inpL);
    struct ggml_tensor * logits = NULL;
    if (with_head) {
        struct ggml_tensor * last = ggml_view_2d(ctx, inpL, E, 1, inpL->nb[1], (size_t) (n_tokens - 1) * inpL->nb[1]);
        cur = ggml_mul(ctx, ggml_rms_norm(ctx, last, d->rms_eps), m->out_norm);
        logits = ggml_mul_mat(ctx, m->output, cur);
        ggml_build_forward_expand(gf, logits);
    }
    enum ggml_status st = ggml_graph_compute_with_ctx(ctx, gf, n_threads);
    if (hidden_out) memcpy(hidden_out, inpL->data, (size_t) E * n_tokens * 4);
    if (logits_out && logits) memcpy(logits_out, logits->data, (size_t) d->n_vocab * 4);
    ggml_free(ctx);
    return st == GGML_STATUS_SUCCESS ? 0 : -2;
}

// read back a KV cache row for tests: layer il, which = 0 (K) / 1 (V), raw f16 bytes
REF_API const void * ref_model_kv_ptr(void * vm, int il, int which) {
    ref_model * m = vm;
    return which ? m->v_l[il]->data : m->k_l[il]->data;
}

// ------------------------------------------------------------------------------------------
// CPU-baseline timing: one decode step's worth of quantized mat-vecs for ONE layer
// (q,k,v,o,gate,up,down) on the reference CPU backend with n_threads threads.
// Weight bytes are whatever the caller passes (random valid blocks are fine for timing).
// Returns seconds per pass (best of `reps`).
// ------------------------------------------------------------------------------------------
REF_API double ref_time_layer_matvecs(int n_mats, const int32_t * types, const int64_t * Ks, const int64_t * Ns,
                                      const void * const * datas, int n_threads, int reps) {
    size_t bytes = 64u << 20;
    int64_t maxK = 0;
    for (int i = 0; i < n_mats; ++i) {
        bytes += ggml_row_size(types[i], Ks[i]) * Ns[i] + (Ks[i] + Ns[i]) * 8 + 4096;
        if (Ks[i] > maxK) maxK = Ks[i];
    }
    bytes += maxK * 64;
    struct ggml_context * ctx = make_ctx(bytes);
    if (!ctx) return -1.0;
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    for (int i = 0; i < n_mats; ++i) {
        struct ggml_tensor * W = ggml_new_tensor_2d(ctx, (enum ggml_type) types[i], Ks[i], Ns[i]);
        memcpy(W->data, datas[i], ggml_nbytes(W));
        struct ggml_tensor * X = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, Ks[i]);
        for (int64_t j = 0; j < Ks[i]; ++j) ((float *) X->data)[j] = 0.01f * (float) ((j * 37) % 201 - 100);
        ggml_build_forward_expand(gf, ggml_mul_mat(ctx, W, X));
    }
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        double t0 = now_s();
        ggml_graph_compute_with_ctx(ctx, gf, n_threads);
        double t1 = now_s();
        if (t1 - t0 < best) best = t1 - t0;
    }
    ggml_free(ctx);
    return best;
}

REF_API const char * ref_build_info(void) {
#if defined(__AVX512F__)
    return "reference ggml CPU backend, gcc, -O2, AVX512 path, no OpenMP, no llamafile";
#elif defined(__AVX2__)
    return "reference ggml CPU backend, gcc, -O2, AVX2 path, no OpenMP, no llamafile";
#else
    return "reference ggml CPU backend, gcc, -O2, scalar path, no OpenMP, no llamafile";
#endif
}
