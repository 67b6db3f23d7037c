// ggml_graph_plan.h — lowering of a ggml_cgraph split (what ggml_backend_sched hands to graph_compute,
// ggml/src/ggml-backend.cpp:2139) into the launch sequence of libprima_mi355.so. Host C++ only; included by
// ggml_backend_mi355.cpp, compiled against the HOST PROJECT's ggml headers.
//
// The reference's graph builders (build_llama src/llama.cpp:11000-11215, build_qwen2 :12736-12900; node names are set by
// the cb callback but NOTHING here depends on names) emit per layer, for ONE token (ne[1] == 1):
//
//   RMS_NORM MUL(attn_norm.w) | MUL_MAT(wq)[ADD bq] RESHAPE ROPE | MUL_MAT(wk)[ADD bk] RESHAPE ROPE | MUL_MAT(wv)[ADD bv]
//   VIEW(k cell) CPY(K) | TRANSPOSE(V) VIEW(v cell) CPY(V) | VIEW(v) VIEW(k) PERMUTE(q) | MUL_MAT(k,q) SOFT_MAX MUL_MAT(v,kq)
//   PERMUTE CONT                                                                   -> launch 1 (QKV) + launch 2 (attention)
//   MUL_MAT(wo) ADD(residual)                                                      -> launch 3
//   RMS_NORM MUL(ffn_norm.w) MUL_MAT(gate) UNARY(silu) MUL_MAT(up) MUL(silu,up)    -> launch 4
//   MUL_MAT(down) ADD(residual)                                                    -> launch 5
// and for the head RMS_NORM MUL(output_norm.w) MUL_MAT(output) -> one launch. Matching is purely structural (ops, data flow,
// shapes, strides, types); a chain is fused only when every intermediate tensor it no longer materialises has no consumer
// outside the chain and carries no input/output flag (cf. upstream ggml_can_fuse), and when no output of a fused launch
// overlaps one of its inputs. Everything that does not match falls back to the node-equivalent kernels, node by node, so
// eval callbacks (single-node graphs), test-backend-ops, partial offload and multi-token batches keep their exact semantics.
//
// Per-token variables never enter the plan: the RoPE position is read from the graph's inp_pos tensor on the device, the KV
// cell of the token and the number of cells attended (kv_self.head / kv_self.n, src/llama.cpp:18433-18453) are written to two
// device ints before the launch sequence. A plan therefore stays valid from token to token and its launch sequence is
// captured once into a hipGraph and replayed (the reference CUDA plug-in does the same with CUDA graphs and per-token
// kernel-parameter updates, ggml/src/ggml-cuda.cu:2513-2780).
#pragma once
#include "ggml.h"
#include "ggml-backend.h"
#include "../../include/prima_mi355.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace mi355 {

// STEP_ROPE_TAB / STEP_QKV / STEP_ATTN_CACHED: the round-3 form of launches 1 and 2 - one cos / sin table launch per graph, RoPE + KV store in the
// epilogue of the wq | wk | wv launch, attention over cached cells (pm355_rope_table / pm355_mul_mat_vec_qkv / pm355_attn_cached)
enum step_kind : int32_t { STEP_NODE = 0, STEP_GEMV = 1, STEP_ATTN = 2, STEP_ATTN_BATCH = 3, STEP_ROPE_TAB = 4, STEP_QKV = 5, STEP_ATTN_CACHED = 6 };

struct tensor_fp {                                   // everything a node-equivalent kernel reads from a tensor
    const void * data; int32_t type; int32_t pad_; int64_t ne[4]; size_t nb[4];
};
struct node_fp {
    int32_t op; int32_t flags; int32_t op_params[16];
    tensor_fp dst; tensor_fp src[4];
};
struct step {
    int32_t kind; int32_t node;                      // STEP_NODE: graph index of the node to run with compute_node()
    int32_t node_lo, node_hi;                        // graph nodes [lo, hi) this step stands for
    // STEP_GEMV
    int64_t K; int32_t njobs; float eps; const float * x; const float * norm_w; pm355_matvec_job job[3];
    // producer-side sum of squares (pm355_mul_mat_vec_fused_ss): ss_out - this single-job launch leaves its per-workgroup partials there;
    // ss_in / n_ss - the rms_norm of this launch (STEP_GEMV or STEP_QKV with norm_w) adds the n_ss partials the step right before it left
    double * ss_out; const double * ss_in; int32_t n_ss; int32_t pad_ss_;
    // STEP_ATTN (STEP_ATTN_CACHED: the same arguments, q = rotated rows; STEP_ROPE_TAB: rope, attn.d_pos, attn.freq_factors, qs.rope_table)
    pm355_attn_token_args attn; pm355_rope_params rope;
    // STEP_QKV (with the STEP_GEMV fields)
    pm355_qkv_store qs;
    // STEP_ATTN_BATCH (multi-token): pm355_attn_prefill_masked
    struct { const float * q; const void * kc, * vc; const void * mask; int64_t mask_stride; float * out;
             int32_t n_tokens, n_head, n_head_kv, head_dim, n_ctx, n_kv; float scale; int32_t flags; } ab;
    // STEP_NODE
    node_fp fp;
};

struct plan {
    std::vector<step> steps;
    bool single_token = true;                        // every node with a token dimension has ne[1] == 1
    bool has_attn = false;
    bool fast_ok = true;                             // no view with a non-zero offset other than the KV cell views handled via `dyn`
    int  i_kcell = -1, i_kview = -1;                 // graph indices: CPY into the K cell view (-> cell), VIEW k (ne[1] = cells attended)
    int  n_fused_nodes = 0;
    int  n_gemv = 0, n_attn = 0, n_node = 0, n_attn_batch = 0;
};

struct plan_ctx {                                    // what the backend provides to the planner
    void * user;
    // device scratch for the raw q / k / v projections of one token (n_q + 2 * n_kv floats, stable while large enough); null = no memory
    float * (*qkv_scratch)(void * user, size_t n_q, size_t n_kv);
    // device scratch of the keys-split-over-workgroups attention; null = no memory
    float * (*split_scratch)(void * user, size_t n_floats);
    int32_t * d_dyn;                                 // device int32[2] {cell, cells attended}
    int split_min;                                   // cells attended from which the split attention is used
    bool attn_mfma;                                  // long contexts in the round-3 form: matrix-core attention over cached cells
    bool fuse;
    // round-3 form of the attention block: device table of 256 floats (null: off) and a comparison of two small device arrays
    // (the per-layer copies of rope_freqs hold the same numbers: one table serves every layer)
    float * rope_tab = nullptr;
    bool (*same_bytes)(void * user, const void * a, const void * b, size_t n) = nullptr;
    double * ss_buf = nullptr;                       // device double[2][256] for the producer-side sum-of-squares partials (null: off)
};

// ---- small pointer -> count map (open addressing; graphs have a few thousand nodes) ---------------------------------------
struct ptr_count {
    std::vector<const void *> k; std::vector<int> v; size_t mask = 0;
    void init(size_t n) { size_t c = 64; while (c < 4 * n) c <<= 1; k.assign(c, nullptr); v.assign(c, 0); mask = c - 1; }
    static size_t h(const void * p) { uint64_t x = (uint64_t) (uintptr_t) p; x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 29; return (size_t) x; }
    int & at(const void * p) { size_t i = h(p) & mask; while (k[i] && k[i] != p) i = (i + 1) & mask; k[i] = p; return v[i]; }
    int get(const void * p) const { size_t i = h(p) & mask; while (k[i] && k[i] != p) i = (i + 1) & mask; return k[i] ? v[i] : 0; }
};

inline bool is_view_op(enum ggml_op op) {
    return op == GGML_OP_NONE || op == GGML_OP_RESHAPE || op == GGML_OP_VIEW || op == GGML_OP_PERMUTE || op == GGML_OP_TRANSPOSE;
}
inline bool quant_matvec_type(enum ggml_type t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K || t == GGML_TYPE_Q8_0; }
inline bool soa_type(enum ggml_type t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q6_K || t == GGML_TYPE_Q8_0; }
inline bool f32_vec(const ggml_tensor * t, int64_t n) {           // contiguous f32 [n, 1, 1, 1]
    return t && t->type == GGML_TYPE_F32 && t->ne[0] == n && t->ne[1] == 1 && t->ne[2] == 1 && t->ne[3] == 1 && t->nb[0] == 4;
}
inline bool overlap(const void * a, size_t na, const void * b, size_t nb) {
    const char * x = (const char *) a, * y = (const char *) b;
    return x < y + nb && y < x + na;
}

inline void fill_tensor_fp(tensor_fp & f, const ggml_tensor * t) {
    f.data = t->data; f.type = (int32_t) t->type; f.pad_ = 0;
    for (int i = 0; i < 4; ++i) { f.ne[i] = t->ne[i]; f.nb[i] = t->nb[i]; }
}

// ---- per-token fingerprint of a graph: equal fingerprints => equal plans (view offsets into F16 leaf tensors = the KV cell views
//      are left out on purpose: they are the per-token `dyn` values) ----------------------------------------------------------
// The fingerprint is a 128-bit hash + the node count (round 3; it used to be one 112-byte record per node, filled and memcmp'ed on every
// graph_compute: 40-50 us of host time for the 1056-node split of an 8B model - a 16-byte compare now, ~10 us to hash).
struct graph_fp { uint64_t h0 = 0, h1 = 0; int32_t n = -1; };

inline const void * fp_base(const ggml_tensor * t) {
    const ggml_tensor * r = t->view_src;           // F16 leaf, or a 1-D Q8_0 leaf (quantized KV cache)
    if (r && (r->type == GGML_TYPE_F16 || (r->type == GGML_TYPE_Q8_0 && r->ne[1] == 1)) && r->op == GGML_OP_NONE && !r->view_src) return r->data;
    return t->data;
}
inline void graph_fingerprint(struct ggml_cgraph * g, graph_fp & out) {
    const int n = ggml_graph_n_nodes(g);
    uint64_t a = 0x9E3779B97F4A7C15ULL, b = 0xC2B2AE3D27D4EB4FULL;
    auto mix = [&](uint64_t v) { a = (a ^ v) * 0xff51afd7ed558ccdULL; a ^= a >> 32; b = (b + v) * 0xc4ceb9fe1a85ec53ULL; b ^= b >> 29; };
    for (int i = 0; i < n; ++i) {
        const ggml_tensor * t = ggml_graph_node(g, i);
        const void * base = fp_base(t);
        mix(((uint64_t) (uint32_t) t->op << 40) ^ ((uint64_t) (uint32_t) t->type << 20) ^ (uint64_t) (uint32_t) t->flags);
        mix((uint64_t) (uintptr_t) base);
        // (a VIEW keeps its byte offset in op_params, ggml_view_impl ggml.c:6490: for the KV cell views that offset is the per-token value the
        // fingerprint leaves out - hashing it made every token miss and re-plan the whole split)
        const bool cell_view = t->op == GGML_OP_VIEW && base != t->data;
        if (!cell_view) for (int k = 0; k < 16; k += 2) mix(((uint64_t) (uint32_t) t->op_params[k] << 32) | (uint32_t) t->op_params[k + 1]);
        for (int k = 0; k < 4; ++k) { mix((uint64_t) t->ne[k]); mix((uint64_t) t->nb[k]); }
        for (int k = 0; k < 3; ++k) mix(t->src[k] ? (uint64_t) (uintptr_t) fp_base(t->src[k]) : 0);
    }
    out.h0 = a; out.h1 = b; out.n = n;
}
inline bool fingerprint_equal(const graph_fp & a, const graph_fp & b) { return a.n == b.n && a.h0 == b.h0 && a.h1 == b.h1; }

// ---- the planner -------------------------------------------------------------------------------------------------------------
class planner {
public:
    planner(struct ggml_cgraph * g, const plan_ctx & c) : g_(g), c_(c), n_(ggml_graph_n_nodes(g)) {
        nodes_.resize(n_);
        for (int i = 0; i < n_; ++i) nodes_[i] = ggml_graph_node(g, i);
    }

    void build(plan & p) {
        p = plan();
        uses_.init((size_t) n_ * 3);
        for (int i = 0; i < n_; ++i) count_refs(nodes_[i], uses_, +1);
        for (int i = 0; i < n_; ++i) if (nodes_[i]->op == GGML_OP_MUL_MAT && nodes_[i]->ne[1] != 1) p.single_token = false;
        int i = 0;
        while (i < n_) {
            int adv = 0;
            if (c_.fuse && p.single_token) {
                adv = try_attention_block(p, i);
                if (!adv) adv = try_norm_matvec(p, i);
                if (!adv) adv = try_matvec_resid(p, i);
            }
            if (!adv && c_.fuse && !p.single_token) adv = try_batch_attention(p, i);
            if (!adv && c_.fuse && !p.single_token) adv = try_batch_flash_attention(p, i);
            if (adv) { p.n_fused_nodes += adv; i += adv; continue; }
            emit_node(p, i);
            ++i;
        }
        wire_sumsq(p);
    }

    // Producer-side sum of squares: a fused mat-vec with rms_norm (wq | wk | wv, ffn_gate | ffn_up, lm_head) whose input row is exactly the output
    // of the single-job mat-vec launched right before it (wo / ffn_down + residual) takes the sum of squares from that launch's per-workgroup
    // partials. Adjacent steps only: nothing can have written the row in between.
    void wire_sumsq(plan & p) {
        if (!c_.ss_buf) return;
        int flip = 0;
        for (size_t i = 1; i < p.steps.size(); ++i) {
            step & cs = p.steps[i]; step & pr = p.steps[i - 1];
            if ((cs.kind != STEP_GEMV && cs.kind != STEP_QKV) || !cs.norm_w) continue;
            if (pr.kind != STEP_GEMV || pr.njobs != 1 || pr.job[0].W2 || pr.job[0].y != cs.x || pr.job[0].N != cs.K) continue;
            const int g = pm355_mul_mat_vec_fused_grid(pr.job, 1, pr.K);
            if (g < 1 || g > 256) continue;
            pr.ss_out = c_.ss_buf + 256 * flip; cs.ss_in = pr.ss_out; cs.n_ss = g;
            flip ^= 1;
        }
    }

private:
    struct ggml_cgraph * g_; const plan_ctx & c_; int n_;
    std::vector<ggml_tensor *> nodes_;
    ptr_count uses_;

    static void count_refs(const ggml_tensor * t, ptr_count & m, int d) {
        for (int k = 0; k < GGML_MAX_SRC; ++k) if (t->src[k]) m.at(t->src[k]) += d;
        if (t->view_src) m.at(t->view_src) += d;
    }
    ggml_tensor * N(int i) const { return i >= 0 && i < n_ ? nodes_[i] : nullptr; }

    // every node of [lo, hi) that is not in `outs` must be invisible outside the range
    bool range_private(int lo, int hi, const int * outs, int n_outs) const {
        ptr_count inner; inner.init((size_t) (hi - lo) * 3);
        for (int i = lo; i < hi; ++i) count_refs(nodes_[i], inner, +1);
        for (int i = lo; i < hi; ++i) {
            bool is_out = false;
            for (int k = 0; k < n_outs; ++k) if (outs[k] == i) is_out = true;
            if (is_out) continue;
            const ggml_tensor * t = nodes_[i];
            if (t->flags & (GGML_TENSOR_FLAG_INPUT | GGML_TENSOR_FLAG_OUTPUT)) return false;
            if (uses_.get(t) != inner.get(t)) return false;
        }
        return true;
    }

    void emit_node(plan & p, int i) {
        ggml_tensor * t = nodes_[i];
        // a view into an F16 leaf (KV cache) at a non-zero offset that no fused path absorbed: its pointer moves from token to token
        // while the graph fingerprint leaves that offset out -> such graphs are never replayed on a fingerprint match
        if (fp_base(t) != t->data) p.fast_ok = false;
        for (int k = 0; k < GGML_MAX_SRC && t->src[k]; ++k) if (fp_base(t->src[k]) != t->src[k]->data) p.fast_ok = false;
        if (ggml_is_empty(t) || is_view_op(t->op)) return;
        step s; memset(&s, 0, sizeof(s));
        s.kind = STEP_NODE; s.node = i; s.node_lo = i; s.node_hi = i + 1;
        s.fp.op = (int32_t) t->op; s.fp.flags = t->flags;
        memcpy(s.fp.op_params, t->op_params, sizeof(s.fp.op_params));
        fill_tensor_fp(s.fp.dst, t);
        for (int k = 0; k < 4; ++k) if (t->src[k]) fill_tensor_fp(s.fp.src[k], t->src[k]);
        p.steps.push_back(s);
        ++p.n_node;
    }

    // MUL_MAT of a quantized, contiguous, non-view weight matrix with ONE f32 activation row
    bool is_matvec(const ggml_tensor * t, const ggml_tensor * x_expected = nullptr) const {
        if (!t || t->op != GGML_OP_MUL_MAT) return false;
        const ggml_tensor * w = t->src[0], * x = t->src[1];
        if (!w || !x || !quant_matvec_type(w->type) || w->view_src || !ggml_is_contiguous(w) || w->ne[2] != 1 || w->ne[3] != 1) return false;
        if (w->ne[0] % (w->type == GGML_TYPE_Q8_0 ? 32 : 256) || w->ne[0] > 131072) return false;
        if (!f32_vec(x, w->ne[0]) || !f32_vec(t, w->ne[1])) return false;
        return !x_expected || x == x_expected;
    }
    static pm355_matvec_job job_of(const ggml_tensor * mm, float * y, const float * bias, const float * resid, const ggml_tensor * w2 = nullptr) {
        pm355_matvec_job j; memset(&j, 0, sizeof(j));
        j.type = (int32_t) mm->src[0]->type; j.N = mm->src[0]->ne[1]; j.W = mm->src[0]->data; j.W2 = w2 ? w2->data : nullptr;
        j.y = y; j.bias = bias; j.resid = resid;
        return j;
    }
    void push_gemv(plan & p, int lo, int hi, const pm355_matvec_job * jobs, int nj, int64_t K, const float * x, const float * norm_w, float eps) {
        step s; memset(&s, 0, sizeof(s));
        s.kind = STEP_GEMV; s.node = -1; s.node_lo = lo; s.node_hi = hi;
        s.K = K; s.njobs = nj; s.eps = eps; s.x = x; s.norm_w = norm_w;
        for (int j = 0; j < nj; ++j) s.job[j] = jobs[j];
        p.steps.push_back(s);
        ++p.n_gemv;
    }
    // outputs of a mat-vec launch may not overlap its inputs (all workgroups read x while others already write y); y == resid is fine
    static bool gemv_alias_ok(const pm355_matvec_job * jobs, int nj, int64_t K, const float * x) {
        for (int j = 0; j < nj; ++j) {
            const size_t yb = (size_t) jobs[j].N * 4;
            if (overlap(jobs[j].y, yb, x, (size_t) K * 4)) return false;
            if (jobs[j].resid && jobs[j].resid != jobs[j].y && overlap(jobs[j].y, yb, jobs[j].resid, yb)) return false;
            for (int k = 0; k < nj; ++k) if (k != j) {
                if (overlap(jobs[j].y, yb, jobs[k].y, (size_t) jobs[k].N * 4)) return false;
                if (jobs[k].resid && overlap(jobs[j].y, yb, jobs[k].resid, (size_t) jobs[k].N * 4)) return false;
            }
        }
        return true;
    }

    // RMS_NORM x ; MUL(norm, w)  -> returns the MUL node (the normalised activation) or null
    const ggml_tensor * match_norm(int i, const ggml_tensor *& x, const float *& w, float & eps) const {
        const ggml_tensor * rn = N(i), * mu = N(i + 1);
        if (!rn || !mu || rn->op != GGML_OP_RMS_NORM || mu->op != GGML_OP_MUL) return nullptr;
        x = rn->src[0];
        if (!x || x->ne[0] % 256 || !f32_vec(x, x->ne[0]) || !f32_vec(rn, x->ne[0]) || !f32_vec(mu, x->ne[0])) return nullptr;
        const ggml_tensor * wt = mu->src[0] == rn ? mu->src[1] : (mu->src[1] == rn ? mu->src[0] : nullptr);
        if (!wt || !f32_vec(wt, x->ne[0])) return nullptr;
        w = (const float *) wt->data;
        memcpy(&eps, rn->op_params, sizeof(float));
        return mu;
    }

    // MUL_MAT(W, xin) [ADD(mm, bias[N])] -> advances i; bias may be absent
    bool match_proj(int & i, const ggml_tensor * xin, const ggml_tensor *& mm, const float *& bias, const ggml_tensor *& out) const {
        mm = N(i);
        if (!is_matvec(mm, xin)) return false;
        ++i; bias = nullptr; out = mm;
        const ggml_tensor * ad = N(i);
        if (ad && ad->op == GGML_OP_ADD && (ad->src[0] == mm) && f32_vec(ad->src[1], mm->ne[0]) && f32_vec(ad, mm->ne[0]) &&
            ad->src[1]->op == GGML_OP_NONE) {                       // a leaf = a weight (bias), not an activation
            bias = (const float *) ad->src[1]->data; out = ad; ++i;
        }
        return true;
    }

    // RESHAPE(x -> [dh, heads, 1]) ROPE(reshaped, pos, [ff])
    bool match_rope(int & i, const ggml_tensor * x, int64_t dh, const ggml_tensor *& rope) const {
        const ggml_tensor * rs = N(i), * rp = N(i + 1);
        if (!rs || !rp || rs->op != GGML_OP_RESHAPE || rs->src[0] != x || rp->op != GGML_OP_ROPE || rp->src[0] != rs) return false;
        if (rs->ne[0] != dh || rs->ne[2] != 1 || rs->ne[3] != 1 || rs->nb[0] != 4 || rs->nb[1] != (size_t) dh * 4 || rs->data != x->data) return false;
        if (rp->type != GGML_TYPE_F32 || rp->ne[0] != dh || rp->ne[1] != rs->ne[1] || rp->ne[2] != 1 || rp->nb[0] != 4 || rp->nb[1] != (size_t) dh * 4) return false;
        const ggml_tensor * pos = rp->src[1];
        if (!pos || pos->type != GGML_TYPE_I32 || pos->ne[0] != 1) return false;
        if (rp->src[2] && !f32_vec(rp->src[2], dh / 2) && !(rp->src[2]->type == GGML_TYPE_F32 && rp->src[2]->ne[0] >= dh / 2 && rp->src[2]->nb[0] == 4)) return false;
        const int mode = rp->op_params[2];
        if (mode != 0 && mode != 2) return false;
        rope = rp; i += 2;
        return true;
    }
    static void rope_params_of(const ggml_tensor * rp, pm355_rope_params & r) {
        const int32_t * prm = rp->op_params;
        memset(&r, 0, sizeof(r));
        r.n_dims = prm[1]; r.mode = prm[2]; r.n_ctx_orig = prm[4];
        memcpy(&r.freq_base, prm + 5, 4); memcpy(&r.freq_scale, prm + 6, 4); memcpy(&r.ext_factor, prm + 7, 4);
        memcpy(&r.attn_factor, prm + 8, 4); memcpy(&r.beta_fast, prm + 9, 4); memcpy(&r.beta_slow, prm + 10, 4);
    }

    // ---- launch 1 + 2 of a layer: everything from the attention norm to the merged attention output ------------------------
    int try_attention_block(plan & p, int i0) {
        const ggml_tensor * x; const float * nw; float eps;
        const ggml_tensor * xn = match_norm(i0, x, nw, eps);
        if (!xn) return 0;
        int i = i0 + 2;
        const ggml_tensor * mq, * mk, * mv, * q, * k, * v, * rq, * rk; const float * bq, * bk, * bv;
        if (!match_proj(i, xn, mq, bq, q)) return 0;
        const int64_t E = x->ne[0], Eq = mq->ne[0];
        // head_dim comes from the RESHAPE that follows
        const ggml_tensor * rs = N(i);
        if (!rs || rs->op != GGML_OP_RESHAPE) return 0;
        const int64_t dh = rs->ne[0];
        if (dh <= 0 || Eq % dh) return 0;
        if (!match_rope(i, q, dh, rq)) return 0;
        if (!match_proj(i, xn, mk, bk, k) || !match_rope(i, k, dh, rk)) return 0;
        if (!match_proj(i, xn, mv, bv, v)) return 0;
        const int64_t Ekv = mk->ne[0];
        if (mv->ne[0] != Ekv || Ekv % dh || Eq % Ekv) return 0;
        const int64_t H = Eq / dh, Hkv = Ekv / dh;
        if (rq->ne[1] != H || rk->ne[1] != Hkv) return 0;
        // both rotations: same parameters, same position tensor, same frequency factors
        if (memcmp(rq->op_params, rk->op_params, sizeof(rq->op_params)) || rq->src[1] != rk->src[1] || rq->src[2] != rk->src[2]) return 0;
        if (rq->op_params[1] > dh || rq->op_params[1] % 2) return 0;
        // KV store: VIEW(k cell) CPY(Krot -> cell) ; then either TRANSPOSE(V) VIEW(v cell) CPY(V^T -> cell) (llm_build_kv_store,
        // src/llama.cpp:9712-9722) or, in flash-attention graphs, VIEW(v row) CPY(V -> row) (:9705-9707)
        const ggml_tensor * kcv = N(i), * kcp = N(i + 1);
        if (!kcv || !kcp || !N(i + 2) || !N(i + 3)) return 0;
        const bool fa = N(i + 2)->op != GGML_OP_TRANSPOSE;
        const ggml_tensor * vt = fa ? nullptr : N(i + 2), * vcv = fa ? N(i + 2) : N(i + 3), * vcp = fa ? N(i + 3) : N(i + 4);
        if (!vcv || !vcp) return 0;
        if (kcv->op != GGML_OP_VIEW || kcp->op != GGML_OP_CPY || vcv->op != GGML_OP_VIEW || vcp->op != GGML_OP_CPY) return 0;
        const ggml_tensor * kcache = kcv->view_src, * vcache = vcv->view_src;
        if (!kcache || !vcache || kcache->view_src || vcache->view_src ||
            kcache->op != GGML_OP_NONE || vcache->op != GGML_OP_NONE || !ggml_is_contiguous(kcache) || !ggml_is_contiguous(vcache)) return 0;
        // F16 caches; with flash attention also Q8_0 (1-D tensors of native 34-byte blocks, attn_q8.hip)
        const bool kq8 = kcache->type == GGML_TYPE_Q8_0, vq8 = vcache->type == GGML_TYPE_Q8_0;
        if ((!kq8 && kcache->type != GGML_TYPE_F16) || (!vq8 && vcache->type != GGML_TYPE_F16) || ((kq8 || vq8) && !fa)) return 0;
        if ((kq8 && (kcache->ne[1] != 1 || dh % 32)) || (vq8 && (vcache->ne[1] != 1 || dh % 32))) return 0;
        const size_t k_ts = kq8 ? 34 : 2, v_ts = vq8 ? 34 : 2;
        const size_t k_row = kq8 ? (size_t) Ekv / 32 * 34 : (size_t) Ekv * 2, v_rowb = vq8 ? (size_t) Ekv / 32 * 34 : (size_t) Ekv * 2;
        const size_t k_hd = kq8 ? (size_t) dh / 32 * 34 : (size_t) dh * 2, v_hd = vq8 ? (size_t) dh / 32 * 34 : (size_t) dh * 2;
        if (kcp->src[0] != rk || kcp->src[1] != kcv || kcv->type != kcache->type || kcv->ne[0] != Ekv || kcv->ne[1] != 1 || kcv->nb[0] != k_ts) return 0;
        const size_t k_off = (const char *) kcv->data - (const char *) kcache->data, v_off = (const char *) vcv->data - (const char *) vcache->data;
        int64_t n_ctx;
        if (!fa) {
            if (vt->src[0] != v || vcp->src[0] != vt || vcp->src[1] != vcv || vcv->type != GGML_TYPE_F16 || vcv->ne[0] != 1 || vcv->ne[1] != Ekv || vcv->nb[0] != 2) return 0;
            if (k_off % k_row || v_off % 2 || k_off / k_row != v_off / 2) return 0;
            n_ctx = (int64_t) (vcv->nb[1] / 2);
        } else {
            if (vcp->src[0] != v || vcp->src[1] != vcv || vcv->type != vcache->type || vcv->ne[0] != Ekv || vcv->ne[1] != 1 || vcv->nb[0] != v_ts) return 0;
            if (k_off % k_row || v_off % v_rowb || v_off / v_rowb != k_off / k_row) return 0;
            n_ctx = (int64_t) ggml_nelements(kcache) / Ekv;
        }
        if (n_ctx <= 0 || n_ctx % 8 || (int64_t) ggml_nelements(kcache) < n_ctx * Ekv || (int64_t) ggml_nelements(vcache) < n_ctx * Ekv) return 0;
        const int64_t cell = (int64_t) (k_off / k_row);
        if (cell >= n_ctx) return 0;
        i += fa ? 4 : 5;
        // VIEW(v) VIEW(k) PERMUTE(q) in any order; flash-attention graphs may also hold the F32 -> F16 cast of the KQ mask here
        // (build_inp_KQ_mask, src/llama.cpp:10466: once per graph, in front of its first use)
        const ggml_tensor * vv = nullptr, * kv = nullptr, * qp = nullptr;
        int mask_cast = -1;
        for (int r = 0; r < 4; ++r) {
            const ggml_tensor * t = N(i);
            if (!t) return 0;
            if (t->op == GGML_OP_VIEW && t->view_src == vcache && !vv) vv = t;
            else if (t->op == GGML_OP_VIEW && t->view_src == kcache && !kv) kv = t;
            else if (t->op == GGML_OP_PERMUTE && t->src[0] == rq && !qp) qp = t;
            else if (fa && t->op == GGML_OP_CPY && t->type == GGML_TYPE_F16 && t->src[0] && t->src[0]->type == GGML_TYPE_F32 && mask_cast < 0) mask_cast = i;
            else break;
            ++i;
        }
        if (!vv || !kv || !qp) return 0;
        const int64_t n_kv = kv->ne[1];
        // k view [dh, n_kv, Hkv] rows of the K cache from cell 0; q [dh, 1, H]
        if (kv->data != kcache->data || kv->ne[0] != dh || kv->ne[2] != Hkv || kv->ne[3] != 1 || kv->nb[0] != k_ts || kv->nb[1] != k_row || kv->nb[2] != k_hd) return 0;
        if (qp->ne[0] != dh || qp->ne[1] != 1 || qp->ne[2] != H || qp->nb[0] != 4 || qp->nb[2] != (size_t) dh * 4 || qp->data != rq->data) return 0;
        if (n_kv > n_ctx || cell >= n_kv) return 0;
        const ggml_tensor * mask = nullptr, * out_t = nullptr;
        float scale;
        if (!fa) {
            const ggml_tensor * kq = N(i), * sm = N(i + 1), * kqv = N(i + 2), * pm = N(i + 3), * ct = N(i + 4);
            if (!kq || !sm || !kqv || !pm || !ct) return 0;
            if (kq->op != GGML_OP_MUL_MAT || sm->op != GGML_OP_SOFT_MAX || kqv->op != GGML_OP_MUL_MAT || pm->op != GGML_OP_PERMUTE || ct->op != GGML_OP_CONT) return 0;
            // v view [n_kv, dh, Hkv] of the transposed V cache
            if (vv->data != vcache->data || vv->ne[0] != n_kv || vv->ne[1] != dh || vv->ne[2] != Hkv || vv->ne[3] != 1 || vv->nb[0] != 2 ||
                vv->nb[1] != (size_t) n_ctx * 2 || vv->nb[2] != (size_t) n_ctx * dh * 2) return 0;
            if (kq->src[0] != kv || kq->src[1] != qp || kq->type != GGML_TYPE_F32 || kq->ne[0] != n_kv || kq->ne[1] != 1 || kq->ne[2] != H) return 0;
            mask = sm->src[1];
            float max_bias; memcpy(&scale, sm->op_params, 4); memcpy(&max_bias, (const float *) sm->op_params + 1, 4);
            if (sm->src[0] != kq || max_bias != 0.0f || sm->ne[0] != n_kv) return 0;
            if (mask && (mask->type != GGML_TYPE_F32 || mask->ne[0] != n_kv || mask->nb[0] != 4 || mask->ne[2] != 1 || mask->ne[3] != 1)) return 0;
            if (kqv->src[0] != vv || kqv->src[1] != sm || kqv->ne[0] != dh || kqv->ne[1] != 1 || kqv->ne[2] != H) return 0;
            if (pm->src[0] != kqv || ct->src[0] != pm || !f32_vec(ct, Eq) || pm->ne[0] != dh || pm->ne[1] != H || pm->ne[2] != 1) return 0;
            out_t = ct;
            i += 5;
        } else {
            // FLASH_ATTN_EXT(q, k, v, mask F16) RESHAPE(-> [Eq, 1])  (llm_build_kqv, src/llama.cpp:10075-10095)
            const ggml_tensor * fe = N(i), * rs2 = N(i + 1);
            if (!fe || !rs2 || fe->op != GGML_OP_FLASH_ATTN_EXT || rs2->op != GGML_OP_RESHAPE || rs2->src[0] != fe) return 0;
            // v view [dh, n_kv, Hkv]: rows of the V cache from cell 0
            if (vv->data != vcache->data || vv->ne[0] != dh || vv->ne[1] != n_kv || vv->ne[2] != Hkv || vv->ne[3] != 1 || vv->nb[0] != v_ts ||
                vv->nb[1] != v_rowb || vv->nb[2] != v_hd) return 0;
            if (fe->src[0] != qp || fe->src[1] != kv || fe->src[2] != vv || fe->type != GGML_TYPE_F32 || fe->ne[0] != dh || fe->ne[1] != H || fe->ne[2] != 1 || fe->ne[3] != 1 ||
                !ggml_is_contiguous(fe)) return 0;
            float max_bias, softcap;
            memcpy(&scale, fe->op_params, 4); memcpy(&max_bias, (const float *) fe->op_params + 1, 4); memcpy(&softcap, (const float *) fe->op_params + 2, 4);
            if (max_bias != 0.0f || softcap != 0.0f) return 0;
            mask = fe->src[3];
            if (mask && (mask->type != GGML_TYPE_F16 || mask->ne[0] != n_kv || mask->nb[0] != 2 || mask->ne[2] != 1 || mask->ne[3] != 1)) return 0;
            if (mask_cast >= 0 && nodes_[mask_cast] != mask) return 0;  // some other cast sitting here: not ours to reorder
            if (!f32_vec(rs2, Eq) || rs2->data != fe->data) return 0;
            out_t = rs2;
            i += 2;
        }
        const int hi = i;
        const int out_idx = hi - 1;                                  // the CONT node / the RESHAPE of the flash-attention result
        // supported by the kernels?
        const bool q8 = kq8 || vq8;                                   // quantized cache: one workgroup per head only, head_dim 64 / 128
        // long contexts: keys split over workgroups; a quantized cache takes the matrix-core kernel over cached cells behind a rope + store launch (round 5)
        const bool can_split = (dh == 64 || dh == 128) && H / Hkv <= 8 &&
                               (!q8 || (fa && c_.attn_mfma && pm355_attn_cached_long_check((int) H, (int) Hkv, (int) dh, (int) n_ctx) == 0));
        const bool can_fused = dh == 64 || dh == 128 || (dh == 256 && !q8);
        bool split = n_kv >= c_.split_min && can_split;
        float * split_mem = nullptr;
        if (split) {
            split_mem = c_.split_scratch ? c_.split_scratch(c_.user, pm355_attn_split_scratch_floats((int) H, (int) dh, (int) n_ctx)) : nullptr;
            if (!split_mem) split = false;
        }
        // one workgroup per head: the score buffer lives in LDS, sized for split_min (+ one padding step of the cache) or, when the
        // split kernels cannot serve this shape, for the whole cache
        const int64_t max_keys = can_split ? c_.split_min + 32 : (q8 && n_ctx > 32768 ? 32768 : n_ctx);
        if (!split && !(can_fused && n_kv <= max_keys && (size_t) (4 * dh + (fa ? 2048 : 256) + max_keys + 16) * 4 <= 150 * 1024)) return 0;
        float * sq = c_.qkv_scratch ? c_.qkv_scratch(c_.user, (size_t) Eq, (size_t) Ekv) : nullptr;
        if (!sq) return 0;
        float * sk = sq + Eq, * sv = sk + Ekv;
        // one cell / cells-attended pair per graph
        if (p.has_attn && (cell_ != cell || n_kv_ != n_kv)) return 0;
        const int outs[2] = {out_idx, mask_cast};
        if (!range_private(i0, hi, outs, mask_cast >= 0 ? 2 : 1)) return 0;
        if (mask_cast >= 0) emit_node(p, mask_cast);                 // the F16 mask is produced first: later layers' graphs-in-graph read it too
        // launch 1: norm + QKV
        pm355_matvec_job jobs[3] = { job_of(mq, sq, bq, nullptr), job_of(mk, sk, bk, nullptr), job_of(mv, sv, bv, nullptr) };
        const bool one_launch = pm355_mul_mat_vec_fused_check(jobs, 3, E) == 0;
        if (!one_launch) for (int j = 0; j < 3; ++j) if (pm355_mul_mat_vec_fused_check(jobs + j, 1, E)) return 0;
        // round-3 form: NORM-mode rope, F16 caches, one-workgroup-per-head regime, every workgroup's row slices hold whole rotation pairs, and
        // the same cos / sin table as the layers planned before (same parameters, positions and frequency factors)
        pm355_rope_params rp_; rope_params_of(rq, rp_);
        const float * ff_ = rq->src[2] ? (const float *) rq->src[2]->data : nullptr;
        bool epi = c_.rope_tab && one_launch && !q8 && (!split || (c_.attn_mfma && pm355_attn_cached_long_check((int) H, (int) Hkv, (int) dh, (int) n_ctx) == 0)) && (rp_.mode == 0 || rp_.mode == 2) && rp_.n_dims <= 256 &&
                   pm355_mul_mat_vec_qkv_check_ex(jobs, E, (int) Hkv, (int) dh, rp_.n_dims, rp_.mode & 2) == 0;
        if (epi && tab_set_) {
            const bool same_ff = ff_ == tab_ff_ || (ff_ && tab_ff_ && c_.same_bytes && c_.same_bytes(c_.user, ff_, tab_ff_, (size_t) rp_.n_dims / 2 * 4));
            if (memcmp(&rp_, &tab_rp_, sizeof(rp_)) || rq->src[1]->data != tab_pos_ || !same_ff) epi = false;
        }
        if (epi && !tab_set_) {
            step t; memset(&t, 0, sizeof(t));
            t.kind = STEP_ROPE_TAB; t.node = -1; t.node_lo = i0; t.node_hi = i0;
            t.rope = rp_; t.attn.d_pos = (const int32_t *) rq->src[1]->data; t.attn.freq_factors = ff_; t.qs.rope_table = c_.rope_tab;
            p.steps.push_back(t);
            tab_set_ = true; tab_rp_ = rp_; tab_pos_ = rq->src[1]->data; tab_ff_ = ff_;
        }
        if (epi) {
            step g; memset(&g, 0, sizeof(g));
            g.kind = STEP_QKV; g.node = -1; g.node_lo = i0; g.node_hi = hi;
            g.K = E; g.njobs = 3; g.eps = eps; g.x = (const float *) x->data; g.norm_w = nw;
            for (int j = 0; j < 3; ++j) g.job[j] = jobs[j];
            g.qs.rope_table = c_.rope_tab; g.qs.d_pos = nullptr; g.qs.d_cell_nkv = c_.d_dyn; g.qs.k_cache = kcache->data; g.qs.v_cache = vcache->data;
            g.qs.n_head_kv = (int32_t) Hkv; g.qs.head_dim = (int32_t) dh; g.qs.n_ctx = (int32_t) n_ctx; g.qs.n_rot = rp_.n_dims; g.qs.v_rowmajor = fa ? 1 : 0; g.qs.rope_neox = (rp_.mode & 2) ? 1 : 0;
            p.steps.push_back(g);
            ++p.n_gemv;
        } else if (one_launch) push_gemv(p, i0, hi, jobs, 3, E, (const float *) x->data, nw, eps);
        else for (int j = 0; j < 3; ++j) push_gemv(p, i0, hi, jobs + j, 1, E, (const float *) x->data, nw, eps);
        // launch 2: rope + KV store + attention (round-3 form: attention over cached cells)
        step s; memset(&s, 0, sizeof(s));
        s.kind = epi ? STEP_ATTN_CACHED : STEP_ATTN; s.node = -1; s.node_lo = i0; s.node_hi = hi;
        s.attn.q = sq; s.attn.k = sk; s.attn.v = sv; s.attn.k_cache = kcache->data; s.attn.v_cache = vcache->data;
        s.attn.d_pos = (const int32_t *) rq->src[1]->data; s.attn.d_cell_nkv = c_.d_dyn;
        s.attn.mask = mask ? mask->data : nullptr;
        s.attn.flags = (fa ? (PM355_ATTN_V_ROWMAJOR | PM355_ATTN_MASK_F16) : 0) | (kq8 ? PM355_ATTN_K_Q8_0 : 0) | (vq8 ? PM355_ATTN_V_Q8_0 : 0);
        s.attn.freq_factors = rq->src[2] ? (const float *) rq->src[2]->data : nullptr;
        s.attn.out = (float *) out_t->data; s.attn.scratch = split ? split_mem : nullptr;
        s.attn.n_head = (int32_t) H; s.attn.n_head_kv = (int32_t) Hkv; s.attn.head_dim = (int32_t) dh; s.attn.n_ctx = (int32_t) n_ctx;
        int64_t grid_cells = 1024;                                    // long-context grid: power-of-two bucket of the cells attended, so that the plan
        while (grid_cells < n_kv) grid_cells *= 2;                    // (and its captured hipGraph) survives the steps of n_kv inside a bucket
        if (grid_cells > n_ctx) grid_cells = n_ctx;
        s.attn.split = split ? 1 : 0; s.attn.max_keys = split ? (int32_t) grid_cells : (int32_t) max_keys; s.attn.kq_scale = scale;
        rope_params_of(rq, s.rope);
        p.steps.push_back(s);
        ++p.n_attn;
        if (!p.has_attn) {
            p.has_attn = true; cell_ = cell; n_kv_ = n_kv;
            for (int j = i0; j < hi; ++j) { if (nodes_[j] == kcp) p.i_kcell = j; if (nodes_[j] == kv) p.i_kview = j; }
        }
        return hi - i0;
    }
    int64_t cell_ = -1, n_kv_ = -1;
    bool tab_set_ = false; pm355_rope_params tab_rp_; const void * tab_pos_ = nullptr; const float * tab_ff_ = nullptr;

    // ---- launch 4 (ffn gate/up pair) and the head: RMS_NORM MUL then mat-vecs that all read the normalised row ----------------
    int try_norm_matvec(plan & p, int i0) {
        const ggml_tensor * x; const float * nw; float eps;
        const ggml_tensor * xn = match_norm(i0, x, nw, eps);
        if (!xn) return 0;
        const int64_t E = x->ne[0];
        int i = i0 + 2;
        // (a) MUL_MAT(gate) UNARY(silu) MUL_MAT(up) MUL(silu, up)   [build order of llm_build_ffn, LLM_FFN_SILU + LLM_FFN_PAR]
        {
            const ggml_tensor * a = N(i), * u = N(i + 1), * b = N(i + 2), * m = N(i + 3);
            if (is_matvec(a, xn) && is_matvec(b, xn) && u && m && u->op == GGML_OP_UNARY && ggml_get_unary_op(u) == GGML_UNARY_OP_SILU &&
                u->src[0] == a && f32_vec(u, a->ne[0]) && m->op == GGML_OP_MUL && f32_vec(m, a->ne[0]) && a->ne[0] == b->ne[0] &&
                ((m->src[0] == u && m->src[1] == b) || (m->src[0] == b && m->src[1] == u)) && a->src[0]->type == b->src[0]->type) {
                pm355_matvec_job job = job_of(a, (float *) m->data, nullptr, nullptr, b->src[0]);     // y = silu(gate.x) * (up.x)
                const int out_idx = i + 3;
                if (pm355_mul_mat_vec_fused_check(&job, 1, E) == 0 && gemv_alias_ok(&job, 1, E, (const float *) x->data) &&
                    range_private(i0, i + 4, &out_idx, 1)) {
                    push_gemv(p, i0, i + 4, &job, 1, E, (const float *) x->data, nw, eps);
                    return i + 4 - i0;
                }
            }
        }
        // (b) up to 3 plain projections of the normalised row, each optionally + bias (lm_head: one)
        {
            pm355_matvec_job jobs[3]; int outs[3]; int nj = 0; int j = i;
            while (nj < 3) {
                const ggml_tensor * mm, * out; const float * bias; int jj = j;
                if (!match_proj(jj, xn, mm, bias, out)) break;
                jobs[nj] = job_of(mm, (float *) out->data, bias, nullptr); outs[nj] = jj - 1; ++nj; j = jj;
            }
            while (nj > 0) {
                bool ok = pm355_mul_mat_vec_fused_check(jobs, nj, E) == 0 && gemv_alias_ok(jobs, nj, E, (const float *) x->data) &&
                          range_private(i0, outs[nj - 1] + 1, outs, nj);
                if (ok) { push_gemv(p, i0, outs[nj - 1] + 1, jobs, nj, E, (const float *) x->data, nw, eps); return outs[nj - 1] + 1 - i0; }
                --nj;
            }
        }
        return 0;
    }

    // ---- multi-token batches (prompt processing): VIEW(v) VIEW(k) PERMUTE(q) | MUL_MAT(k,q) SOFT_MAX(mask) MUL_MAT(v,kq) PERMUTE CONT of
    //      llm_build_kqv -> ONE launch of the MFMA attention with the additive mask (pm355_attn_prefill_masked). The rotations and the
    //      KV-store copies in front of it stay node-equivalent (the graph's Qcur / Kcur / Vcur buffers alias each other, so the chain can
    //      only be entered after ROPE materialised q and the cache holds the batch).
    int try_batch_attention(plan & p, int i0) {
        const ggml_tensor * vv = nullptr, * kv = nullptr, * qp = nullptr;
        int i = i0;
        for (int r = 0; r < 3; ++r) {
            const ggml_tensor * t = N(i);
            if (!t) return 0;
            if (t->op == GGML_OP_VIEW && t->view_src && t->type == GGML_TYPE_F16 && !kv && t->nb[1] > t->nb[2] && t->ne[3] == 1) kv = t;   // k [dh, n_kv, Hkv]: nb1 = cache row > nb2 = dh*2 (v: nb1 = n_ctx*2 < nb2)
            else if (t->op == GGML_OP_VIEW && t->view_src && t->type == GGML_TYPE_F16 && !vv) vv = t;
            else if (t->op == GGML_OP_PERMUTE && !qp) qp = t;
            else return 0;
            ++i;
        }
        if (!vv || !kv || !qp) return 0;
        const ggml_tensor * kq = N(i), * sm = N(i + 1), * kqv = N(i + 2), * pm = N(i + 3), * ct = N(i + 4);
        if (!kq || !sm || !kqv || !pm || !ct) return 0;
        if (kq->op != GGML_OP_MUL_MAT || sm->op != GGML_OP_SOFT_MAX || kqv->op != GGML_OP_MUL_MAT || pm->op != GGML_OP_PERMUTE || ct->op != GGML_OP_CONT) return 0;
        const ggml_tensor * kcache = kv->view_src, * vcache = vv->view_src;
        if (kcache->view_src || vcache->view_src || kcache->op != GGML_OP_NONE || vcache->op != GGML_OP_NONE || !ggml_is_contiguous(kcache) || !ggml_is_contiguous(vcache)) return 0;
        if (kv->data != kcache->data || vv->data != vcache->data) return 0;
        const int64_t dh = kv->ne[0], n_kv = kv->ne[1], Hkv = kv->ne[2];
        const ggml_tensor * qr = qp->src[0];                            // the rotated queries [dh, H, T] f32 contiguous
        if (!qr || qr->type != GGML_TYPE_F32 || !ggml_is_contiguous(qr) || qr->ne[0] != dh || qr->ne[3] != 1) return 0;
        const int64_t H = qr->ne[1], T = qr->ne[2];
        if (T < 2 || (dh != 64 && dh != 128) || Hkv < 1 || H % Hkv) return 0;
        if (kv->nb[0] != 2 || kv->nb[1] != (size_t) (Hkv * dh * 2) || kv->nb[2] != (size_t) (dh * 2)) return 0;
        const int64_t n_ctx = (int64_t) (vv->nb[1] / 2);
        if (vv->ne[0] != n_kv || vv->ne[1] != dh || vv->ne[2] != Hkv || vv->ne[3] != 1 || vv->nb[0] != 2 || vv->nb[2] != (size_t) (n_ctx * dh * 2)) return 0;
        if (n_ctx <= 0 || n_ctx % 32 || n_kv > n_ctx || n_kv % 4 || (int64_t) ggml_nelements(kcache) < n_ctx * Hkv * dh || (int64_t) ggml_nelements(vcache) < n_ctx * Hkv * dh) return 0;
        if (qp->data != qr->data || qp->ne[0] != dh || qp->ne[1] != T || qp->ne[2] != H || qp->nb[0] != 4 || qp->nb[1] != (size_t) (H * dh * 4) || qp->nb[2] != (size_t) (dh * 4)) return 0;
        if (kq->src[0] != kv || kq->src[1] != qp || kq->type != GGML_TYPE_F32 || kq->ne[0] != n_kv || kq->ne[1] != T || kq->ne[2] != H) return 0;
        const ggml_tensor * mask = sm->src[1];
        float scale, max_bias; memcpy(&scale, sm->op_params, 4); memcpy(&max_bias, (const float *) sm->op_params + 1, 4);
        if (sm->src[0] != kq || max_bias != 0.0f || !mask || mask->type != GGML_TYPE_F32 || mask->ne[0] != n_kv || mask->ne[1] < T || mask->nb[0] != 4 ||
            mask->nb[1] % 16 || mask->ne[2] != 1 || mask->ne[3] != 1) return 0;
        if (kqv->src[0] != vv || kqv->src[1] != sm || kqv->ne[0] != dh || kqv->ne[1] != T || kqv->ne[2] != H) return 0;
        if (pm->src[0] != kqv || ct->src[0] != pm || ct->type != GGML_TYPE_F32 || !ggml_is_contiguous(ct) || ct->ne[0] != H * dh || ct->ne[1] != T ||
            pm->ne[0] != dh || pm->ne[1] != H || pm->ne[2] != T) return 0;
        const int hi = i + 5, out_idx = hi - 1;
        if (!range_private(i0, hi, &out_idx, 1)) return 0;
        if (overlap(ct->data, ggml_nbytes(ct), qr->data, ggml_nbytes(qr))) return 0;       // every query row is read by 1 workgroup, written by it last
        step s; memset(&s, 0, sizeof(s));
        s.kind = STEP_ATTN_BATCH; s.node = -1; s.node_lo = i0; s.node_hi = hi;
        s.ab.q = (const float *) qr->data; s.ab.kc = kcache->data; s.ab.vc = vcache->data; s.ab.mask = mask->data;
        s.ab.mask_stride = (int64_t) (mask->nb[1] / 4); s.ab.out = (float *) ct->data;
        s.ab.n_tokens = (int32_t) T; s.ab.n_head = (int32_t) H; s.ab.n_head_kv = (int32_t) Hkv; s.ab.head_dim = (int32_t) dh;
        s.ab.n_ctx = (int32_t) n_ctx; s.ab.n_kv = (int32_t) n_kv; s.ab.scale = scale;
        p.steps.push_back(s);
        ++p.n_attn_batch;
        return hi - i0;
    }

    // ---- the same for flash-attention graphs: PERMUTE(q) VIEW(k) VIEW(v) [CPY(mask F32 -> F16)] FLASH_ATTN_EXT RESHAPE over a multi-token
    //      batch (llm_build_kqv, src/llama.cpp:10075-10095) -> one launch of the MFMA attention on the row-major F16 V cache
    int try_batch_flash_attention(plan & p, int i0) {
        int i = i0, mask_cast = -1, n_pre = 0;
        for (; n_pre < 4; ++n_pre, ++i) {
            const ggml_tensor * t = N(i);
            if (!t) return 0;
            if (t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE) continue;
            if (t->op == GGML_OP_CPY && t->type == GGML_TYPE_F16 && t->src[0] && t->src[0]->type == GGML_TYPE_F32 && mask_cast < 0) { mask_cast = i; continue; }
            break;
        }
        const ggml_tensor * fe = N(i), * rs2 = N(i + 1);
        if (n_pre < 3 || !fe || !rs2 || fe->op != GGML_OP_FLASH_ATTN_EXT || rs2->op != GGML_OP_RESHAPE || rs2->src[0] != fe || rs2->data != fe->data) return 0;
        const ggml_tensor * qp = fe->src[0], * kv = fe->src[1], * vv = fe->src[2], * mask = fe->src[3];
        // the three operands are the nodes in front of the op
        int found = 0;
        for (int j = i0; j < i; ++j) if (nodes_[j] == qp || nodes_[j] == kv || nodes_[j] == vv) ++found;
        if (found != 3 || qp->op != GGML_OP_PERMUTE || kv->op != GGML_OP_VIEW || vv->op != GGML_OP_VIEW) return 0;
        const ggml_tensor * kcache = kv->view_src, * vcache = vv->view_src;
        if (!kcache || !vcache || kcache->type != GGML_TYPE_F16 || vcache->type != GGML_TYPE_F16 || kcache->view_src || vcache->view_src ||
            kcache->op != GGML_OP_NONE || vcache->op != GGML_OP_NONE || !ggml_is_contiguous(kcache) || !ggml_is_contiguous(vcache)) return 0;
        if (kv->data != kcache->data || vv->data != vcache->data) return 0;
        const int64_t dh = kv->ne[0], n_kv = kv->ne[1], Hkv = kv->ne[2];
        const ggml_tensor * qr = qp->src[0];                            // the rotated queries [dh, H, T] f32 contiguous
        if (!qr || qr->type != GGML_TYPE_F32 || !ggml_is_contiguous(qr) || qr->ne[0] != dh || qr->ne[3] != 1) return 0;
        const int64_t H = qr->ne[1], T = qr->ne[2];
        if (T < 2 || (dh != 64 && dh != 128) || Hkv < 1 || H % Hkv) return 0;
        const size_t row = (size_t) (Hkv * dh * 2);
        if (kv->type != GGML_TYPE_F16 || kv->nb[0] != 2 || kv->nb[1] != row || kv->nb[2] != (size_t) (dh * 2) || kv->ne[3] != 1) return 0;
        if (vv->type != GGML_TYPE_F16 || vv->ne[0] != dh || vv->ne[1] != n_kv || vv->ne[2] != Hkv || vv->ne[3] != 1 || vv->nb[0] != 2 || vv->nb[1] != row || vv->nb[2] != (size_t) (dh * 2)) return 0;
        const int64_t n_ctx = (int64_t) ggml_nelements(kcache) / (Hkv * dh);
        if (n_ctx <= 0 || n_ctx % 32 || n_kv > n_ctx || n_kv % 4 || (int64_t) ggml_nelements(vcache) < n_ctx * Hkv * dh) return 0;
        if (qp->data != qr->data || qp->ne[0] != dh || qp->ne[1] != T || qp->ne[2] != H || qp->nb[0] != 4 || qp->nb[1] != (size_t) (H * dh * 4) || qp->nb[2] != (size_t) (dh * 4)) return 0;
        float scale, max_bias, softcap;
        memcpy(&scale, fe->op_params, 4); memcpy(&max_bias, (const float *) fe->op_params + 1, 4); memcpy(&softcap, (const float *) fe->op_params + 2, 4);
        if (max_bias != 0.0f || softcap != 0.0f) return 0;
        if (!mask || mask->type != GGML_TYPE_F16 || mask->ne[0] != n_kv || mask->ne[1] < T || mask->nb[0] != 2 || mask->nb[1] % 8 || mask->ne[2] != 1 || mask->ne[3] != 1) return 0;
        if (mask_cast >= 0 && nodes_[mask_cast] != mask) return 0;
        if (fe->type != GGML_TYPE_F32 || !ggml_is_contiguous(fe) || fe->ne[0] != dh || fe->ne[1] != H || fe->ne[2] != T || fe->ne[3] != 1) return 0;
        const int hi = i + 2, out_idx = hi - 1;
        const int outs[2] = {out_idx, mask_cast};
        if (!range_private(i0, hi, outs, mask_cast >= 0 ? 2 : 1)) return 0;
        if (overlap(fe->data, ggml_nbytes(fe), qr->data, ggml_nbytes(qr))) return 0;
        if (mask_cast >= 0) emit_node(p, mask_cast);
        step s; memset(&s, 0, sizeof(s));
        s.kind = STEP_ATTN_BATCH; s.node = -1; s.node_lo = i0; s.node_hi = hi;
        s.ab.q = (const float *) qr->data; s.ab.kc = kcache->data; s.ab.vc = vcache->data; s.ab.mask = mask->data;
        s.ab.mask_stride = (int64_t) (mask->nb[1] / 2); s.ab.out = (float *) fe->data;
        s.ab.n_tokens = (int32_t) T; s.ab.n_head = (int32_t) H; s.ab.n_head_kv = (int32_t) Hkv; s.ab.head_dim = (int32_t) dh;
        s.ab.n_ctx = (int32_t) n_ctx; s.ab.n_kv = (int32_t) n_kv; s.ab.scale = scale; s.ab.flags = PM355_ATTN_V_ROWMAJOR | PM355_ATTN_MASK_F16;
        p.steps.push_back(s);
        ++p.n_attn_batch;
        return hi - i0;
    }

    // ---- launches 3 and 5: MUL_MAT(W, x) [ADD(mm, residual)] with the f32 row quantized in the kernel prologue -----------------
    int try_matvec_resid(plan & p, int i0) {
        const ggml_tensor * mm = N(i0);
        if (!is_matvec(mm)) return 0;
        const ggml_tensor * x = mm->src[1];
        const int64_t K = x->ne[0];
        const ggml_tensor * ad = N(i0 + 1);
        if (ad && ad->op == GGML_OP_ADD && f32_vec(ad, mm->ne[0]) && (ad->src[0] == mm || ad->src[1] == mm)) {
            const ggml_tensor * r = ad->src[0] == mm ? ad->src[1] : ad->src[0];
            if (f32_vec(r, mm->ne[0])) {
                pm355_matvec_job job = job_of(mm, (float *) ad->data, nullptr, (const float *) r->data);
                const int out_idx = i0 + 1;
                if (pm355_mul_mat_vec_fused_check(&job, 1, K) == 0 && gemv_alias_ok(&job, 1, K, (const float *) x->data) &&
                    range_private(i0, i0 + 2, &out_idx, 1)) {
                    push_gemv(p, i0, i0 + 2, &job, 1, K, (const float *) x->data, nullptr, 0.0f);
                    return 2;
                }
            }
        }
        // last layer of build_llama (src/llama.cpp:11120-11125): GET_ROWS(mm, out_ids) GET_ROWS(residual, out_ids) ADD. With one token
        // in the batch both sources have a single row, so the only valid index is 0 and each GET_ROWS is a copy of its source row
        {
            const ggml_tensor * g1 = N(i0 + 1), * g2 = N(i0 + 2), * a2 = N(i0 + 3);
            auto row_copy = [](const ggml_tensor * gr, int64_t n) {
                return gr && gr->op == GGML_OP_GET_ROWS && f32_vec(gr, n) && f32_vec(gr->src[0], n) && gr->src[1] &&
                       gr->src[1]->type == GGML_TYPE_I32 && ggml_nelements(gr->src[1]) == 1;
            };
            if (row_copy(g1, mm->ne[0]) && row_copy(g2, mm->ne[0]) && g1->src[0] == mm && g1->src[1] == g2->src[1] && a2 && a2->op == GGML_OP_ADD &&
                f32_vec(a2, mm->ne[0]) && ((a2->src[0] == g1 && a2->src[1] == g2) || (a2->src[0] == g2 && a2->src[1] == g1))) {
                pm355_matvec_job job = job_of(mm, (float *) a2->data, nullptr, (const float *) g2->src[0]->data);
                const int out_idx = i0 + 3;
                if (pm355_mul_mat_vec_fused_check(&job, 1, K) == 0 && gemv_alias_ok(&job, 1, K, (const float *) x->data) &&
                    range_private(i0, i0 + 4, &out_idx, 1)) {
                    push_gemv(p, i0, i0 + 4, &job, 1, K, (const float *) x->data, nullptr, 0.0f);
                    return 4;
                }
            }
        }
        pm355_matvec_job job = job_of(mm, (float *) mm->data, nullptr, nullptr);
        if (pm355_mul_mat_vec_fused_check(&job, 1, K) == 0 && gemv_alias_ok(&job, 1, K, (const float *) x->data)) {
            push_gemv(p, i0, i0 + 1, &job, 1, K, (const float *) x->data, nullptr, 0.0f);
            return 1;
        }
        return 0;
    }
};

// two plans describe the same launch sequence (the per-token cell / cells-attended pair is not part of a plan)
inline bool plan_equal(const plan & a, const plan & b) {
    return a.steps.size() == b.steps.size() && a.single_token == b.single_token && a.has_attn == b.has_attn &&
           (a.steps.empty() || memcmp(a.steps.data(), b.steps.data(), a.steps.size() * sizeof(step)) == 0);
}

// the per-token values of a planned graph
inline bool plan_dyn(struct ggml_cgraph * g, const plan & p, int32_t & cell, int32_t & n_kv) {
    cell = 0; n_kv = 0;
    if (!p.has_attn) return true;
    if (p.i_kcell < 0 || p.i_kview < 0 || p.i_kcell >= ggml_graph_n_nodes(g) || p.i_kview >= ggml_graph_n_nodes(g)) return false;
    const ggml_tensor * cp = ggml_graph_node(g, p.i_kcell), * kv = ggml_graph_node(g, p.i_kview);
    if (cp->op != GGML_OP_CPY || !cp->view_src || kv->op != GGML_OP_VIEW) return false;
    const size_t row = ggml_row_size(cp->type, cp->ne[0]);
    cell = (int32_t) (((const char *) cp->data - (const char *) cp->view_src->data) / row);
    n_kv = (int32_t) kv->ne[1];
    return true;
}

} // namespace mi355
