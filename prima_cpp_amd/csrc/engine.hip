// engine.hip — resident decoder for one piped-ring layer window (C ABI part (B) of include/prima_mi355.h).
//
// What it replaces in the reference (per token, per rank): llama_build_graph + ggml_backend_sched_alloc_graph +
// llama_graph_compute + the D2H/H2D activation bounce of llama_decode_internal's ring loop
// (src/llama.cpp:18455-18564). Here the layer window's kernel sequence is fixed at finalize() time,
// positions live in device memory, and the whole single-token step is one hipGraph that is replayed.
//
// HBM layout: all weights of the window resident (288 GB per GPU: Llama-3-70B Q4_K_M is 42.5 GB),
// one allocation per tensor in the layouts of mmvq.hip / repack.hip; KV cache F16, K [n_ctx][Hkv*dh],
// V transposed [Hkv*dh][n_ctx] per layer (reference layout without flash-attn, src/llama.cpp:9707-9714).
#include "../../include/prima_mi355.h"
#include "pm355_device.h"
#include "pm355_kernels.h"
#include "pm355_layer_ops.h"
#include "pm355_engine.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <initializer_list>
#include <algorithm>

namespace {

struct Tensor { void * d = nullptr; int type = -1; int64_t K = 0, N = 0; size_t bytes = 0; };

struct Layer { Tensor t[12]; void * kc = nullptr; void * vc = nullptr; void * host[12] = {}; size_t hbm[12] = {}; };
// weight streaming (models larger than HBM): a device slot holds the tensors of ONE layer; layers are cycled through n slots
struct Slot { void * d[12] = {}; hipEvent_t ready = nullptr, free_ = nullptr; int layer = -1; };

} // namespace

struct pm355_model {
    pm355_hparams hp;
    int lo, hi, flags;
    std::vector<Layer> layers;            // hi - lo
    Tensor tok_embd, out_norm, output, rope_freqs;
    int max_tokens = 0;
    bool finalized = false;
    pm_rope_cfg rope;
    // scratch (device)
    float * x = nullptr, * x1 = nullptr, * q = nullptr, * k = nullptr, * v = nullptr, * att = nullptr, * h = nullptr;
    float * logits = nullptr, * xn = nullptr, * h2 = nullptr;
    uint8_t * aq_k = nullptr, * aq_0 = nullptr;     // quantized activation scratch (Q8_K / Q8_0), sized for max(K)
    int32_t * d_pos = nullptr, * d_tok = nullptr, * d_ctl = nullptr;   // d_pos[n_seq]; d_ctl = {current seq, n_seq}
    int n_seq = 1;
    bool no_fuse = false;                 // PM355_NO_FUSE=1: node-by-node kernels (debug / A-B)
    bool no_multi = false;                // PM355_NO_MMQ_MULTI=1
    bool no_small_cols = false;           // PM355_SMALL_COLS=0
    bool no_small_epi = false;            // PM355_SMALL_ROPE_EPI=0: small batches run rope_kv_store as a launch of its own
    bool no_mmq = false;                  // PM355_NO_MMQ_I8=1: 4..64-token batches on the round-1 paths (mat-vec columns, F16 GEMM from 16 tokens)
    // single-token decode, short contexts, NORM-mode rope: RoPE + F16 KV store happen in the epilogue of the wq | wk | wv launch (per-token cos / sin
    // table `rope_tab`), the attention launch reads everything from the cache (attn_cached.hip). PM355_QKV_EPI=0: the round-2 form (raw q / k / v,
    // rope + store inside the attention kernel)
    bool qkv_epi = true; float * rope_tab = nullptr;
    // producer-side sum of squares (round 5): the wo / ffn_down launches leave the per-workgroup partials of the NEXT rms_norm's sum of squares
    // in ss[0] / ss[1] (256 doubles each); the consuming wq | wk | wv, ffn_gate | ffn_up and lm_head launches add them instead of reducing the row.
    // OPT-IN (PM355_SS=1): the same bits, and a net LOSS of 0.4 % on the 70B token - the partials cost the wo / ffn_down launches 0.8 us each and buy
    // the consumers nothing measurable (their reduction pass hides behind the pre-issued weight loads): profiles/r05_ab_sumsq_q6k_tail.txt. The code
    // lives in kernel instantiations of its own (PM_FEAT_SS) because its mere presence cost every launch 0.55 us. Default: every prologue reduces its row
    bool use_ss = false; double * ss = nullptr;
    // attention in the tail of the wq | wk | wv launch (round 5, pm_qkv_epi::att_out): the short-context regime's attention launch and its boundary
    // disappear - four launches per layer. att_tk: 64 per-KV-head ticket counters + the watchdog word. OPT-IN (PM355_ATTN_TAIL=1): the same bits as the
    // separate launch and 1.6 % SLOWER per 70B token (8.955 against 8.817 ms, interleaved A/B, profiles/r05_attention_tail.txt) - q and the new cell
    // cross XCDs inside the launch (write-through store, ticket, cache-bypassing loads: ~6 us) where the kernel boundary + attn_cached cost 4.5
    bool attn_tail = false; unsigned * att_tk = nullptr;
    // the persistent decode engine (decode_engine.hip, round 5): the whole layer stack of a single-token step as ONE launch. Plans are keyed on the
    // activation pointers baked into their phase tables. OPT-IN (PM355_ENGINE=1): bit-identical to the five launches per layer (run_layers_fused) and
    // measured SLOWER than them - 10.2-10.4 ms against 8.55 ms per 70B token, 2.23 against 1.67 ms on the 8B shape (profiles/r05_engine_measured.txt):
    // the default stays the five launches
    // Hand-off buffers are WRITE-ONCE per launch: every layer has its own q / attention output / ffn activation / residual rows / partial sums
    // (eng_act, ~250 KB per 70B layer). A buffer re-used by the next layer was served stale from the reading XCD's L2 whatever the load's scope
    // bits (two 70B layers differed from the launches by 1e-4, one by 1e-14: found on the hardware; round 1's persistent kernel had met the same).
    struct EnginePlan { const float * in; float * out; pm_eng_plan * plan; int n_ss_end; const double * ss_end; const float * end; float * act; double * ss_in0; };
    bool use_engine = false; std::vector<EnginePlan> eng_plans; bool eng_refused = false;
    // PM355_PROMPT_I8=1: prompts (> MMQ_MAX_TOKENS = 32 tokens) run their Q4_K / Q6_K matrices on the integer matrix cores over Q8_K activations (mmq_big.hip) - the CPU
    // reference's own arithmetic, at 0.6-0.75 of the F16 GEMMs' rate (mmq.hip, the default); tab_big = the activation tables of the current
    // activation set (pm_q8k_tables)
    bool no_big = true; uint8_t * tab_big = nullptr;
    // WINDOW STREAMING (pm355_model_set_streaming): the layer tensors live in pinned HOST memory (in the HBM layout) and are streamed
    // through `slots` device-side layer slots by a copy stream, slot (l - lo) % n_slots for layer l, one layer ahead of the compute
    // stream per free slot - the GPU-side form of prima.cpp's "prefetch the next layer window while this one computes"
    // (manage_graph_tensors + posix_madvise, src/llama.cpp:18152-18218, :18566-18575) with hipMemcpyAsync from pinned memory in place of
    // mmap page faults. KV caches, embedding and head stay resident.
    int n_slots = 0;
    std::vector<Slot> slots;
    hipStream_t copy_stream = nullptr;
    size_t slot_bytes[12] = {};
    uint64_t streamed_bytes = 0, stream_step = 0;
    // long-context decode attention (attn_split.hip): the host mirrors the device position counters to choose, per step, between
    // the fused one-workgroup-per-head kernel and the keys-split-over-workgroups path (different launch sequences = different
    // captured graphs). PM355_ATTN_SPLIT_MIN positions (default 320: the crossover on the 70B head shape once the matrix-core kernel existed - one workgroup
    // per head costs 12.8 / 15.4 / 18.1 us at 384 / 512 / 640 cells, the matrix-core kernel ~9; rounds 2-4 kept 640, measured against attn_flash.hip).
    float * split_scratch = nullptr;
    std::vector<int> h_pos; int h_seq = 0; int split_min = 320; bool long_ctx = false;
    bool attn_mfma = true;                        // long contexts in the QKV-epilogue form: matrix-core attention (attn_flash_mfma.hip); PM355_ATTN_MFMA=0 = round-2 kernel
    int flash_cells = 0; bool use_flash = true;   // long contexts: one-launch flash-decoding (attn_flash.hip), grid sized for `flash_cells` (a power-of-two bucket of the position)
    // staging for set_tensor
    pm355_uploader * up = nullptr;        // pinned ring + copier threads + private stream (upload.hip)
    hipStream_t cap_stream = nullptr;
    hipStream_t side = nullptr; hipEvent_t side_a = nullptr, side_b = nullptr;   // prefill: wk and wv GEMMs (64 workgroups each) run side by side
    // captured single-token step graphs, keyed on everything that is baked into the kernel arguments
    struct StepGraph { const void * in, * tok; void * out, * logits, * argmax; int adv, rot, head, regime; hipGraphExec_t exec; };
    std::vector<StepGraph> graphs;
    char err[256];
};

namespace {

int tensor_shape(const pm355_model * m, int kind, int64_t & K, int64_t & N) {
    const pm355_hparams & h = m->hp;
    const int64_t E = h.n_embd, Eq = (int64_t) h.head_dim * h.n_head, Ekv = (int64_t) h.head_dim * h.n_head_kv, F = h.n_ff;
    switch (kind) {
        case PM355_T_ATTN_NORM: case PM355_T_FFN_NORM: case PM355_T_OUT_NORM: K = E; N = 1; return 0;
        case PM355_T_WQ: K = E; N = Eq; return 0;
        case PM355_T_WK: case PM355_T_WV: K = E; N = Ekv; return 0;
        case PM355_T_WO: K = Eq; N = E; return 0;
        case PM355_T_FFN_GATE: case PM355_T_FFN_UP: K = E; N = F; return 0;
        case PM355_T_FFN_DOWN: K = F; N = E; return 0;
        case PM355_T_BQ: K = Eq; N = 1; return 0;
        case PM355_T_BK: case PM355_T_BV: K = Ekv; N = 1; return 0;
        case PM355_T_TOK_EMBD: case PM355_T_OUTPUT: K = E; N = h.n_vocab; return 0;
        case PM355_T_ROPE_FREQS: K = h.head_dim / 2; N = 1; return 0;
    }
    return -1;
}

Tensor * tensor_slot(pm355_model * m, int kind, int layer) {
    if (kind < 12) {
        if (layer < m->lo || layer >= m->hi) return nullptr;
        return &m->layers[layer - m->lo].t[kind];
    }
    switch (kind) {
        case PM355_T_TOK_EMBD: return &m->tok_embd;
        case PM355_T_OUT_NORM: return &m->out_norm;
        case PM355_T_OUTPUT: return &m->output;
        case PM355_T_ROPE_FREQS: return &m->rope_freqs;
    }
    return nullptr;
}

int seterr(pm355_model * m, int code, const char * msg) {
    const hipError_t e = hipGetLastError();        // also clears the sticky error
    if (code == PM355_E_HIP) snprintf(m->err, sizeof(m->err), "%s: %s", msg, hipGetErrorString(e));
    else snprintf(m->err, sizeof(m->err), "%s", msg);
    return code;
}
// hipGetLastError() is sticky across unrelated earlier calls: test-and-clear around OUR launches only
bool hip_ok() { return hipGetLastError() == hipSuccess; }

size_t g_last_hbm = 0;        // HBM bytes of the tensor alloc_tensor allocated last (single-threaded loader)
int alloc_tensor(pm355_model * m, Tensor * t, int kind, int type) {
    int64_t K, N;
    if (tensor_shape(m, kind, K, N)) return PM355_E_UNSUPPORTED;
    const size_t rb = pm_weight_row_bytes(type, K);
    if (!rb) return PM355_E_UNSUPPORTED;
    if (t->d) { (void) hipFree(t->d); t->d = nullptr; }
    t->type = type; t->K = K; t->N = N; t->bytes = rb * (size_t) N;
    const bool mat = (kind >= PM355_T_WQ && kind <= PM355_T_WO) || (kind >= PM355_T_FFN_GATE && kind <= PM355_T_FFN_DOWN) ||
                     kind == PM355_T_TOK_EMBD || kind == PM355_T_OUTPUT;
    const size_t hbm = (mat ? pm_weight_row_stride(type, K) : rb) * (size_t) N;
    if (hipMalloc(&t->d, hbm + 256) != hipSuccess) return PM355_E_NOMEM;        // +256: tail slack for 16-B vector reads
    g_last_hbm = hbm + 256;
    return 0;
}

bool is_matrix(int kind) {
    return (kind >= PM355_T_WQ && kind <= PM355_T_WO) || (kind >= PM355_T_FFN_GATE && kind <= PM355_T_FFN_DOWN) ||
           kind == PM355_T_TOK_EMBD || kind == PM355_T_OUTPUT;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// synthetic data in HBM
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pm_hash(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t) x;
}
// every dword random; then the fp16 scale fields are overwritten with finite values of magnitude ~scale
__global__ void fill_random_blocks_kernel(int type, uint32_t * dst, long n_dwords, long nb_total, long nb_row, uint64_t seed, float scale) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_dwords) dst[i] = pm_hash(seed * 0x9E3779B97F4A7C15ULL + (uint64_t) i);
}
__global__ void fix_scales_kernel(int type, uint8_t * dst, long nb_total, long nb_row, long row_bytes, uint64_t seed, float scale) {
    const long b = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb_total) return;
    const long row = b / nb_row, bi = b % nb_row;
    uint8_t * r = dst + row * row_bytes;
    const float u = 0.5f + (float) (pm_hash(seed ^ (0xABCDEF12345ULL + (uint64_t) b)) & 0xFFFF) / 65536.0f;   // [0.5, 1.5)
    if (type == PM_Q4_K || type == PM_Q5_K) {                         // HBM layout: Q4_K header stream at 128 nb, Q5_K native blocks
        uint16_t * h = (uint16_t *) (type == PM_Q4_K ? r + nb_row * 128 + bi * 16 : r + bi * PM_BS_Q5_K);
        const float qmax = type == PM_Q4_K ? 15.f : 31.f;
        h[0] = f2h(u * scale / (qmax * 32.f));
        h[1] = f2h(u * scale / 64.f);
    } else if (type == PM_Q6_K) {
        *(uint16_t *) (r + pm_q6k_d_off((uint32_t) nb_row, (uint32_t) bi)) = f2h(u * scale / 2048.f);
    } else if (type == PM_Q8_0) {
        ((uint16_t *) (r + nb_row * 32))[bi] = f2h(u * scale / 64.f);
    }
}
__global__ void fill_random_f32_kernel(float * dst, long n, uint64_t seed, float mean, float amp) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = mean + amp * ((float) (pm_hash(seed + (uint64_t) i) & 0xFFFFFF) / 8388608.0f - 1.0f);
}
__global__ void fill_random_f16_kernel(uint16_t * dst, long n, uint64_t seed, float amp) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f2h(amp * ((float) (pm_hash(seed + (uint64_t) i) & 0xFFFFFF) / 8388608.0f - 1.0f));
}

void pm_launch_fill_random_blocks(int type, void * dst, int64_t K, int64_t nrows, uint64_t seed, float scale, hipStream_t st) {
    // rows in the HBM layout: the row STRIDE (== ggml_row_size except where the trailing fp16 stream of a row-SoA layout is padded to 16 bytes,
    // e.g. Q8_0 at K = 29568). Filling with the unpadded row size put every scale of such a matrix in the wrong place (NaN activations in the
    // synthetic Qwen2.5-72B model of rounds 1-2: its timing was unaffected, its values were garbage).
    const size_t rb = (type == PM_F32 || type == PM_F16) ? pm_weight_row_bytes(type, K) : pm_weight_row_stride(type, K);
    const long n = (long) (rb * nrows);
    if (type == PM_F32) { hipLaunchKernelGGL(fill_random_f32_kernel, dim3((unsigned) ((n / 4 + 255) / 256)), dim3(256), 0, st, (float *) dst, n / 4, seed, 0.0f, scale); return; }
    if (type == PM_F16) { hipLaunchKernelGGL(fill_random_f16_kernel, dim3((unsigned) ((n / 2 + 255) / 256)), dim3(256), 0, st, (uint16_t *) dst, n / 2, seed, scale); return; }
    const long nd = (n + 3) / 4;
    const long nb_row = type == PM_Q8_0 ? K / 32 : K / 256;
    hipLaunchKernelGGL(fill_random_blocks_kernel, dim3((unsigned) ((nd + 255) / 256)), dim3(256), 0, st, type, (uint32_t *) dst, nd, nb_row * nrows, nb_row, seed, scale);
    hipLaunchKernelGGL(fix_scales_kernel, dim3((unsigned) ((nb_row * nrows + 255) / 256)), dim3(256), 0, st, type, (uint8_t *) dst, nb_row * nrows, nb_row, (long) rb, seed, scale);
}
void pm_launch_fill_random_f32(float * dst, int64_t n, uint64_t seed, float mean, float amp, hipStream_t st) {
    hipLaunchKernelGGL(fill_random_f32_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, dst, (long) n, seed, mean, amp);
}

// ------------------------------------------------------------------------------------------------
// the window's kernel sequence
// ------------------------------------------------------------------------------------------------
namespace {

// ---- window streaming ------------------------------------------------------------------------------------------------------------
// enqueue the H2D copies of layer `il` into its slot on the copy stream (after the slot was released by its previous user)
void stream_prefetch(pm355_model * m, int il, int slot) {
    const int n = m->hi - m->lo;
    Slot & S = m->slots[slot];
    Layer & L = m->layers[il - m->lo];
    (void) hipStreamWaitEvent(m->copy_stream, S.free_, 0);
    for (int k = 0; k < 12; ++k) if (L.host[k]) {
        (void) hipMemcpyAsync(S.d[k], L.host[k], L.hbm[k], hipMemcpyHostToDevice, m->copy_stream);
        m->streamed_bytes += L.hbm[k];
    }
    (void) hipEventRecord(S.ready, m->copy_stream);
    S.layer = il;
    (void) n;
}
// the layer as the kernels see it: resident tensors, or (streaming) the slot's copies once the copy stream has delivered them
Layer layer_acquire(pm355_model * m, int il, hipStream_t st) {
    Layer L = m->layers[il - m->lo];
    if (!m->n_slots) return L;
    // slots are dealt out by a running step counter, not by layer index: with n % n_slots != 0 a layer-indexed slot made the layer
    // that wraps round to the next token land on a slot that still held an unconsumed prefetched layer (redundant, un-overlapped copies)
    const int slot = (int) (m->stream_step % (uint64_t) m->n_slots);
    Slot & S = m->slots[slot];
    if (S.layer != il) stream_prefetch(m, il, slot);                    // (only after an interrupted pass; normally already in flight)
    (void) hipStreamWaitEvent(st, S.ready, 0);
    for (int k = 0; k < 12; ++k) if (L.host[k]) L.t[k].d = S.d[k];
    return L;
}
// the layer's last kernel has been enqueued: release its slot and start fetching the layer that uses the slot next (wrapping around
// to the first layers of the next token)
void layer_release(pm355_model * m, int il, hipStream_t st) {
    if (!m->n_slots) return;
    const int n = m->hi - m->lo;
    const int slot = (int) (m->stream_step % (uint64_t) m->n_slots);
    Slot & S = m->slots[slot];
    (void) hipEventRecord(S.free_, st);
    // the layer that is executed n_slots steps from now takes the slot that has just been released
    if (n > m->n_slots) stream_prefetch(m, m->lo + (il - m->lo + m->n_slots) % n, slot);
    ++m->stream_step;
}

// quantize `src` [T][K] into the activation format(s) the given weights need; returns pointers
struct ActQ { const void * k = nullptr; const void * z = nullptr; bool tab = false; };   // tab: the small-batch mat-mul's activation tables were written too
const int MMQ_MIN_TOKENS = 3;                           // (3 columns cost the multi-column mat-vec two passes: 57 vs 39 us on the ffn shape)
// largest batch on the integer matrix cores (mmq_i8.hip: the CPU's Q8_K arithmetic, one weight pass per 32 tokens); beyond it the F16 GEMMs (mmq_pf.hip).
// Round 6: 32 (was 64) - with 64-token tiles the prompt GEMM takes 33..64 tokens through a 70B layer in 304-314 us against 378-447 for two passes of the
// integer kernel (profiles/r06_small_batch.txt; round 5's 256-token tiles: 3 x slower, profiles/r05_small_batch_crossover.txt). PM355_MMQ_MAX_TOKENS = 16 .. 64
// moves the crossover (the F16 path is the prompt's parity tier: activations not re-quantized)
const int MMQ_MAX_TOKENS = [] { const char * e = getenv("PM355_MMQ_MAX_TOKENS"); const int v = e ? atoi(e) : 0; return v >= 16 && v <= 64 ? v : 32; }();
// small batches from this size on attend through the prompt path's matrix-core kernel (attn_prefill.hip: one workgroup per head, the same rounding points);
// below it one workgroup per (head, token) (attn_cached.hip). 70B layer at 16 / 24 / 32 tokens: 190.1 / 219.1 / 229.9 -> 188.9 / 214.0 / 217.0 us
const int SMALL_ATTN_MFMA_MIN = [] { const char * e = getenv("PM355_SMALL_ATTN_MFMA_MIN"); return e ? atoi(e) : 16; }();
const int MMQ_MULTI_MIN_TOKENS = 2;       // the fused wq | wk | wv and ffn_gate | ffn_up launches already win at 2 tokens (3 / 4 mat-vec launches otherwise)
// the table output of the Q8_K quantizers, when this batch size takes the small-batch mat-mul (mmq_i8.hip)
pm_q8k_tables mmq_tables(const pm355_model * m, int K, int T, hipStream_t st) {
    pm_q8k_tables tb;
    const bool on = T >= (m->no_multi ? MMQ_MIN_TOKENS : MMQ_MULTI_MIN_TOKENS) && T <= MMQ_MAX_TOKENS && !m->no_mmq;
    if (on && pm_mmq_i8_tables(K, st, &tb) != 0) tb = pm_q8k_tables();
    return on ? tb : pm_q8k_tables();
}
ActQ quantize_for(pm355_model * m, const float * src, int K, int T, const Tensor * const * ws, int nw, hipStream_t st) {
    ActQ a; bool need_k = false, need_0 = false;
    for (int i = 0; i < nw; ++i) if (ws[i] && ws[i]->d) { if (ws[i]->type == PM_Q8_0) need_0 = true; else need_k = true; }
    if (need_k) { const pm_q8k_tables tb = mmq_tables(m, K, T, st); pm_launch_quantize_q8k(src, m->aq_k, K, T, st, tb); a.k = m->aq_k; a.tab = tb.base != nullptr; }
    if (need_0) { pm_launch_quantize_q80(src, m->aq_0, K, T, st); a.z = m->aq_0; }
    return a;
}

ActQ norm_quantize_for(pm355_model * m, const float * src, const float * w, int K, int T, const Tensor * const * ws, int nw, hipStream_t st) {
    bool need_0 = false;
    for (int i = 0; i < nw; ++i) if (ws[i] && ws[i]->d && ws[i]->type == PM_Q8_0) need_0 = true;
    if (!need_0) {
        const pm_q8k_tables tb = mmq_tables(m, K, T, st);
        pm_launch_rmsnorm_q8k(src, w, nullptr, m->aq_k, K, T, m->hp.rms_eps, st, nullptr, tb);
        ActQ a; a.k = m->aq_k; a.tab = tb.base != nullptr; return a;
    }
    pm_launch_rmsnorm_q8k(src, w, m->xn, nullptr, K, T, m->hp.rms_eps, st);
    return quantize_for(m, m->xn, K, T, ws, nw, st);
}

int gemv(const Tensor & w, const Tensor * w2, const ActQ & a, int T, float * y, const float * bias, const float * resid, hipStream_t st) {
    pm_gemv_args g = {};
    g.type = w.type; g.K = (int) w.K; g.N = (int) w.N; g.W = w.d; g.W2 = w2 ? w2->d : nullptr;
    g.xq = w.type == PM_Q8_0 ? a.z : a.k; g.ncols = T; g.y = y; g.y_stride = (size_t) w.N; g.bias = bias; g.resid = resid;
    return pm_launch_gemv(g, st);
}

// 3..MMQ_MAX_TOKENS (32; on request 64) tokens: one pass over the weights per 32 tokens on the integer matrix cores (mmq_i8.hip) where the type / shape is served, else the mat-vec
// (one to three passes per 8 columns). `prepped`: the kernel's activation tables already describe THIS activation set (set by the first
// served call, cleared by the caller whenever the activations change)
int matmul_small(pm355_model * m, const Tensor & w, const ActQ & a, int T, float * y, const float * bias, const float * resid, bool & prepped, hipStream_t st) {
    // (3 / 4 tokens on the 4-slot multi-column mat-vec - two rows per activation fetch: ffn_down Q6_K 46 us - measured against this kernel once it had
    //  its operand-ordered activation table (54 us without the prologue launch): 177 vs 172 us per layer, so the matrix-core kernel keeps them)
    const bool q80 = w.type == PM_Q8_0;                  // Q8_0 weights take Q8_0 activations; the kernel builds their tables per call
    const void * xa = q80 ? a.z : a.k;
    if (T >= MMQ_MIN_TOKENS && T <= MMQ_MAX_TOKENS && !m->no_mmq && xa && pm_mmq_i8_check(w.type, (int) w.K, (int) w.N, T) == 0) {
        const int rc = pm_launch_mmq_i8(w.type, w.d, xa, nullptr, y, (int) w.K, (int) w.N, T, bias, resid, q80 ? 0 : ((prepped || a.tab) ? 1 : 0), st);
        if (rc == 0) { if (!q80) prepped = true; return 0; }
    }
    return gemv(w, nullptr, a, T, y, bias, resid, st);
}

// single-token fused GEMV launch helper: jobs share the f32 activation `xf` (rms_norm'ed with norm_w when given)
// ss_out: this launch leaves its per-workgroup sum-of-squares partials there and *n_ss_out receives their number (0: not served - the consumer
// then reduces the row itself); ss_in / n_ss_in: partials of xf's sum of squares left by the producing launch
int gemv_f32(pm355_model * m, const Tensor * const * ws, const Tensor * const * w2s, float * const * ys,
             const float * const * biases, const float * const * resids, int nj,
             const float * xf, const float * norm_w, hipStream_t st, const pm_qkv_epi * epi = nullptr,
             double * ss_out = nullptr, int * n_ss_out = nullptr, const double * ss_in = nullptr, int n_ss_in = 0) {
    pm_gemv_fused f = {};
    f.K = (int) ws[0]->K; f.njobs = nj; f.xf = xf; f.norm_w = norm_w; f.eps = m->hp.rms_eps; f.epi = epi;
    for (int j = 0; j < nj; ++j) {
        f.job[j].type = ws[j]->type; f.job[j].N = (int) ws[j]->N; f.job[j].W = ws[j]->d; f.job[j].W2 = w2s ? (w2s[j] ? w2s[j]->d : nullptr) : nullptr;
        f.job[j].y = ys[j]; f.job[j].bias = biases ? biases[j] : nullptr; f.job[j].resid = resids ? resids[j] : nullptr;
    }
    if (n_ss_out) *n_ss_out = 0;
    if (ss_out && n_ss_out) {
        const int g = pm_gemv_fused_grid(f);
        if (g > 0 && g <= 256) { f.ss_out = ss_out; *n_ss_out = g; }
    }
    if (ss_in && n_ss_in > 0 && norm_w) { f.ss_in = ss_in; f.n_ss = n_ss_in; }
    return pm_launch_gemv_fused(f, st);
}

// result_norm + lm_head (+ greedy argmax) on ONE hidden row (build_llama's last sub-graph, src/llama.cpp:11191-11215)
int run_head(pm355_model * m, const float * x_row, float * d_logits, int32_t * d_argmax, hipStream_t st, const double * ss_in = nullptr, int n_ss_in = 0) {
    if (!m->output.d || !m->out_norm.d) return seterr(m, PM355_E_UNSUPPORTED, "head: output / output_norm missing");
    const Tensor * ow[1] = {&m->output};
    float * lg = d_logits ? d_logits : m->logits;
    if (!m->no_fuse) {
        float * y[1] = {lg};
        if (gemv_f32(m, ow, nullptr, y, nullptr, nullptr, 1, x_row, (const float *) m->out_norm.d, st, nullptr, nullptr, nullptr, ss_in, n_ss_in))
            return seterr(m, PM355_E_UNSUPPORTED, "head: fused lm_head gemv");
    } else {
        ActQ a = norm_quantize_for(m, x_row, (const float *) m->out_norm.d, m->hp.n_embd, 1, ow, 1, st);
        if (gemv(m->output, nullptr, a, 1, lg, nullptr, nullptr, st)) return seterr(m, PM355_E_UNSUPPORTED, "head: lm_head gemv");
    }
    if (d_argmax) pm_launch_argmax(lg, m->hp.n_vocab, d_argmax, nullptr, st);
    return 0;
}

// The single-token layer stack as ONE persistent launch: the same launches run_layers_fused issues, appended as phases (decode_engine.hip).
// nullptr: this window is not served (types, shapes, streaming, long-context regime) - the caller takes the five-launch path.
// may_build = false (the stream is being captured: no allocation, no synchronous copy): only a plan that exists already is returned.
#if PM_EXPERIMENTS
pm355_model::EnginePlan * engine_plan_for(pm355_model * m, const float * cur, float * d_x_out, bool may_build = true) {
    for (auto & e : m->eng_plans) if (e.in == cur && e.out == d_x_out) return &e;
    if (m->eng_refused || !may_build) return nullptr;
    const float * in0 = cur;
    const pm355_hparams & hp = m->hp;
    const int H = hp.n_head, Hkv = hp.n_head_kv, dh = hp.head_dim;
    const size_t E = hp.n_embd, Eq = (size_t) dh * H, F = hp.n_ff;
    const float kq_scale = 1.0f / sqrtf((float) dh);
    const int nl = m->hi - m->lo;
    // per layer: q[Eq] | att[Eq] | h[F] | x_mid[E] | x_next[E] floats, then ss_wo[256] | ss_dn[256] doubles (the first layer's attn_norm reads the
    // one-partial sum the launch in front of the engine leaves in ss_in0)
    const size_t per_f = 2 * Eq + F + 2 * E, per_bytes = ((per_f * 4 + 255) & ~(size_t) 255) + 2 * 256 * sizeof(double);
    float * act = nullptr;
    if (hipMalloc((void **) &act, per_bytes * (size_t) nl + 256 * sizeof(double) + 256) != hipSuccess) { (void) hipGetLastError(); m->eng_refused = true; return nullptr; }
    pm_eng_plan * pl = pm_eng_plan_new();
    auto refuse = [&](const char * why, int rc) -> pm355_model::EnginePlan * {
        if (getenv("PM355_ENGINE_VERBOSE")) fprintf(stderr, "prima_mi355 engine: not served (%s, rc %d) - five launches per layer\n", why, rc);
        (void) hipGetLastError();
        pm_eng_plan_free(pl); (void) hipFree(act); m->eng_refused = true; return nullptr;
    };
    const int max_keys = (m->split_scratch && m->split_min + 8 < hp.n_ctx) ? m->split_min + 8 : hp.n_ctx;
    double * ss_in0 = (double *) ((char *) act + per_bytes * (size_t) nl);
    const double * ss_prev = ss_in0; int n_prev = 1;          // partials of the sum of squares of `cur`
    auto job = [](pm_gemv_fused & f, int j, const Tensor & w, const Tensor * w2, float * y, const float * bias, const float * resid) {
        f.job[j].type = w.type; f.job[j].N = (int) w.N; f.job[j].W = w.d; f.job[j].W2 = w2 ? w2->d : nullptr; f.job[j].y = y; f.job[j].bias = bias; f.job[j].resid = resid;
    };
    for (int il = m->lo; il < m->hi; ++il) {
        Layer & L = m->layers[il - m->lo];
        char * base = (char *) act + per_bytes * (size_t) (il - m->lo);
        float * q = (float *) base, * att = q + Eq, * hbuf = att + Eq, * x_mid = hbuf + F, * x_own = x_mid + E;
        double * ss_wo = (double *) (base + ((per_f * 4 + 255) & ~(size_t) 255)), * ss_dn = ss_wo + 256;
        float * x_next = (il == m->hi - 1 && d_x_out) ? d_x_out : x_own;
        const long kvs = m->n_seq > 1 ? (long) hp.n_ctx * Hkv * dh : 0;
        int rc, n_wo, n_dn;
        {   // wq | wk | wv + rope + KV store
            pm_gemv_fused f = {};
            f.K = hp.n_embd; f.njobs = 3; f.xf = cur; f.norm_w = (const float *) L.t[PM355_T_ATTN_NORM].d; f.eps = hp.rms_eps;
            job(f, 0, L.t[PM355_T_WQ], nullptr, q, (const float *) L.t[PM355_T_BQ].d, nullptr);
            job(f, 1, L.t[PM355_T_WK], nullptr, m->k, (const float *) L.t[PM355_T_BK].d, nullptr);
            job(f, 2, L.t[PM355_T_WV], nullptr, m->v, (const float *) L.t[PM355_T_BV].d, nullptr);
            const pm_qkv_epi qe = {m->rope_tab, m->d_pos, m->d_ctl, nullptr, kvs, L.kc, L.vc, Hkv, dh, hp.n_ctx, m->rope.n_dims, 0, (m->rope.mode & 2) ? 1 : 0};
            f.epi = &qe; f.ss_in = ss_prev; f.n_ss = n_prev;
            if ((rc = pm_eng_plan_add_matvec(pl, f))) return refuse("wq | wk | wv", rc);
        }
        if ((rc = pm_eng_plan_add_attention(pl, q, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, att, H, Hkv, dh, hp.n_ctx, kq_scale, max_keys))) return refuse("attention", rc);
        {   // wo + residual, leaves the partials of ffn_norm
            pm_gemv_fused f = {};
            f.K = (int) L.t[PM355_T_WO].K; f.njobs = 1; f.xf = att; f.eps = hp.rms_eps;
            job(f, 0, L.t[PM355_T_WO], nullptr, x_mid, nullptr, cur);
            n_wo = pm_gemv_fused_grid(f);
            if (n_wo < 1 || n_wo > 256) return refuse("wo grid", n_wo);
            f.ss_out = ss_wo;
            if ((rc = pm_eng_plan_add_matvec(pl, f))) return refuse("wo", rc);
        }
        if (L.t[PM355_T_FFN_GATE].type != L.t[PM355_T_FFN_UP].type) return refuse("ffn_gate / ffn_up types", -1);
        {   // ffn_gate | ffn_up + silu * mul
            pm_gemv_fused f = {};
            f.K = hp.n_embd; f.njobs = 1; f.xf = x_mid; f.norm_w = (const float *) L.t[PM355_T_FFN_NORM].d; f.eps = hp.rms_eps;
            job(f, 0, L.t[PM355_T_FFN_GATE], &L.t[PM355_T_FFN_UP], hbuf, nullptr, nullptr);
            f.ss_in = ss_wo; f.n_ss = n_wo;
            if ((rc = pm_eng_plan_add_matvec(pl, f))) return refuse("ffn_gate | ffn_up", rc);
        }
        {   // ffn_down + residual, leaves the partials of the next attn_norm / output_norm
            pm_gemv_fused f = {};
            f.K = (int) L.t[PM355_T_FFN_DOWN].K; f.njobs = 1; f.xf = hbuf; f.eps = hp.rms_eps;
            job(f, 0, L.t[PM355_T_FFN_DOWN], nullptr, x_next, nullptr, x_mid);
            n_dn = pm_gemv_fused_grid(f);
            if (n_dn < 1 || n_dn > 256) return refuse("ffn_down grid", n_dn);
            f.ss_out = ss_dn;
            if ((rc = pm_eng_plan_add_matvec(pl, f))) return refuse("ffn_down", rc);
        }
        cur = x_next; ss_prev = ss_dn; n_prev = n_dn;
    }
    const int rc = pm_eng_plan_finish(pl);
    if (rc) return refuse("finish", rc);
    m->eng_plans.push_back({in0, d_x_out, pl, n_prev, ss_prev, cur, act, ss_in0});
    return &m->eng_plans.back();
}

#else
pm355_model::EnginePlan * engine_plan_for(pm355_model *, const float *, float *, bool = true) { return nullptr; }      // (experiments library only)
#endif
bool engine_eligible(const pm355_model * m) {
    return m->use_engine && !m->no_fuse && !m->long_ctx && !m->n_slots && m->ss && m->qkv_epi && m->rope_tab && (m->rope.mode == 0 || m->rope.mode == 2) && m->hi > m->lo;
}

// which single-token attention path the current sequence takes: 0 = one workgroup per head; else the cells the long-context grid is
// sized for (power-of-two bucket >= position + 1, capped at n_ctx) - also the key of the captured step graph
int attn_regime(const pm355_model * m) {
    const int pos = m->h_pos[m->h_seq];
    if (!m->split_scratch || pos < m->split_min) return 0;
    long b = 1024;
    while (b < (long) pos + 2) b *= 2;
    return (int) (b > m->hp.n_ctx ? m->hp.n_ctx : b);
}

// the single-token layer sequence (5 launches per layer)
// *n_ss_end: number of partials the LAST layer's ffn_down left in m->ss + 256 for the sum of squares of the window's output row (0: none)
int run_layers_fused(pm355_model * m, const float * cur, float * d_x_out, const float ** cur_out, hipStream_t st, int * n_ss_end = nullptr) {
    const pm355_hparams & hp = m->hp;
    const int H = hp.n_head, Hkv = hp.n_head_kv, dh = hp.head_dim;
    const float kq_scale = 1.0f / sqrtf((float) dh);
    float * bufs[2] = {m->x, m->x1};
    // rope + KV store in the QKV epilogue: NORM-mode rope, one-workgroup-per-head attention regime
    // (NEOX rope - build_qwen2 - where every workgroup's slice of wq / wk holds both halves of its rotation pairs: pm_launch_gemv_fused tells)
    bool epi = m->qkv_epi && m->rope_tab && (m->rope.mode == 0 || m->rope.mode == 2) &&
               (!m->long_ctx || (m->use_flash && m->attn_mfma && H / Hkv <= 8 && pm_attn_flash_cached_ok(H, Hkv, dh, hp.n_ctx) == 0));
    if (epi) pm_launch_rope_table(m->rope, m->d_pos, m->d_ctl, (const float *) m->rope_freqs.d, m->rope_tab, st);
    bool tail = epi && !m->long_ctx && m->attn_tail && m->att_tk && Hkv <= 64;
    double * ss_wo = (m->use_ss && m->ss) ? m->ss : nullptr, * ss_dn = ss_wo ? m->ss + 256 : nullptr;
    int n_wo = 0, n_dn = 0;                           // partials the previous wo / ffn_down launch left (0: the consumer reduces the row itself)
    for (int il = m->lo; il < m->hi; ++il) {
        Layer Lv = layer_acquire(m, il, st); Layer & L = Lv;
        float * q = m->q, * k = m->k, * v = m->v, * att = m->att, * hbuf = m->h;
        float * x_mid = (cur == bufs[0]) ? bufs[1] : bufs[0];
        float * x_nxt = (x_mid == bufs[0]) ? bufs[1] : bufs[0];
        {
            const Tensor * ws[3] = {&L.t[PM355_T_WQ], &L.t[PM355_T_WK], &L.t[PM355_T_WV]};
            float * ys[3] = {q, k, v};
            const float * bs[3] = {(const float *) L.t[PM355_T_BQ].d, (const float *) L.t[PM355_T_BK].d, (const float *) L.t[PM355_T_BV].d};
            const float * nw = (const float *) L.t[PM355_T_ATTN_NORM].d;
            const long kvs_e = m->n_seq > 1 ? (long) hp.n_ctx * Hkv * dh : 0;
            pm_qkv_epi qe = {m->rope_tab, m->d_pos, m->d_ctl, nullptr, kvs_e, L.kc, L.vc, Hkv, dh, hp.n_ctx, m->rope.n_dims, 0, (m->rope.mode & 2) ? 1 : 0};
            bool qkv_done = false;
            if (tail) {
                qe.att_out = att; qe.att_ticket = m->att_tk; qe.att_err = (int *) (m->att_tk + 64); qe.kq_scale = kq_scale; qe.n_head = H;
                qe.att_max_keys = (m->split_scratch && m->split_min + 8 < hp.n_ctx) ? m->split_min + 8 : 0;
                qkv_done = gemv_f32(m, ws, nullptr, ys, bs, nullptr, 3, cur, nw, st, &qe, nullptr, nullptr, ss_dn, n_dn) == 0;
                if (!qkv_done) {
                    if (il != m->lo) return seterr(m, PM355_E_UNSUPPORTED, "decode: attention tail served for some layers only");
                    tail = false; m->attn_tail = false;    // (grid does not split into power-of-two runs per KV head, head_dim: the separate launch)
                    qe.att_out = nullptr;
                }
            }
            if (epi && !qkv_done) {
                qkv_done = gemv_f32(m, ws, nullptr, ys, bs, nullptr, 3, cur, nw, st, &qe, nullptr, nullptr, ss_dn, n_dn) == 0;
                if (!qkv_done) {
                    if (il != m->lo) return seterr(m, PM355_E_UNSUPPORTED, "decode: QKV epilogue served for some layers only");
                    epi = false;                       // (shape / type mix without the epilogue kernel: the whole window takes the round-2 form)
                }
            }
            if (!qkv_done && gemv_f32(m, ws, nullptr, ys, bs, nullptr, 3, cur, nw, st, nullptr, nullptr, nullptr, ss_dn, n_dn)) {
                for (int j = 0; j < 3; ++j)               // type mix without a 3-job kernel: one launch per matrix
                    if (gemv_f32(m, ws + j, nullptr, ys + j, bs + j, nullptr, 1, cur, nw, st)) return seterr(m, PM355_E_UNSUPPORTED, "decode: fused qkv gemv");
            }
        }
        const long kvs = m->n_seq > 1 ? (long) hp.n_ctx * Hkv * dh : 0;     // one slab: the kernels may address the cache before the sequence id arrives
        if (m->long_ctx && epi) {
            // long context, cells complete: scores and P.V on the matrix cores, keys split over workgroups, in-launch merge (attn_flash_mfma.hip)
            if (pm_launch_attn_flash_cached(q, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, att, m->split_scratch, H, Hkv, dh, hp.n_ctx, kq_scale, st,
                                            nullptr, nullptr, 0, m->flash_cells))
                return seterr(m, PM355_E_RANGE, "decode: matrix-core flash-decoding attention unsupported for this shape");
        } else if (m->long_ctx && m->use_flash) {
            // long context: keys split over workgroups, rope + KV store + online softmax + in-launch merge in ONE launch (attn_flash.hip)
            if (pm_launch_attn_flash(q, k, v, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, (const float *) m->rope_freqs.d, att, m->split_scratch,
                                     H, Hkv, dh, hp.n_ctx, kq_scale, m->rope, st, nullptr, nullptr, 0, 0, m->flash_cells))
                return seterr(m, PM355_E_RANGE, "decode: flash-decoding attention unsupported for this shape");
        } else if (m->long_ctx) {
            // (PM355_ATTN_FLASH=0) the three-launch form: scores, probabilities + partial P.V, combine (attn_split.hip)
            if (pm_launch_attn_split(q, k, v, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, (const float *) m->rope_freqs.d, att, m->split_scratch,
                                     H, Hkv, dh, hp.n_ctx, kq_scale, &m->rope, st))
                return seterr(m, PM355_E_RANGE, "decode: split attention unsupported for this shape");
        } else if (epi && tail) {
            // (the attention ran in the tail of the wq | wk | wv launch)
        } else if (epi) {
            if (pm_launch_attn_cached(q, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, att, H, Hkv, dh, hp.n_ctx, kq_scale, st, nullptr, nullptr,
                                      (m->split_scratch && m->split_min + 8 < hp.n_ctx) ? m->split_min + 8 : 0))
                return seterr(m, PM355_E_RANGE, "decode: cached attention unsupported for this head_dim / n_ctx");
        } else if (pm_launch_attn_rope_fused(q, k, v, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, (const float *) m->rope_freqs.d,
                                             att, H, Hkv, dh, hp.n_ctx, kq_scale, m->rope, st, nullptr, nullptr,
                                             // with the split path available this kernel only ever sees < split_min cells: its LDS
                                             // score buffer is sized for that, not for n_ctx (long contexts stay launchable)
                                             (m->split_scratch && m->split_min + 8 < hp.n_ctx) ? m->split_min + 8 : 0))
            return seterr(m, PM355_E_RANGE, "decode: fused attention unsupported for this head_dim / n_ctx");
        {
            const Tensor * w[1] = {&L.t[PM355_T_WO]}; float * y[1] = {x_mid}; const float * r[1] = {cur};
            if (gemv_f32(m, w, nullptr, y, nullptr, r, 1, att, nullptr, st, nullptr, ss_wo, &n_wo)) return seterr(m, PM355_E_UNSUPPORTED, "decode: fused wo gemv");
        }
        if (L.t[PM355_T_FFN_GATE].type != L.t[PM355_T_FFN_UP].type)
            return seterr(m, PM355_E_UNSUPPORTED, "decode: ffn_gate and ffn_up of different types");
        {
            const Tensor * w[1] = {&L.t[PM355_T_FFN_GATE]}; const Tensor * w2[1] = {&L.t[PM355_T_FFN_UP]}; float * y[1] = {hbuf};
            if (gemv_f32(m, w, w2, y, nullptr, nullptr, 1, x_mid, (const float *) L.t[PM355_T_FFN_NORM].d, st, nullptr, nullptr, nullptr, ss_wo, n_wo))
                return seterr(m, PM355_E_UNSUPPORTED, "decode: fused gate/up gemv");
        }
        float * x_next = (il == m->hi - 1 && d_x_out) ? d_x_out : x_nxt;
        {
            const Tensor * w[1] = {&L.t[PM355_T_FFN_DOWN]}; float * y[1] = {x_next}; const float * r[1] = {x_mid};
            if (gemv_f32(m, w, nullptr, y, nullptr, r, 1, hbuf, nullptr, st, nullptr, ss_dn, &n_dn)) return seterr(m, PM355_E_UNSUPPORTED, "decode: fused down gemv");
        }
        layer_release(m, il, st);
        cur = x_next;
    }
    *cur_out = cur;
    if (n_ss_end) *n_ss_end = n_dn;
    return 0;
}

// x_in -> x_out for layers [lo, hi); positions from device memory d_pos (pos of token 0)
int run_window(pm355_model * m, const int32_t * d_tokens, const float * d_x_in, int T, float * d_x_out,
               float * d_logits, int32_t * d_argmax, hipStream_t st) {
    (void) hipGetLastError();
    const pm355_hparams & hp = m->hp;
    const int E = hp.n_embd, H = hp.n_head, Hkv = hp.n_head_kv, dh = hp.head_dim, F = hp.n_ff;
    const int Eq = H * dh;
    const float kq_scale = 1.0f / sqrtf((float) dh);
    const float * cur = d_x_in;
    if (d_tokens) {
        if (!(m->flags & PM355_HAS_EMBD) || !m->tok_embd.d) return seterr(m, PM355_E_UNSUPPORTED, "decode: tokens given but window has no tok_embd");
        pm_launch_embed(m->tok_embd.type, m->tok_embd.d, E, d_tokens, T, m->x, st);
        cur = m->x;
    }
    if (!cur) return seterr(m, PM355_E_SHAPE, "decode: neither tokens nor x_in");
    float * bufs[2] = {m->x, m->x1};
    int n_ss_head = 0; const double * ss_head = m->ss ? m->ss + 256 : nullptr;   // partials of the output row's sum of squares left by the last ffn_down (single-token path)
    // small batches (2..32 tokens) with NORM rope: RoPE + KV store ride in the epilogue of the wq | wk | wv small-batch launch; the tokens' cos / sin tables
    // are written here, once for all layers
    const bool small_epi = T >= MMQ_MULTI_MIN_TOKENS && T <= 32 && T <= MMQ_MAX_TOKENS && !m->no_fuse && !m->no_mmq && !m->no_multi && !m->no_small_epi && m->qkv_epi && m->rope_tab &&
                           m->rope.mode == 0 && m->hi > m->lo;
    if (small_epi) pm_launch_rope_table(m->rope, m->d_pos, m->d_ctl, (const float *) m->rope_freqs.d, m->rope_tab, st, T);
    if (T == 1 && !m->no_fuse) {
        m->flash_cells = attn_regime(m); m->long_ctx = m->flash_cells != 0;
        // ---- single token: every activation transform is fused into a mat-vec prologue / epilogue, 5 launches per layer
        const float * end = nullptr;
        pm355_model::EnginePlan * eng = nullptr;
        if (engine_eligible(m)) {
            hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(st, &cst) == hipSuccess && cst == hipStreamCaptureStatusActive;
            eng = engine_plan_for(m, cur, d_x_out, !capturing);
        }
        if (eng) {
            // ---- ... or, where served, ALL layers as one persistent launch (decode_engine.hip): cos / sin table, the first norm's sum of squares, the engine
            pm_launch_rope_table(m->rope, m->d_pos, m->d_ctl, (const float *) m->rope_freqs.d, m->rope_tab, st);
            pm_launch_sumsq_row(cur, E, eng->ss_in0, st);
#if PM_EXPERIMENTS
            if (pm_eng_plan_launch(eng->plan, st)) return seterr(m, PM355_E_HIP, "decode: engine launch");
#endif
            end = eng->end; n_ss_head = eng->n_ss_end; ss_head = eng->ss_end;
        } else {
            int rc = run_layers_fused(m, cur, d_x_out, &end, st, &n_ss_head);
            if (rc) return rc;
        }
        cur = end;
    } else
    for (int il = m->lo; il < m->hi; ++il) {
        Layer Lv = layer_acquire(m, il, st); Layer & L = Lv;
        const Tensor * qkv[3] = {&L.t[PM355_T_WQ], &L.t[PM355_T_WK], &L.t[PM355_T_WV]};
        // batches up to MMQ_MAX_TOKENS take the small-batch path below unless its per-token attention kernel cannot hold n_ctx scores in LDS
        const bool small_attn_ok = (size_t) (dh + hp.n_ctx) * 4 <= 150 * 1024;
        bool small_ok = !m->no_mmq && small_attn_ok;              // ... and unless one of the layer's large matrices has a type neither small-batch path serves
        for (int k : {PM355_T_WQ, PM355_T_WO, PM355_T_FFN_GATE, PM355_T_FFN_UP, PM355_T_FFN_DOWN}) {
            // Q4_K / Q5_K / Q6_K and (round 4) Q8_0 - Qwen2.5-72B's ffn_down: 29568 % 256 != 0, src/llama.cpp:19547 - on the integer matrix cores
            // (mmq_i8.hip; Q8_0 with Q8_0 activations = ggml_vec_dot_q8_0_q8_0's arithmetic); a Q8_0 shape it does not serve falls back to the multi-column mat-vec
            const bool served = pm_mmq_i8_check(L.t[k].type, (int) L.t[k].K, (int) L.t[k].N, T < 1 ? 1 : (T > MMQ_MAX_TOKENS ? MMQ_MAX_TOKENS : T)) == 0;
            small_ok = small_ok && (served || L.t[k].type == PM_Q8_0);
        }
        // prompts: every large matrix the integer kernel serves (Q4_K / Q6_K, K % 1024 == 0) runs on Q8_K activations with the CPU reference's
        // arithmetic; wv alone may stay on the F16 GEMM (Llama-3-70B Q4_K_M: Q5_K in half the layers, src/llama.cpp:19360-19373)
        bool big_ok = T > MMQ_MAX_TOKENS && !m->no_fuse && !m->no_big && m->tab_big && m->h2;
        for (int k : {PM355_T_WQ, PM355_T_WK, PM355_T_WO, PM355_T_FFN_GATE, PM355_T_FFN_UP, PM355_T_FFN_DOWN})
            big_ok = big_ok && pm_mmq_big_check(L.t[k].type, (int) L.t[k].K, (int) L.t[k].N, T) == 0;
        if (big_ok) {
            auto TB = [&](int K) { pm_q8k_tables tb; tb.base = m->tab_big; tb.tab_bytes = (size_t) (K / 256) * (1024 + 128); tb.nsb = K / 256; return tb; };
            auto GI = [&](const Tensor & w, float * y, const float * bias, const float * resid, hipStream_t s2 = nullptr) {
                return pm_launch_mmq_big(w.type, w.d, m->aq_k, m->tab_big, y, (int) w.K, (int) w.N, T, bias, resid, s2 ? s2 : st);
            };
            const Tensor & wv = L.t[PM355_T_WV];
            const bool wv8 = pm_mmq_big_check(wv.type, (int) wv.K, (int) wv.N, T) == 0;
            pm_launch_rmsnorm_q8k(cur, (const float *) L.t[PM355_T_ATTN_NORM].d, nullptr, m->aq_k, E, T, hp.rms_eps, st, wv8 ? nullptr : m->xn, TB(E));
            int rc = GI(L.t[PM355_T_WQ], m->q, (const float *) L.t[PM355_T_BQ].d, nullptr);
            if (!m->side) {
                if (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->side_a, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&m->side_b, hipEventDisableTiming) != hipSuccess) return seterr(m, PM355_E_HIP, "prefill: side stream");
            }
            (void) hipEventRecord(m->side_a, st);
            (void) hipStreamWaitEvent(m->side, m->side_a, 0);
            rc |= GI(L.t[PM355_T_WK], m->k, (const float *) L.t[PM355_T_BK].d, nullptr);
            rc |= wv8 ? GI(wv, m->v, (const float *) L.t[PM355_T_BV].d, nullptr, m->side)
                      : pm_launch_gemm_q_h(wv.type, wv.d, nullptr, m->xn, m->v, nullptr, (int) wv.K, (int) wv.N, T, (const float *) L.t[PM355_T_BV].d, nullptr, nullptr, 0, m->side);
            (void) hipEventRecord(m->side_b, m->side);
            (void) hipStreamWaitEvent(st, m->side_b, 0);
            if (rc) return seterr(m, PM355_E_UNSUPPORTED, "prefill: qkv mat-mul (integer matrix cores)");
            const long kvs = (long) hp.n_ctx * Hkv * dh;
            pm_launch_rope_kv_store(m->q, m->k, m->v, m->q, nullptr, L.kc, L.vc, m->d_pos, m->d_ctl, kvs,
                                    (const float *) m->rope_freqs.d, T, H, Hkv, dh, hp.n_ctx, m->rope, st);
            if (pm_launch_attn_prefill(m->q, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, m->att, T, H, Hkv, dh, hp.n_ctx, kq_scale, st) &&
                pm_launch_attn_decode(m->q, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, m->att, T, H, Hkv, dh, hp.n_ctx, kq_scale, st))
                return seterr(m, PM355_E_RANGE, "prefill: n_ctx too large for the attention kernel");
            pm_launch_quantize_q8k(m->att, m->aq_k, Eq, T, st, TB(Eq));
            float * x_mid = (cur == bufs[0]) ? bufs[1] : bufs[0];
            if (GI(L.t[PM355_T_WO], x_mid, nullptr, cur)) return seterr(m, PM355_E_UNSUPPORTED, "prefill: wo mat-mul");
            pm_launch_rmsnorm_q8k(x_mid, (const float *) L.t[PM355_T_FFN_NORM].d, nullptr, m->aq_k, E, T, hp.rms_eps, st, nullptr, TB(E));
            if (GI(L.t[PM355_T_FFN_GATE], m->h, nullptr, nullptr) || GI(L.t[PM355_T_FFN_UP], m->h2, nullptr, nullptr))
                return seterr(m, PM355_E_UNSUPPORTED, "prefill: gate/up mat-mul");
            pm_launch_silu_mul_q8k(m->h, m->h2, m->aq_k, F, T, st, TB(F));                   // silu(gate) * up straight into ffn_down's Q8_K rows
            float * x_next = (il == m->hi - 1 && d_x_out) ? d_x_out : ((x_mid == bufs[0]) ? bufs[1] : bufs[0]);
            if (GI(L.t[PM355_T_FFN_DOWN], x_next, nullptr, x_mid)) return seterr(m, PM355_E_UNSUPPORTED, "prefill: down mat-mul");
            layer_release(m, il, st);
            cur = x_next;
            continue;
        }
        if (T > (small_ok ? MMQ_MAX_TOKENS : 15) && !m->no_fuse) {
            // ---- prefill: batched GEMMs on the MFMA matrix cores (mmq.hip), f32 activations
            // F16 plumbing: the producers of GEMM activations write them as F16 (the rounding the GEMM's own conversion pass would apply) into
            // the engine's scratch - xn / att in place of their f32 forms, h = silu(gate) * up into h2: no conversion launches, half the bytes
            auto G = [&](const Tensor & w, const void * xh, float * y, void * yh, const float * bias, const float * resid,
                         const float * silu_gate = nullptr, hipStream_t s2 = nullptr) {
                return pm_launch_gemm_q_h(w.type, w.d, nullptr, xh, y, yh, (int) w.K, (int) w.N, T, bias, resid, silu_gate, 0, s2 ? s2 : st);
            };
            pm_launch_rmsnorm_q8k(cur, (const float *) L.t[PM355_T_ATTN_NORM].d, nullptr, nullptr, E, T, hp.rms_eps, st, m->xn);
            // wq | wk | wv: ONE launch over the shared activations, each matrix with its own quant type (mmq_pf.hip jobs; wk / wv alone - N = n_head_kv * head_dim -
            // filled a fraction of the chip: 212 TFLOP/s as their own launches)
            int rc;
            {
                const Tensor & wq = L.t[PM355_T_WQ], & wk = L.t[PM355_T_WK], & wv = L.t[PM355_T_WV];
                const pm_gemm_pf_job qkvj[3] = {
                    {wq.type, (int) wq.N, wq.d, m->q, nullptr, (const float *) L.t[PM355_T_BQ].d, nullptr, nullptr, 0},
                    {wk.type, (int) wk.N, wk.d, m->k, nullptr, (const float *) L.t[PM355_T_BK].d, nullptr, nullptr, 0},
                    {wv.type, (int) wv.N, wv.d, m->v, nullptr, (const float *) L.t[PM355_T_BV].d, nullptr, nullptr, 0}};
                rc = pm_launch_gemm_q_multi(qkvj, 3, nullptr, m->xn, E, T, st);
            }
            if (rc) return seterr(m, PM355_E_UNSUPPORTED, "prefill: qkv gemm");
            const long kvs = (long) hp.n_ctx * Hkv * dh;
            pm_launch_rope_kv_store(m->q, m->k, m->v, m->q, nullptr, L.kc, L.vc, m->d_pos, m->d_ctl, kvs,
                                    (const float *) m->rope_freqs.d, T, H, Hkv, dh, hp.n_ctx, m->rope, st);
            bool att_h = true;
            if (pm_launch_attn_prefill(m->q, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, nullptr, T, H, Hkv, dh, hp.n_ctx, kq_scale, st, nullptr, 0, 0, 0, 0, m->att)) {
                att_h = false;
                if (pm_launch_attn_decode(m->q, L.kc, L.vc, m->d_pos, m->d_ctl, kvs, m->att, T, H, Hkv, dh, hp.n_ctx, kq_scale, st))
                    return seterr(m, PM355_E_RANGE, "prefill: n_ctx too large for the attention kernel");
            }
            float * x_mid = (cur == bufs[0]) ? bufs[1] : bufs[0];
            if (att_h ? G(L.t[PM355_T_WO], m->att, x_mid, nullptr, nullptr, cur)
                      : pm_launch_gemm_q_ex(L.t[PM355_T_WO].type, L.t[PM355_T_WO].d, m->att, x_mid, (int) L.t[PM355_T_WO].K, (int) L.t[PM355_T_WO].N, T, nullptr, cur, nullptr, 0, st))
                return seterr(m, PM355_E_UNSUPPORTED, "prefill: wo gemm");
            pm_launch_rmsnorm_q8k(x_mid, (const float *) L.t[PM355_T_FFN_NORM].d, nullptr, nullptr, E, T, hp.rms_eps, st, m->xn);
            {
                // ffn_gate | ffn_up as ONE launch of pair tiles: h2 = F16(silu(gate) * up) comes out of the up waves' epilogue, the gate result crosses over inside
                // the workgroup and never reaches HBM (two launches: gate as f32, read back by ffn_up's epilogue - 2 x T x n_ff x 4 bytes per layer)
                const Tensor & wg = L.t[PM355_T_FFN_GATE], & wu = L.t[PM355_T_FFN_UP];
                const pm_gemm_pf_job gu[2] = {{wg.type, (int) wg.N, wg.d, nullptr, nullptr, nullptr, nullptr, nullptr, 0},
                                              {wu.type, (int) wu.N, wu.d, nullptr, m->h2, nullptr, nullptr, nullptr, 0}};
                static const bool no_pair = [] { const char * e = getenv("PM355_GEMM_PAIR"); return e && e[0] == '0'; }();
                if (!no_pair && wg.type == wu.type && wg.N == wu.N && pm_gemm_pf_check(wg.type, E, (int) wg.N, T) == 0 && pm_gemm_pf_enabled())
                    rc = pm_launch_gemm_pf_ex(gu, 2, m->xn, E, T, 1, st);
                else {
                    rc = G(wg, m->xn, m->h, nullptr, nullptr, nullptr);
                    rc |= G(wu, m->xn, nullptr, m->h2, nullptr, nullptr, m->h);                          // h2 = F16(silu(gate) * up), in the epilogue
                }
            }
            if (rc) return seterr(m, PM355_E_UNSUPPORTED, "prefill: gate/up gemm");
            float * x_next = (il == m->hi - 1 && d_x_out) ? d_x_out : ((x_mid == bufs[0]) ? bufs[1] : bufs[0]);
            if (G(L.t[PM355_T_FFN_DOWN], m->h2, x_next, nullptr, nullptr, x_mid)) return seterr(m, PM355_E_UNSUPPORTED, "prefill: down gemm");
            layer_release(m, il, st);
            cur = x_next;
            continue;
        }
        // attn_norm (+weight), quantized for q/k/v in the same pass when only Q8_K is needed
        ActQ a = norm_quantize_for(m, cur, (const float *) L.t[PM355_T_ATTN_NORM].d, E, T, qkv, 3, st);
        int rc = 0;
        bool prepped = false;
        const bool small = T >= MMQ_MULTI_MIN_TOKENS && T <= MMQ_MAX_TOKENS && !m->no_mmq && a.k != nullptr;   // (single launches: from MMQ_MIN_TOKENS, in matmul_small)
        // matrices of one type that share the activations go out as ONE small-batch launch per 32 tokens: wq | wk (| wv), ffn_gate | ffn_up
        auto multi = [&](std::initializer_list<const Tensor *> ws, std::initializer_list<float *> ys, std::initializer_list<const float *> bs, const pm_qkv_epi * epi = nullptr) {
            const void * W[3]; int N[3]; float * Y[3]; const float * B[3]; int n = 0;
            for (const Tensor * w : ws) { W[n] = w->d; N[n] = (int) w->N; ++n; }
            n = 0; for (float * y : ys) Y[n++] = y;
            n = 0; for (const float * b : bs) B[n++] = b;
            const Tensor * w0 = *ws.begin();
            return pm_launch_mmq_i8_multi(w0->type, n, W, N, Y, B, a.k, (int) w0->K, T, (prepped || a.tab) ? 1 : 0, st, epi);
        };
        // RoPE + KV store in the epilogue of the wq | wk | wv launch (NORM rope; the tokens' cos / sin tables were written once for this step): no
        // rope_kv_store launch. PM355_SMALL_ROPE_EPI=0: the separate launch
        const long kv_stride = (long) hp.n_ctx * Hkv * dh;
        const pm_qkv_epi qe = {m->rope_tab, m->d_pos, m->d_ctl, nullptr, kv_stride, L.kc, L.vc, Hkv, dh, hp.n_ctx, m->rope.n_dims, 0, 0};
        const pm_qkv_epi * epi = small_epi ? &qe : nullptr;
        bool rope_done = false;
        const Tensor & wq_ = L.t[PM355_T_WQ], & wk_ = L.t[PM355_T_WK], & wv_ = L.t[PM355_T_WV];
        const float * bq_ = (const float *) L.t[PM355_T_BQ].d, * bk_ = (const float *) L.t[PM355_T_BK].d, * bv_ = (const float *) L.t[PM355_T_BV].d;
        bool qkv_done = false;
        if (small && T <= MMQ_MAX_TOKENS && !m->no_multi && wq_.type == wk_.type) {
            if (wv_.type == wq_.type) {
                if (epi && multi({&wq_, &wk_, &wv_}, {m->q, m->k, m->v}, {bq_, bk_, bv_}, epi) == 0) qkv_done = rope_done = true;
                else if (multi({&wq_, &wk_, &wv_}, {m->q, m->k, m->v}, {bq_, bk_, bv_}) == 0) qkv_done = true;
            } else if (epi && [&] {
                           const void * W[2] = {wq_.d, wk_.d}; const int N[2] = {(int) wq_.N, (int) wk_.N}; float * Y[2] = {m->q, m->k}; const float * B[2] = {bq_, bk_};
                           return wv_.K == wq_.K && pm_launch_mmq_i8_dual(wq_.type, 2, W, N, Y, B, wv_.type, wv_.d, (int) wv_.N, m->v, bv_, a.k, (int) wq_.K, T, (prepped || a.tab) ? 1 : 0, st, epi) == 0;
                       }()) {
                qkv_done = rope_done = true;
            } else if ([&] {        // wv of another K-quant type (Q6_K / Q5_K in the Q4_K_M files): the same grid, its own workgroups (mmq_i8_dual_kernel)
                           const void * W[2] = {wq_.d, wk_.d}; const int N[2] = {(int) wq_.N, (int) wk_.N}; float * Y[2] = {m->q, m->k}; const float * B[2] = {bq_, bk_};
                           return wv_.K == wq_.K && pm_launch_mmq_i8_dual(wq_.type, 2, W, N, Y, B, wv_.type, wv_.d, (int) wv_.N, m->v, bv_, a.k, (int) wq_.K, T, (prepped || a.tab) ? 1 : 0, st) == 0;
                       }()) {
                qkv_done = true;
            } else if (multi({&wq_, &wk_}, {m->q, m->k}, {bq_, bk_}) == 0) {
                prepped = true; qkv_done = true;
                rc |= matmul_small(m, wv_, a, T, m->v, bv_, nullptr, prepped, st);
            }
        }
        if (!qkv_done) {
        rc |= matmul_small(m, L.t[PM355_T_WQ], a, T, m->q, (const float *) L.t[PM355_T_BQ].d, nullptr, prepped, st);
        rc |= matmul_small(m, L.t[PM355_T_WK], a, T, m->k, (const float *) L.t[PM355_T_BK].d, nullptr, prepped, st);
        rc |= matmul_small(m, L.t[PM355_T_WV], a, T, m->v, (const float *) L.t[PM355_T_BV].d, nullptr, prepped, st);
        }
        if (rc) return seterr(m, rc, "decode: qkv gemv");
        bool fused_attn = false;
        if (T == 1 && !m->no_fuse)
            fused_attn = pm_launch_attn_rope_fused(m->q, m->k, m->v, L.kc, L.vc, m->d_pos, m->d_ctl, kv_stride,
                                                   (const float *) m->rope_freqs.d, m->att, H, Hkv, dh, hp.n_ctx, kq_scale, m->rope, st) == 0;
        if (!fused_attn) {
        // q rotated AND rounded to F16, every token's K row / V column stored: the attention of a small batch then is the single-token kernel over
        // cached cells, one workgroup per (head, token) (attn_cached.hip: one barrier up to 64 cells; 11 -> ~4 us per layer at 2..8 tokens)
        if (!rope_done)
        pm_launch_rope_kv_store(m->q, m->k, m->v, m->q, nullptr, L.kc, L.vc, m->d_pos, m->d_ctl, kv_stride,
                                (const float *) m->rope_freqs.d, T, H, Hkv, dh, hp.n_ctx, m->rope, st, 1);
        // (a shape attn_cached refuses - head_dim other than 64 / 128 / 256, scores beyond 150 KiB of LDS - falls back to attn_decode, which rounds q to
        //  F16 itself: the pre-rounded rows give it the same bits as raw ones, tests/test_gpu_ops.py::test_small_batch_attention_fallback_...)
        // (from SMALL_ATTN_MFMA_MIN tokens on: the prompt path's matrix-core kernel - 12 us per layer at 32 tokens against 26 for 2048 small workgroups)
        if (!(T >= SMALL_ATTN_MFMA_MIN && pm_launch_attn_prefill(m->q, L.kc, L.vc, m->d_pos, m->d_ctl, kv_stride, m->att, T, H, Hkv, dh, hp.n_ctx, kq_scale, st) == 0))
        if (pm_launch_attn_cached(m->q, L.kc, L.vc, m->d_pos, m->d_ctl, kv_stride, m->att, H, Hkv, dh, hp.n_ctx, kq_scale, st, nullptr, nullptr, 0, 0, 0, T) &&
            pm_launch_attn_decode(m->q, L.kc, L.vc, m->d_pos, m->d_ctl, kv_stride, m->att, T, H, Hkv, dh, hp.n_ctx, kq_scale, st))
            return seterr(m, PM355_E_RANGE, "decode: n_ctx too large for the decode-attention kernel");
        }
        const Tensor * wo[1] = {&L.t[PM355_T_WO]};
        a = quantize_for(m, m->att, Eq, T, wo, 1, st);
        float * x_mid = (cur == bufs[0]) ? bufs[1] : bufs[0];                 // ffn_inp = wo.att + inpSA
        prepped = false;
        if (matmul_small(m, L.t[PM355_T_WO], a, T, x_mid, nullptr, cur, prepped, st)) return seterr(m, PM355_E_UNSUPPORTED, "decode: wo gemv");
        const Tensor * gu[2] = {&L.t[PM355_T_FFN_GATE], &L.t[PM355_T_FFN_UP]};
        a = norm_quantize_for(m, x_mid, (const float *) L.t[PM355_T_FFN_NORM].d, E, T, gu, 2, st);
        if (L.t[PM355_T_FFN_GATE].type != L.t[PM355_T_FFN_UP].type)
            return seterr(m, PM355_E_UNSUPPORTED, "decode: ffn_gate and ffn_up of different types");
        const Tensor & wg = L.t[PM355_T_FFN_GATE], & wu = L.t[PM355_T_FFN_UP];
        bool gu_done = false;
        // 2 tokens: the pair mat-vec with two column slots - one pass over both matrices, silu(gate) * up in its epilogue (~50 us against 60 for the
        // two-matrix launch of the matrix-core kernel + the product launch)
        const bool pair_cols = T == 2 && !m->no_small_cols && wg.type == wu.type && (wg.type == PM_Q4_K || wg.type == PM_Q6_K);
        if (!pair_cols && small && m->h2 && pm_mmq_i8_check(wg.type, (int) wg.K, (int) wg.N, T) == 0) {
            // gate and up: one weight pass each for all tokens, then silu(gate) * up (the pair mat-vec would take one launch per token)
            prepped = false;
            if (T <= MMQ_MAX_TOKENS && !m->no_multi && wg.type == wu.type && multi({&wg, &wu}, {m->h, m->h2}, {nullptr, nullptr}) == 0) gu_done = true;
            else if (T >= MMQ_MIN_TOKENS) {
                if (matmul_small(m, wg, a, T, m->h, nullptr, nullptr, prepped, st) || matmul_small(m, wu, a, T, m->h2, nullptr, nullptr, prepped, st))
                    return seterr(m, PM355_E_UNSUPPORTED, "decode: gate/up small-batch mat-mul");
                gu_done = true;
            }
        }
        if (!gu_done)
        if (gemv(L.t[PM355_T_FFN_GATE], &L.t[PM355_T_FFN_UP], a, T, m->h, nullptr, nullptr, st)) return seterr(m, PM355_E_UNSUPPORTED, "decode: gate/up gemv");
        const Tensor * dn[1] = {&L.t[PM355_T_FFN_DOWN]};
        if (gu_done && L.t[PM355_T_FFN_DOWN].type != PM_Q8_0) {
            // silu(gate) * up goes straight into the Q8_K rows (and activation tables) ffn_down reads: no f32 product, one launch less
            const pm_q8k_tables tb = mmq_tables(m, F, T, st);
            pm_launch_silu_mul_q8k(m->h, m->h2, m->aq_k, F, T, st, tb);
            a = ActQ(); a.k = m->aq_k; a.tab = tb.base != nullptr;
        } else {
            // Q8_0 ffn_down (n_ff no multiple of 256) on the small-batch mat-mul: silu(gate) * up quantized straight into its activation tables - one launch
            // instead of silu_mul, quantize_q80 and the mat-mul's table prologue
            const Tensor & wd = L.t[PM355_T_FFN_DOWN];
            void * tab = nullptr; size_t qb = 0, db = 0;
            if (gu_done && T >= MMQ_MIN_TOKENS && T <= MMQ_MAX_TOKENS && !m->no_mmq && !m->no_multi && pm_mmq_i8_check(wd.type, (int) wd.K, (int) wd.N, T) == 0 &&
                pm_mmq_i8_q80_tables((int) wd.K, st, &tab, &qb, &db) == 0) {
                pm_launch_silu_mul_q80_tab(m->h, m->h2, tab, qb, db, F, T, st);
                float * x_nx = (il == m->hi - 1 && d_x_out) ? d_x_out : ((x_mid == bufs[0]) ? bufs[1] : bufs[0]);
                if (pm_launch_mmq_i8(wd.type, wd.d, nullptr, nullptr, x_nx, (int) wd.K, (int) wd.N, T, nullptr, x_mid, 1, st)) return seterr(m, PM355_E_UNSUPPORTED, "decode: down mat-mul");
                layer_release(m, il, st);
                cur = x_nx;
                continue;
            }
            if (gu_done) pm_launch_silu_mul(m->h, m->h2, m->h, (long) T * F, st);
            a = quantize_for(m, m->h, F, T, dn, 1, st);
        }
        // `cur` is dead after the wo GEMV consumed it as residual, so the other scratch buffer can be reused
        float * x_next = (il == m->hi - 1 && d_x_out) ? d_x_out : ((x_mid == bufs[0]) ? bufs[1] : bufs[0]);
        prepped = false;
        if (matmul_small(m, L.t[PM355_T_FFN_DOWN], a, T, x_next, nullptr, x_mid, prepped, st)) return seterr(m, PM355_E_UNSUPPORTED, "decode: down gemv");
        layer_release(m, il, st);
        cur = x_next;
    }
    if (d_x_out && cur != d_x_out) (void) hipMemcpyAsync(d_x_out, cur, (size_t) T * E * 4, hipMemcpyDeviceToDevice, st);
    if ((d_logits || d_argmax) && (m->flags & PM355_HAS_HEAD)) {
        int rc = run_head(m, cur + (size_t) (T - 1) * E, d_logits, d_argmax, st, n_ss_head ? ss_head : nullptr, n_ss_head);
        if (rc) return rc;
    }
    return hip_ok() ? 0 : seterr(m, PM355_E_HIP, "decode: kernel launch failed");
}

} // namespace

extern "C" {

pm355_model * pm355_model_new(const pm355_hparams * hp, int lo, int hi, int flags) {
    if (!hp || lo < 0 || hi > hp->n_layer || lo > hi) return nullptr;
    if (hp->head_dim % 8 || hp->n_embd % 256 || hp->n_head % hp->n_head_kv) return nullptr;
    pm355_model * m = new pm355_model();
    m->hp = *hp; m->lo = lo; m->hi = hi; m->flags = flags;
    m->layers.resize(hi - lo);
    m->err[0] = 0;
    m->rope.n_dims = hp->head_dim; m->rope.mode = hp->arch == 1 ? 2 : 0; m->rope.n_ctx_orig = hp->n_ctx_orig;
    m->rope.freq_base = hp->rope_freq_base; m->rope.freq_scale = hp->rope_freq_scale;
    m->rope.ext_factor = 0.0f; m->rope.attn_factor = 1.0f; m->rope.beta_fast = 32.0f; m->rope.beta_slow = 1.0f;
    pm_rope_params(m->rope);
    { const char * e = getenv("PM355_NO_FUSE"); m->no_fuse = e && e[0] == '1'; }
    { const char * e = getenv("PM355_SMALL_ROPE_EPI"); m->no_small_epi = e && e[0] == '0'; }
    { const char * e = getenv("PM355_SMALL_COLS"); m->no_small_cols = e && e[0] == '0'; }   // A/B: 2..4-token steps without the round-4 multi-column choices
    { const char * e = getenv("PM355_NO_MMQ_MULTI"); m->no_multi = e && e[0] == '1'; }   // small batches: one launch per matrix (A/B of the multi-job launches)
    { const char * e = getenv("PM355_NO_MMQ_I8"); m->no_mmq = e && e[0] == '1'; }     // 4..64-token batches: mat-vec columns / F16 GEMM from 16 (the round-1 paths)
    { const char * e = getenv("PM355_QKV_EPI"); m->qkv_epi = !(e && e[0] == '0'); }
    { const char * e = getenv("PM355_PROMPT_I8"); m->no_big = !(e && e[0] == '1'); }
#if PM_EXPERIMENTS          // (round 5's measured-slower forms: libprima_mi355_exp.so only - the product library has no code behind these switches)
    { const char * e = getenv("PM355_SS"); m->use_ss = e && e[0] == '1'; }
    { const char * e = getenv("PM355_ENGINE"); m->use_engine = e && e[0] == '1'; }
    { const char * e = getenv("PM355_ATTN_TAIL"); m->attn_tail = e && e[0] == '1'; }
#endif
    return m;
}

void pm355_model_free(pm355_model * m) {
    if (!m) return;
    (void) hipDeviceSynchronize();
    for (auto & g : m->graphs) (void) hipGraphExecDestroy(g.exec);
#if PM_EXPERIMENTS
    for (auto & e : m->eng_plans) { pm_eng_plan_free(e.plan); if (e.act) (void) hipFree(e.act); }
#endif
    if (m->copy_stream) { (void) hipStreamSynchronize(m->copy_stream); (void) hipStreamDestroy(m->copy_stream); }
    for (auto & S : m->slots) { for (auto p : S.d) if (p) (void) hipFree(p); if (S.ready) (void) hipEventDestroy(S.ready); if (S.free_) (void) hipEventDestroy(S.free_); }
    for (auto & L : m->layers) for (auto p : L.host) if (p) (void) hipHostFree(p);
    for (auto & L : m->layers) { for (auto & t : L.t) if (t.d) (void) hipFree(t.d); if (L.kc) (void) hipFree(L.kc); if (L.vc) (void) hipFree(L.vc); }
    Tensor * g[4] = {&m->tok_embd, &m->out_norm, &m->output, &m->rope_freqs};
    for (auto t : g) if (t->d) (void) hipFree(t->d);
    void * s[] = {m->x, m->x1, m->q, m->k, m->v, m->att, m->h, m->h2, m->logits, m->xn, m->aq_k, m->aq_0, m->d_pos, m->d_tok, m->d_ctl, m->split_scratch, m->rope_tab, m->tab_big, m->ss, m->att_tk};
    for (auto p : s) if (p) (void) hipFree(p);
    pm355_uploader_free(m->up);
    if (m->cap_stream) (void) hipStreamDestroy(m->cap_stream);
    if (m->side) { (void) hipStreamDestroy(m->side); (void) hipEventDestroy(m->side_a); (void) hipEventDestroy(m->side_b); }
    delete m;
}

// streaming mode: a freshly uploaded (and repacked) layer tensor is moved to pinned host memory; its device copy is released
static int stream_park(pm355_model * m, int kind, int layer) {
    if (!m->n_slots || kind >= 12) return 0;
    Layer & L = m->layers[layer - m->lo];
    Tensor & t = L.t[kind];
    (void) pm355_uploader_sync(m->up);
    (void) hipDeviceSynchronize();
    if (L.host[kind]) { (void) hipHostFree(L.host[kind]); L.host[kind] = nullptr; }
    const size_t n = g_last_hbm;
    if (hipHostMalloc(&L.host[kind], n, hipHostMallocDefault) != hipSuccess) return seterr(m, PM355_E_NOMEM, "streaming: pinned host memory");
    if (hipMemcpy(L.host[kind], t.d, n, hipMemcpyDeviceToHost) != hipSuccess) return seterr(m, PM355_E_HIP, "streaming: park tensor");
    (void) hipFree(t.d);
    t.d = nullptr;
    L.hbm[kind] = n;
    if (n > m->slot_bytes[kind]) m->slot_bytes[kind] = n;
    return 0;
}

int pm355_model_set_streaming(pm355_model * m, int n_slots) {
    if (!m || m->finalized) return m ? seterr(m, PM355_E_SHAPE, "set_streaming: call before finalize") : PM355_E_SHAPE;
    for (auto & L : m->layers) for (auto & t : L.t) if (t.d) return seterr(m, PM355_E_SHAPE, "set_streaming: call before the first layer tensor is set");
    if (n_slots < 0 || n_slots > 1024) return seterr(m, PM355_E_RANGE, "set_streaming: n_slots");
    m->n_slots = n_slots;
    return 0;
}
uint64_t pm355_model_streamed_bytes(const pm355_model * m) { return m ? m->streamed_bytes : 0; }

int pm355_model_set_tensor(pm355_model * m, int kind, int layer, int type, const void * host, size_t nbytes) {
    (void) hipGetLastError();
    Tensor * t = tensor_slot(m, kind, layer);
    if (!t) return seterr(m, PM355_E_RANGE, "set_tensor: tensor not in this window");
    int rc = alloc_tensor(m, t, kind, type);
    if (rc) return seterr(m, rc, "set_tensor: alloc");
    if (nbytes != t->bytes) return seterr(m, PM355_E_SHAPE, "set_tensor: byte size does not match type/shape");
    // pinned ring staging: host -> pinned (copier threads) -> device (hipMemcpyAsync) -> repack kernel, all overlapped (upload.hip)
    if (!m->up && !(m->up = pm355_uploader_new(0, -1))) return seterr(m, PM355_E_NOMEM, "set_tensor: uploader (stream / pinned memory)");
    rc = pm355_upload(m->up, type, t->K, host, t->d, nbytes, is_matrix(kind) ? 1 : 0);
    if (rc) return seterr(m, rc, "set_tensor: upload failed");
    if (!hip_ok()) return seterr(m, PM355_E_HIP, "set_tensor: upload failed");
    return stream_park(m, kind, layer);
}

int pm355_model_fill_tensor(pm355_model * m, int kind, int layer, int type, uint64_t seed, float scale) {
    (void) hipGetLastError();
    Tensor * t = tensor_slot(m, kind, layer);
    if (!t) return seterr(m, PM355_E_RANGE, "fill_tensor: tensor not in this window");
    int rc = alloc_tensor(m, t, kind, type);
    if (rc) return seterr(m, rc, "fill_tensor: alloc");
    if (!is_matrix(kind)) {
        if (type != PM_F32) return seterr(m, PM355_E_UNSUPPORTED, "fill_tensor: 1-D tensors are F32");
        const bool norm = kind == PM355_T_ATTN_NORM || kind == PM355_T_FFN_NORM || kind == PM355_T_OUT_NORM;
        pm_launch_fill_random_f32((float *) t->d, t->K, seed, norm ? 1.0f : (kind == PM355_T_ROPE_FREQS ? 4.0f : 0.0f),
                                  norm ? 0.05f : (kind == PM355_T_ROPE_FREQS ? 3.0f : scale), nullptr);
    } else {
        pm_launch_fill_random_blocks(type, t->d, t->K, t->N, seed, scale, nullptr);
    }
    if (!hip_ok()) return seterr(m, PM355_E_HIP, "fill_tensor");
    return stream_park(m, kind, layer);
}

int pm355_model_finalize(pm355_model * m, int max_tokens) { return pm355_model_finalize_seqs(m, max_tokens, 1); }

int pm355_model_finalize_seqs(pm355_model * m, int max_tokens, int n_seq) {
    (void) hipGetLastError();
    if (n_seq < 1 || n_seq > 64) return seterr(m, PM355_E_RANGE, "finalize: n_seq must be 1..64");
    m->n_seq = n_seq;
    (void) pm355_uploader_sync(m->up);
    (void) hipDeviceSynchronize();
    // (a re-finalize re-allocates the KV caches, the rope table and the split scratch that the kept engine plans and captured graphs have baked in)
#if PM_EXPERIMENTS
    for (auto & e : m->eng_plans) { pm_eng_plan_free(e.plan); if (e.act) (void) hipFree(e.act); }
#endif
    m->eng_plans.clear(); m->eng_refused = false;
    for (auto & g : m->graphs) (void) hipGraphExecDestroy(g.exec);
    m->graphs.clear();
    const pm355_hparams & hp = m->hp;
    const size_t E = hp.n_embd, Eq = (size_t) hp.head_dim * hp.n_head, Ekv = (size_t) hp.head_dim * hp.n_head_kv, F = hp.n_ff;
    const size_t T = max_tokens < 1 ? 1 : max_tokens;
    m->max_tokens = (int) T;
    for (int il = m->lo; il < m->hi; ++il) {
        Layer & L = m->layers[il - m->lo];
        for (int kd : {PM355_T_ATTN_NORM, PM355_T_WQ, PM355_T_WK, PM355_T_WV, PM355_T_WO, PM355_T_FFN_NORM, PM355_T_FFN_GATE, PM355_T_FFN_UP, PM355_T_FFN_DOWN})
            if (!L.t[kd].d && !L.host[kd]) return seterr(m, PM355_E_SHAPE, "finalize: a layer tensor is missing");
        const size_t kvb = Ekv * (size_t) hp.n_ctx * 2 * (size_t) n_seq;
        if (hipMalloc(&L.kc, kvb + 256) != hipSuccess || hipMalloc(&L.vc, kvb + 256) != hipSuccess) return seterr(m, PM355_E_NOMEM, "finalize: kv cache");
        (void) hipMemset(L.kc, 0, kvb + 256); (void) hipMemset(L.vc, 0, kvb + 256);
    }
    const size_t maxK = F > Eq ? (F > E ? F : E) : (Eq > E ? Eq : E);
    auto A = [&](void ** p, size_t n) { return hipMalloc(p, n + 256) == hipSuccess; };
    bool ok = A((void **) &m->x, T * E * 4) && A((void **) &m->x1, T * E * 4) && A((void **) &m->xn, T * E * 4) &&
              A((void **) &m->q, T * Eq * 4) && A((void **) &m->k, T * Ekv * 4) && A((void **) &m->v, T * Ekv * 4) &&
              A((void **) &m->att, T * Eq * 4) && A((void **) &m->h, T * F * 4) && (T < MMQ_MULTI_MIN_TOKENS || A((void **) &m->h2, T * F * 4)) && A((void **) &m->logits, (size_t) hp.n_vocab * 4) &&
              A((void **) &m->aq_k, T * pm_q8k_row_bytes((int) ((maxK + 255) / 256 * 256))) &&
              A((void **) &m->aq_0, T * pm_q80_row_bytes((int) ((maxK + 31) / 32 * 32))) &&
              A((void **) &m->d_pos, 64 * 4) && A((void **) &m->d_ctl, 64) && A((void **) &m->d_tok, 64 + T * 4) &&
              A((void **) &m->rope_tab, (size_t) hp.head_dim * 4 * 32) &&      // cos / sin tables of up to 32 tokens (small-batch epilogue)
              A((void **) &m->ss, 2 * 256 * sizeof(double)) &&
              A((void **) &m->att_tk, 65 * 4);
    if (ok) (void) hipMemset(m->att_tk, 0, 65 * 4);
    if (ok && T > MMQ_MAX_TOKENS && !m->no_big) ok = A((void **) &m->tab_big, pm_mmq_big_table_bytes((int) ((maxK + 255) / 256 * 256), (int) T));
    if (!ok) return seterr(m, PM355_E_NOMEM, "finalize: scratch");
    if (hp.n_head / hp.n_head_kv <= 8 && (hp.head_dim == 64 || hp.head_dim == 128) &&
        !A((void **) &m->split_scratch, std::max(pm_attn_split_scratch_floats(hp.n_head, hp.head_dim, hp.n_ctx),
                                                 pm_attn_flash_scratch_floats(hp.n_head, hp.n_head_kv, hp.head_dim, hp.n_ctx)) * 4)) return seterr(m, PM355_E_NOMEM, "finalize: attention scratch");
    if (m->split_scratch) (void) hipMemset(m->split_scratch, 0, std::max(pm_attn_split_scratch_floats(hp.n_head, hp.head_dim, hp.n_ctx),
                                                                         pm_attn_flash_scratch_floats(hp.n_head, hp.n_head_kv, hp.head_dim, hp.n_ctx)) * 4);   // (the flash kernel's tickets start at 0)
    if (m->n_slots) {
        if (m->n_slots > m->hi - m->lo) m->n_slots = m->hi - m->lo;
        if (hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking) != hipSuccess) return seterr(m, PM355_E_HIP, "finalize: copy stream");
        m->slots.resize(m->n_slots);
        for (auto & S : m->slots) {
            for (int k = 0; k < 12; ++k) if (m->slot_bytes[k] && hipMalloc(&S.d[k], m->slot_bytes[k]) != hipSuccess) return seterr(m, PM355_E_NOMEM, "finalize: layer slot");
            if (hipEventCreateWithFlags(&S.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&S.free_, hipEventDisableTiming) != hipSuccess)
                return seterr(m, PM355_E_HIP, "finalize: slot events");
            (void) hipEventRecord(S.free_, nullptr);
        }
        (void) hipDeviceSynchronize();
        for (int i = 0; i < m->n_slots; ++i) stream_prefetch(m, m->lo + i, i);
        m->stream_step = 0;
    }
    m->h_pos.assign(n_seq, 0); m->h_seq = 0;
    { const char * e = getenv("PM355_ATTN_SPLIT_MIN"); if (e && e[0]) m->split_min = atoi(e); }
    { const char * e = getenv("PM355_ATTN_FLASH"); m->use_flash = !(e && e[0] == '0'); }
    { const char * e = getenv("PM355_ATTN_MFMA"); m->attn_mfma = !(e && e[0] == '0'); }
    (void) hipMemset(m->d_pos, 0, 64 * 4);
    { const int32_t ctl[2] = {0, n_seq}; (void) hipMemcpy(m->d_ctl, ctl, 8, hipMemcpyHostToDevice); }
    (void) hipDeviceSynchronize();
    m->finalized = true;
    return hip_ok() ? 0 : seterr(m, PM355_E_HIP, "finalize");
}

size_t pm355_model_weight_bytes(const pm355_model * m) {
    size_t n = 0;
    for (auto & L : m->layers) for (int kd = 0; kd < 12; ++kd) n += L.t[kd].bytes;
    if (m->flags & PM355_HAS_HEAD) n += m->output.bytes + m->out_norm.bytes;
    return n;
}
size_t pm355_model_kv_bytes_per_pos(const pm355_model * m) {
    return (size_t) (m->hi - m->lo) * 2 * (size_t) m->hp.head_dim * m->hp.n_head_kv * 2;
}
int pm355_model_kv_clear(pm355_model * m, pm355_stream_t st) {
    const size_t kvb = (size_t) m->hp.head_dim * m->hp.n_head_kv * (size_t) m->hp.n_ctx * 2 * (size_t) m->n_seq;
    for (auto & L : m->layers) { (void) hipMemsetAsync(L.kc, 0, kvb, (hipStream_t) st); (void) hipMemsetAsync(L.vc, 0, kvb, (hipStream_t) st); }
    return 0;
}
void * pm355_model_kv_ptr(pm355_model * m, int layer, int which) {
    if (layer < m->lo || layer >= m->hi) return nullptr;
    return which ? m->layers[layer - m->lo].vc : m->layers[layer - m->lo].kc;
}
const char * pm355_model_error(pm355_model * m) { return m->err; }
void * pm355_model_tensor_ptr(pm355_model * m, int kind, int layer, int * type_out) {
    Tensor * t = tensor_slot(m, kind, layer);
    if (!t || !t->d) return nullptr;
    if (type_out) *type_out = t->type;
    return t->d;
}

int pm355_model_set_pos(pm355_model * m, int pos, pm355_stream_t st) {
    if (!m->finalized) return seterr(m, PM355_E_SHAPE, "set_pos: model not finalized");
    return pm355_model_set_seq_pos(m, -1, pos, st);
}

// seq >= 0: set that sequence's position (does not change the current sequence); seq == -1: sequence 0 and make it current
int pm355_model_set_seq_pos(pm355_model * m, int seq, int pos, pm355_stream_t st) {
    if (!m->finalized) return seterr(m, PM355_E_SHAPE, "set_seq_pos: model not finalized");
    if (seq >= m->n_seq) return seterr(m, PM355_E_RANGE, "set_seq_pos: seq >= n_seq");
    if (pos < 0 || pos > m->hp.n_ctx) return seterr(m, PM355_E_RANGE, "set_seq_pos: position outside [0, n_ctx]");
    // values travel as kernel arguments: capture-safe, no host buffer lifetime to manage
    if (seq < 0) { pm_launch_set_i32(m->d_ctl, 0, (hipStream_t) st); seq = 0; m->h_seq = 0; }
    pm_launch_set_i32(m->d_pos + seq, pos, (hipStream_t) st);
    m->h_pos[seq] = pos;
    return 0;
}
int pm355_model_set_seq(pm355_model * m, int seq, pm355_stream_t st) {
    if (!m->finalized || seq < 0 || seq >= m->n_seq) return seterr(m, PM355_E_RANGE, "set_seq: bad sequence id");
    pm_launch_set_i32(m->d_ctl, seq, (hipStream_t) st);
    m->h_seq = seq;
    return 0;
}
int pm355_model_head(pm355_model * m, const float * d_x_row, float * d_logits, int32_t * d_argmax, pm355_stream_t st) {
    if (!m->finalized || !(m->flags & PM355_HAS_HEAD)) return seterr(m, PM355_E_UNSUPPORTED, "head: window has no head");
    (void) hipGetLastError();
    int rc = run_head(m, d_x_row, d_logits, d_argmax, (hipStream_t) st);
    if (rc) return rc;
    return hip_ok() ? 0 : seterr(m, PM355_E_HIP, "head: kernel launch failed");
}

int pm355_model_decode(pm355_model * m, const int32_t * d_tokens, const float * d_x_in, int T, int pos0,
                       float * d_x_out, float * d_logits, int32_t * d_argmax, pm355_stream_t st) {
    if (!m->finalized) return seterr(m, PM355_E_SHAPE, "decode: model not finalized");
    if (T < 1 || T > m->max_tokens) return seterr(m, PM355_E_RANGE, "decode: n_tokens exceeds finalize(max_tokens)");
    if (pos0 < 0 || pos0 + T > m->hp.n_ctx) return seterr(m, PM355_E_RANGE, "decode: position outside n_ctx");
    int rc = pm355_model_set_pos(m, pos0, st);
    if (rc) return rc;
    return run_window(m, d_tokens, d_x_in, T, d_x_out, d_logits, d_argmax, (hipStream_t) st);   // device counter and mirror stay at pos0
}

// the same for sequence `seq` of a window finalized with n_seq > 1 (its KV slab, its position counter); leaves `seq` current
int pm355_model_decode_seq(pm355_model * m, int seq, const int32_t * d_tokens, const float * d_x_in, int T, int pos0,
                           float * d_x_out, float * d_logits, int32_t * d_argmax, pm355_stream_t st) {
    if (!m->finalized) return seterr(m, PM355_E_SHAPE, "decode: model not finalized");
    if (seq < 0 || seq >= m->n_seq) return seterr(m, PM355_E_RANGE, "decode_seq: seq >= n_seq");
    if (T < 1 || T > m->max_tokens) return seterr(m, PM355_E_RANGE, "decode: n_tokens exceeds finalize(max_tokens)");
    if (pos0 < 0 || pos0 + T > m->hp.n_ctx) return seterr(m, PM355_E_RANGE, "decode: position outside n_ctx");
    int rc = pm355_model_set_seq_pos(m, seq, pos0, st);
    if (!rc) rc = pm355_model_set_seq(m, seq, st);
    if (rc) return rc;
    return run_window(m, d_tokens, d_x_in, T, d_x_out, d_logits, d_argmax, (hipStream_t) st);
}
int pm355_model_n_embd(const pm355_model * m) { return m ? m->hp.n_embd : 0; }
int pm355_model_n_seq(const pm355_model * m) { return m ? m->n_seq : 0; }
int pm355_small_batch_max_tokens(void) { return MMQ_MAX_TOKENS; }

// head_first != 0 (ring rank 0): d_x_in is the LAST rank's activation; apply the head to it (-> d_argmax / d_logits),
// then embed the token found at d_token (which may be d_argmax itself) and run the window -> d_x_out.
static int step_body(pm355_model * m, const int32_t * d_token, const float * d_x_in, float * d_x_out,
                     float * d_logits, int32_t * d_argmax, int advance, int rotate, int head_first, hipStream_t st) {
    int rc;
    if (head_first) {
        (void) hipGetLastError();
        rc = run_head(m, d_x_in, d_logits, d_argmax, st);
        if (rc) return rc;
        rc = run_window(m, d_token, nullptr, 1, d_x_out, nullptr, nullptr, st);
    } else {
        rc = run_window(m, d_token, d_token ? nullptr : d_x_in, 1, d_x_out, d_logits, d_argmax, st);
    }
    if (rc) return rc;
    if (advance || rotate) pm_launch_advance(m->d_pos, m->d_ctl, advance, rotate, st);
    return 0;
}

int pm355_model_step(pm355_model * m, const int32_t * d_token, const float * d_x_in, float * d_x_out,
                     float * d_logits, int32_t * d_argmax, int advance, int use_graph, pm355_stream_t pst) {
    return pm355_model_step_ex(m, d_token, d_x_in, d_x_out, d_logits, d_argmax, advance, 0, 0, use_graph, pst);
}

int pm355_model_step_ex(pm355_model * m, const int32_t * d_token, const float * d_x_in, float * d_x_out,
                        float * d_logits, int32_t * d_argmax, int advance, int rotate, int head_first, int use_graph,
                        pm355_stream_t pst) {
    if (!m->finalized) return seterr(m, PM355_E_SHAPE, "step: model not finalized");
    if (head_first && (!(m->flags & PM355_HAS_HEAD) || !d_token || !d_x_in)) return seterr(m, PM355_E_SHAPE, "step: head_first needs HEAD, d_token and d_x_in");
    hipStream_t st = (hipStream_t) pst;
    // host mirror of the device-side counters: which attention path this step takes, and the state after it
    if (m->hi > m->lo && m->h_pos[m->h_seq] + 1 > m->hp.n_ctx)
        return seterr(m, PM355_E_RANGE, "step: the sequence is at n_ctx - no KV cell left (llama_decode would fail to find a slot)");
    const int regime = attn_regime(m);
    m->flash_cells = regime; m->long_ctx = regime != 0;
    // the mirror follows the device counters, which only move when the step was really enqueued
    auto commit = [&]() { m->h_pos[m->h_seq] += advance; if (rotate) m->h_seq = (m->h_seq + rotate) % m->n_seq; return 0; };
    if (!use_graph || m->n_slots) {              // (streaming: copies and kernels are ordered with events across two streams, not captured)
        const int rc = step_body(m, d_token, d_x_in, d_x_out, d_logits, d_argmax, advance, rotate, head_first, st);
        return rc ? rc : commit();
    }
    // the engine's phase table is allocated and uploaded outside the capture (keyed on the activation pointers of this step)
    if (engine_eligible(m)) (void) engine_plan_for(m, d_token ? m->x : d_x_in, d_x_out);
    hipGraphExec_t exec = nullptr;
    for (auto & g : m->graphs)
        if (g.in == d_x_in && g.tok == d_token && g.out == d_x_out && g.logits == d_logits && g.argmax == d_argmax &&
            g.adv == advance && g.rot == rotate && g.head == head_first && g.regime == regime) { exec = g.exec; break; }
    if (!exec) {
        hipGraph_t g = nullptr;
        // capture on a private stream (the legacy default stream cannot capture); nothing executes during capture,
        // the instantiated graph is then launched on the caller's stream
        if (!m->cap_stream && hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking) != hipSuccess)
            return seterr(m, PM355_E_HIP, "step: capture stream");
        hipStream_t cs = m->cap_stream;
        if (hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed) != hipSuccess) return seterr(m, PM355_E_HIP, "step: begin capture");
        int rc = step_body(m, d_token, d_x_in, d_x_out, d_logits, d_argmax, advance, rotate, head_first, cs);
        hipError_t e = hipStreamEndCapture(cs, &g);
        if (rc || e != hipSuccess || !g) { if (g) (void) hipGraphDestroy(g); return rc ? rc : seterr(m, PM355_E_HIP, "step: end capture"); }
        e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void) hipGraphDestroy(g);
        if (e != hipSuccess) return seterr(m, PM355_E_HIP, "step: graph instantiate");
        if (m->graphs.size() >= 16) {
            (void) hipDeviceSynchronize();                    // the evicted exec may still be in flight on a caller's stream
            (void) hipGraphExecDestroy(m->graphs.front().exec); m->graphs.erase(m->graphs.begin());
        }
        m->graphs.push_back({d_x_in, d_token, d_x_out, d_logits, d_argmax, advance, rotate, head_first, regime, exec});
    }
    if (hipGraphLaunch(exec, st) != hipSuccess) return seterr(m, PM355_E_HIP, "step: graph launch");
    return commit();
}

// Synchronizes the device; non-zero if the window's launches left an error behind.
int pm355_model_check(pm355_model * m) {
    if (!m) return PM355_E_SHAPE;
    if (hipDeviceSynchronize() != hipSuccess) return seterr(m, PM355_E_HIP, "check: device error");
    if (m->att_tk) {
        int w = 0;
        if (hipMemcpy(&w, m->att_tk + 64, 4, hipMemcpyDeviceToHost) == hipSuccess && w) {
            (void) hipMemset(m->att_tk, 0, 65 * 4);
            snprintf(m->err, sizeof(m->err), "check: a workgroup of the wq | wk | wv launch gave up waiting for its KV-head group (attention tail, code %d)", w);
            return PM355_E_HIP;
        }
    }
#if PM_EXPERIMENTS
    for (auto & e : m->eng_plans) {
        const int w = pm_eng_plan_status(e.plan);
        if (w) { snprintf(m->err, sizeof(m->err), "check: the decode engine's watchdog fired (code %d: 1 loader, 2 consumer barrier, 3 device-wide barrier, 4 item wait, 6 attention barrier)", w); return PM355_E_HIP; }
    }
#endif
    return 0;
}

// d_tokens_io[i] -> d_tokens_io[i+1]: the per-step graph reads its token from m->d_tok[0] and writes argmax to
// m->d_tok[1]; two tiny D2D copies per step move tokens in and out (all on the stream, no host sync).
int pm355_model_generate(pm355_model * m, int32_t * d_io, int pos0, int n_steps, int use_graph, pm355_stream_t pst) {
    if (!m->finalized) return seterr(m, PM355_E_SHAPE, "generate: model not finalized");
    if (!(m->flags & PM355_HAS_EMBD) || !(m->flags & PM355_HAS_HEAD)) return seterr(m, PM355_E_UNSUPPORTED, "generate: needs EMBD and HEAD");
    if (pos0 < 0 || pos0 + n_steps > m->hp.n_ctx) return seterr(m, PM355_E_RANGE, "generate: position outside n_ctx");
    hipStream_t st = (hipStream_t) pst;
    int rc = pm355_model_set_pos(m, pos0, pst);
    if (rc) return rc;
    for (int i = 0; i < n_steps; ++i) {
        (void) hipMemcpyAsync(m->d_tok, d_io + i, 4, hipMemcpyDeviceToDevice, st);
        rc = pm355_model_step(m, m->d_tok, nullptr, nullptr, nullptr, m->d_tok + 1, 1, use_graph, pst);
        if (rc) return rc;
        (void) hipMemcpyAsync(d_io + i + 1, m->d_tok + 1, 4, hipMemcpyDeviceToDevice, st);
    }
    return 0;
}

} // extern "C"
