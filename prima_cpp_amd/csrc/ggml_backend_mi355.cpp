// ggml_backend_mi355.cpp — the ggml-backend plug-in (libggml-mi355.so): registry / device / buffer-type / buffer /
// backend(stream) vtables of the reference's plug-in interface (ggml/src/ggml-backend-impl.h:15-216), implemented on
// the C ABI of libprima_mi355.so. Plain host C++: no HIP in this file. Compiled against the HOST PROJECT's ggml headers
// (prima_cpp_amd/build_plugin.py passes -I<reference>/ggml/include -I<reference>/ggml/src), never copied.
//
// Who calls what (reference call sites): ggml_backend_sched_* (ggml-backend.cpp:1440-2188), ggml_gallocr (alloc sizes),
// llama.cpp directly (tensor_set/get, buffer_clear, dev_memory ...) and tests/test-backend-ops.cpp (:3801-3852).
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#define GGML_BACKEND_MI355_HAVE_GGML
#include "../../include/ggml_backend_mi355.h"
#include "../../include/prima_mi355.h"
#include "ggml_graph_plan.h"

#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>

#define MI355_CHECK(expr) do { int rc_ = (expr); if (rc_ != 0) { fprintf(stderr, "ggml-mi355: %s failed (rc=%d): %s\n", #expr, rc_, pm355_last_error()); GGML_ABORT("ggml-mi355 error"); } } while (0)

namespace {

// GGML_MI355_PLAN_ONLY=1: a DEBUG / TEST mode for machines without a GPU. One pretend device whose "device memory" is host
// memory; graph_compute lowers every graph to its launch plan (ggml_graph_plan.h), reports it, and launches NOTHING, so tensors
// keep whatever bytes they had: results are garbage by design. It exists so that the planner can be exercised under the
// reference's real llama_decode on the CPU test box (tests/test_plugin_plan.py). It is never a compute path.
bool plan_only() { static const bool v = [] { const char * e = getenv("GGML_MI355_PLAN_ONLY"); return e && e[0] == '1'; }(); return v; }
bool env_on(const char * name) { const char * e = getenv(name); return e && e[0] && e[0] != '0'; }

void * dmalloc(size_t n) {
    if (!plan_only()) return pm355_malloc(n);
    void * p = aligned_alloc(256, (n + 255) & ~(size_t) 255);       // device allocations are (at least) 256-byte aligned
    if (p) memset(p, 0, n);
    return p;
}
void   dfree(void * p) { if (plan_only()) free(p); else pm355_free(p); }
int h2d(void * d, const void * s, size_t n, pm355_stream_t st) { if (plan_only()) { memcpy(d, s, n); return 0; } return pm355_memcpy_h2d(d, s, n, st); }
int d2h(void * d, const void * s, size_t n, pm355_stream_t st) { if (plan_only()) { memcpy(d, s, n); return 0; } return pm355_memcpy_d2h(d, s, n, st); }
int d2d(void * d, const void * s, size_t n, pm355_stream_t st) { if (plan_only()) { memmove(d, s, n); return 0; } return pm355_memcpy_d2d(d, s, n, st); }
int dset(void * d, int v, size_t n, pm355_stream_t st) { if (plan_only()) { memset(d, v, n); return 0; } return pm355_memset(d, v, n, st); }
int dsync(pm355_stream_t st) { return plan_only() ? 0 : pm355_sync(st); }
int dsetdev(int d) { return plan_only() ? 0 : pm355_set_device(d); }
int drepack(int type, const void * src, void * dst, int64_t K, int64_t rows, int to_dev, pm355_stream_t st) {
    if (plan_only()) { memcpy(dst, src, (size_t) rows * pm355_row_size(type, K)); return 0; }
    return pm355_repack_rows(type, src, dst, K, rows, to_dev, st);
}

bool is_soa_type(enum ggml_type t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q6_K || t == GGML_TYPE_Q8_0; }   // repack.hip
// ... of weight MATRICES. A 1-D quantized tensor - the quantized KV caches (ggml_new_tensor_1d(type_k, n_embd_k_gqa * kv_size),
// llama_kv_cache_init src/llama.cpp:3548-3551) - keeps ggml's native block order, so the byte offsets of the reference's cache views
// and of state save / restore stay valid (attn_q8.hip)
bool is_soa_tensor(const struct ggml_tensor * t) {
    const struct ggml_tensor * r = t->view_src ? t->view_src : t;
    return is_soa_type(r->type) && r->ne[1] > 1;
}
bool is_gemv_type(enum ggml_type t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K || t == GGML_TYPE_Q8_0; }

struct dev_ctx { int device; std::string name, desc; };
struct buft_ctx { int device; std::string name; };
struct buf_ctx { int device; void * base; size_t size; std::string name; void * stage = nullptr; size_t stage_bytes = 0;
                 pm355_uploader * up = nullptr; bool up_pending = false; };
// Asynchronous weight staging (upload.hip): set_tensor of a large tensor returns when its bytes sit in the pinned ring; the DMA and
// the repack run on the uploader's private stream. Everything that must observe the data drains the pending uploaders first:
// every other buffer operation on that buffer, and graph_compute (all buffers). g_up_pending keeps that check to one load.
// host-side time spent inside the plug-in (GGML_MI355_STATS=1 prints it): where a token's wall time goes besides the kernels
struct host_timers { std::atomic<uint64_t> ns_compute{0}, ns_set{0}, ns_get{0}, ns_sync{0}, n_set{0}, n_get{0}, n_sync{0};
                     // phases of graph_compute: ordering behind uploads, fingerprint, plan lookup / build, KV cell lookup + patch, launch (replay / capture / eager)
                     std::atomic<uint64_t> ns_order{0}, ns_fp{0}, ns_plan{0}, ns_dyn{0}, ns_launch{0}, ns_seta{0}, ns_geta{0}; };
host_timers g_ht;
// steady state: the counters at the first hipGraph replay (everything before it is model load, prompt, planning and capture) and the time of the last one
struct host_snapshot { uint64_t v[12] = {}; std::chrono::steady_clock::time_point t0, t_last; bool taken = false; };
host_snapshot g_snap;
std::atomic<int> g_n_backends{0};        // backends created in this process (the steady-state statistics are meaningful for one)
void host_counters(uint64_t (&v)[12]) {
    const uint64_t x[12] = {g_ht.ns_compute, g_ht.ns_set, g_ht.ns_get, g_ht.ns_sync, g_ht.ns_order, g_ht.ns_fp, g_ht.ns_plan, g_ht.ns_dyn, g_ht.ns_launch, g_ht.ns_seta, g_ht.ns_geta, 0};
    for (int i = 0; i < 12; ++i) v[i] = x[i];
}
struct scoped_ns {
    std::atomic<uint64_t> & acc; std::chrono::steady_clock::time_point t0;
    explicit scoped_ns(std::atomic<uint64_t> & a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~scoped_ns() { acc += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
std::mutex g_up_mu;
std::vector<buf_ctx *> g_up_bufs;
std::atomic<int> g_up_pending{0};
constexpr size_t UPLOAD_ASYNC_MIN = (size_t) 4 << 20;
// Small synchronous uploads (the per-token graph inputs: tokens / embeddings, positions, KQ mask, output ids) are enqueued on the null
// stream WITHOUT a host-side wait: the source bytes have left the caller's buffer when hipMemcpyAsync returns (pageable memory is staged
// by the runtime), and whoever consumes the tensor next is ordered behind the copy on the DEVICE - graph_compute / async copies make
// their stream wait for an event recorded on the null stream, get_tensor and the other buffer functions run on the null stream
// themselves, synchronize() drains it. Saves one host round trip (~10 us) per input tensor and token.
std::atomic<uint64_t> g_null_epoch{0};             // bumped by every such upload; each backend remembers the epoch its stream is ordered after
// Round 6: those uploads are BATCHED - set_tensor copies the bytes into a pinned staging block and returns (the caller's buffer is free); the batch leaves as ONE
// hipMemcpyAsync into a device staging block + ONE scatter launch (pm355_scatter_bytes) on the null stream the next time anything could observe it: every
// function of this file that touches device memory, orders a stream behind the null stream, or synchronizes calls small_flush() first. Four pageable copies
// per token (token, positions, KQ mask row, output ids: ~77 us) become ~25 us. GGML_MI355_BATCH_UPLOADS=0 keeps the per-tensor copies.
struct small_stage {
    uint8_t * pinned[2] = {nullptr, nullptr}, * dev[2] = {nullptr, nullptr};
    pm355_event_t done[2] = {nullptr, nullptr}; bool busy[2] = {false, false};
    int slot = 0; size_t used = 0; int n = 0; pm355_scatter_seg seg[16];
};
constexpr size_t SMALL_CAP = (size_t) 256 << 10, SMALL_MAX = (size_t) 64 << 10;
small_stage g_small[GGML_MI355_MAX_DEVICES];
std::mutex g_small_mu;
bool small_batching() { static const bool v = !plan_only() && !(getenv("GGML_MI355_BATCH_UPLOADS") && getenv("GGML_MI355_BATCH_UPLOADS")[0] == '0'); return v; }
void small_flush_dev(int device) {
    small_stage & s = g_small[device];
    if (!s.n) return;
    dsetdev(device);
    MI355_CHECK(pm355_memcpy_h2d(s.dev[s.slot], s.pinned[s.slot], s.used, nullptr));
    MI355_CHECK(pm355_scatter_bytes(s.dev[s.slot], s.seg, s.n, nullptr));
    MI355_CHECK(pm355_event_record(s.done[s.slot], nullptr));
    s.busy[s.slot] = true;
    s.slot ^= 1; s.used = 0; s.n = 0;
    g_null_epoch.fetch_add(1, std::memory_order_release);
}
void small_flush() {
    if (!small_batching()) return;
    std::lock_guard<std::mutex> lk(g_small_mu);
    for (int d = 0; d < GGML_MI355_MAX_DEVICES; ++d) if (g_small[d].n) small_flush_dev(d);
}
void small_enqueue(int device, void * dst, const void * data, size_t size) {
    std::lock_guard<std::mutex> lk(g_small_mu);
    small_stage & s = g_small[device];
    if (!s.pinned[0]) {
        for (int i = 0; i < 2; ++i) {
            s.pinned[i] = (uint8_t *) pm355_host_malloc(SMALL_CAP); s.dev[i] = (uint8_t *) pm355_malloc(SMALL_CAP); s.done[i] = pm355_event_create();
            GGML_ASSERT(s.pinned[i] && s.dev[i] && s.done[i] && "small-upload staging");
        }
    }
    size_t at = (s.used + 15) & ~(size_t) 15;
    if (s.n == 16 || at + size > SMALL_CAP) { small_flush_dev(device); at = 0; }
    if (s.busy[s.slot]) { MI355_CHECK(pm355_event_sync(s.done[s.slot])); s.busy[s.slot] = false; }      // (two batches ago: long done)
    memcpy(s.pinned[s.slot] + at, data, size);
    s.seg[s.n++] = {dst, (uint32_t) at, (uint32_t) size};
    s.used = at + size;
}
void buf_drain(buf_ctx * c) {
    if (c->up_pending) { pm355_uploader_sync(c->up); c->up_pending = false; }
}
void drain_all_uploads() {
    if (!g_up_pending.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(g_up_mu);
    for (buf_ctx * c : g_up_bufs) buf_drain(c);
    g_up_pending.store(0, std::memory_order_release);
}
// device staging area of a buffer for the row-local repack (kept: the loader calls set_tensor once per weight tensor)
void * buf_stage(buf_ctx * c, size_t n) {
    if (n > c->stage_bytes) { dsync(nullptr); dfree(c->stage); c->stage = dmalloc(n); GGML_ASSERT(c->stage); c->stage_bytes = n; }
    return c->stage;
}
// one lowered graph the backend has seen: its per-token fingerprint, its launch plan and (after a warm-up run) the captured hipGraph
struct graph_entry {
    mi355::graph_fp fp;
    mi355::plan plan;
    pm355_graph_t exec = nullptr;
    int runs = 0;
    uint64_t last_use = 0;
};
struct backend_ctx {
    int device; std::string name; pm355_stream_t stream;
    void * scratch = nullptr; size_t scratch_bytes = 0;
    int32_t * d_i32 = nullptr;                       // small device scratch (positions)
    // graph lowering (ggml_graph_plan.h)
    int32_t * d_dyn = nullptr;                       // device int32[2]: {KV cell of the token, cells attended}
    float * rope_tab = nullptr;                      // device float[256]: this token's cos / sin per rotation pair (round-3 attention block)
    double * ss_buf = nullptr; bool use_ss = false;  // producer-side sum-of-squares partials, double[2][256]: opt-in (GGML_MI355_SS=1; same bits, measured -0.4 %)
    bool attn_mfma = true;                           // GGML_MI355_ATTN_MFMA=0: long contexts keep the round-2 flash-decoding kernel
    bool qkv_epi = true;                             // GGML_MI355_QKV_EPI=0: rope + KV store inside the attention kernel (the round-2 form)
    float * qkv = nullptr; size_t qkv_floats = 0;    // raw q / k / v projections of one token
    float * split = nullptr; size_t split_floats = 0;
    std::vector<graph_entry *> graphs;
    // the host told us (ggml_backend_mi355_set_graph_reused, reached through get_proc_address by the graph-reuse patch) that the ggml_cgraph objects it hands to
    // graph_compute are the SAME, un-rebuilt objects as last time: entries are then found by graph pointer + node count, no walk over the nodes
    bool host_reuses = false;
    struct by_ptr_t { const struct ggml_cgraph * g; int n_nodes; graph_entry * e; };
    std::vector<by_ptr_t> by_ptr;
    uint64_t n_ptr_hit = 0;
    mi355::graph_fp fp_tmp;
    uint64_t tick = 0;
    bool fuse = true, use_graphs = true, debug_plan = false;
    int split_min = 320;                             // cells attended from which the keys are split over workgroups (engine.hip: measured crossover)
    // counters (GGML_MI355_STATS=1 prints them when the backend is freed; tests read them through the log)
    uint64_t n_compute = 0, n_replay = 0, n_capture = 0, n_eager = 0, n_plan = 0, n_fp_hit = 0;
    pm355_event_t null_ev = nullptr; uint64_t null_epoch = 0;   // orders this stream behind the null stream's pending small uploads
};

// make `c`'s stream wait (on the device) for everything enqueued on the null stream so far
void order_after_null_stream(backend_ctx * c) {
    small_flush();
    const uint64_t ep = g_null_epoch.load(std::memory_order_acquire);
    if (ep == c->null_epoch || plan_only()) return;
    if (!c->null_ev) { c->null_ev = pm355_event_create(); GGML_ASSERT(c->null_ev); }
    c->null_epoch = ep;
    MI355_CHECK(pm355_event_record(c->null_ev, nullptr));
    MI355_CHECK(pm355_event_wait(c->stream, c->null_ev));
}

pm355_tensor to_pm(const struct ggml_tensor * t) {
    pm355_tensor d;
    d.data = t->data; d.type = (int32_t) t->type; d.pad_ = 0;
    for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = t->nb[i]; }
    return d;
}

// ------------------------------------------------------------------------------------------------ buffer
const char * buf_get_name(ggml_backend_buffer_t b) { return ((buf_ctx *) b->context)->name.c_str(); }
bool buffer_is_mi355(ggml_backend_buffer_t b) { return b && b->iface.get_name == buf_get_name; }

void buf_free(ggml_backend_buffer_t b) {
    small_flush();
    buf_ctx * c = (buf_ctx *) b->context;
    dsetdev(c->device);
    if (c->up) {
        std::lock_guard<std::mutex> lk(g_up_mu);
        buf_drain(c);
        for (size_t i = 0; i < g_up_bufs.size(); ++i) if (g_up_bufs[i] == c) { g_up_bufs.erase(g_up_bufs.begin() + i); break; }
        pm355_uploader_free(c->up);
    }
    dfree(c->base);
    dfree(c->stage);
    delete c;
}
void * buf_get_base(ggml_backend_buffer_t b) { return ((buf_ctx *) b->context)->base; }
void buf_init_tensor(ggml_backend_buffer_t, struct ggml_tensor *) {}

void buf_memset_tensor(ggml_backend_buffer_t b, struct ggml_tensor * t, uint8_t v, size_t off, size_t size) {
    small_flush();
    dsetdev(((buf_ctx *) b->context)->device);
    buf_drain((buf_ctx *) b->context);
    MI355_CHECK(dset((char *) t->data + off, v, size, nullptr));
    MI355_CHECK(dsync(nullptr));
}

// host GGUF-order bytes -> HBM layout (row-local repack for the row-SoA types; see repack.hip)
void buf_set_tensor(ggml_backend_buffer_t b, struct ggml_tensor * t, const void * data, size_t off, size_t size) {
    scoped_ns tm(g_ht.ns_set); ++g_ht.n_set;
    buf_ctx * c = (buf_ctx *) b->context;
    dsetdev(c->device);
    const bool soa = is_soa_tensor(t);
    const size_t rb = ggml_row_size(t->type, t->ne[0]), stride = soa ? pm355_row_stride(t->type, t->ne[0]) : rb;
    if (soa) {
        GGML_ASSERT(off % rb == 0 && size % rb == 0 && "row-granular access to a row-SoA tensor");
        GGML_ASSERT(!t->view_src && "row-SoA tensors are addressed per allocated tensor, not through views");
    }
    if (size >= UPLOAD_ASYNC_MIN) small_flush();
    if (size >= UPLOAD_ASYNC_MIN && !plan_only() && !env_on("GGML_MI355_SYNC_UPLOAD")) {
        // weights: pinned ring + copier threads + private stream, returns once the bytes have left `data`
        if (!c->up) {
            c->up = pm355_uploader_new(0, -1);
            GGML_ASSERT(c->up && "uploader: pinned memory / stream");
            std::lock_guard<std::mutex> lk(g_up_mu);
            g_up_bufs.push_back(c);
        }
        MI355_CHECK(dsync(nullptr));                                   // order after earlier synchronous writes to this range
        MI355_CHECK(pm355_upload(c->up, (int) t->type, t->ne[0], data, (char *) t->data + (soa ? (off / rb) * stride : off), size, soa ? 1 : 0));
        c->up_pending = true;
        g_up_pending.store(1, std::memory_order_release);
        return;
    }
    buf_drain(c);
    if (!soa && small_batching() && size <= SMALL_MAX && !env_on("GGML_MI355_SYNC_UPLOAD")) {
        // per-token graph inputs: into the pinned staging block (the caller's bytes are consumed here), on their way with the next flush
        small_enqueue(c->device, (char *) t->data + off, data, size);
        return;
    }
    small_flush();
    if (!soa && !plan_only() && size <= ((size_t) 1 << 20) && !env_on("GGML_MI355_SYNC_UPLOAD")) {
        // per-token graph inputs: enqueued on the null stream, the compute stream is ordered behind them on the device. The source bytes
        // have left `data` when hipMemcpyAsync returns ONLY for pageable memory (the runtime stages it); a page-locked source - e.g. a
        // tensor of the plug-in's own host buffer type handed over by ggml_backend_tensor_copy - is read by the DMA later, and the
        // scheduler relies on the copy being complete on return (ggml-backend.cpp:2110): wait for it.
        MI355_CHECK(h2d((char *) t->data + off, data, size, nullptr));
        if (pm355_host_is_pinned(data)) { MI355_CHECK(pm355_sync_null_stream()); return; }      // (the copy's own stream, not the whole device)
        g_null_epoch.fetch_add(1, std::memory_order_release);
        return;
    }
    if (soa) {
        void * stage = buf_stage((buf_ctx *) b->context, size);
        MI355_CHECK(h2d(stage, data, size, nullptr));
        MI355_CHECK(drepack(t->type, stage, (char *) t->data + (off / rb) * stride, t->ne[0], (int64_t) (size / rb), 1, nullptr));
        MI355_CHECK(dsync(nullptr));
        return;
    }
    MI355_CHECK(h2d((char *) t->data + off, data, size, nullptr));
    MI355_CHECK(dsync(nullptr));
}
void buf_get_tensor(ggml_backend_buffer_t b, const struct ggml_tensor * t, void * data, size_t off, size_t size) {
    scoped_ns tm(g_ht.ns_get); ++g_ht.n_get;
    small_flush();
    dsetdev(((buf_ctx *) b->context)->device);
    buf_drain((buf_ctx *) b->context);
    if (is_soa_tensor(t)) {
        const size_t rb = ggml_row_size(t->type, t->ne[0]), stride = pm355_row_stride(t->type, t->ne[0]);
        GGML_ASSERT(off % rb == 0 && size % rb == 0 && "row-granular access to a row-SoA tensor");
        GGML_ASSERT(!t->view_src && "row-SoA tensors are addressed per allocated tensor, not through views");
        void * stage = buf_stage((buf_ctx *) b->context, size);
        MI355_CHECK(drepack(t->type, (const char *) t->data + (off / rb) * stride, stage, t->ne[0], (int64_t) (size / rb), 0, nullptr));
        MI355_CHECK(d2h(data, stage, size, nullptr));
        MI355_CHECK(dsync(nullptr));
        return;
    }
    MI355_CHECK(d2h(data, (const char *) t->data + off, size, nullptr));
    MI355_CHECK(dsync(nullptr));
}
size_t hbm_bytes(const struct ggml_tensor * t) {
    if (is_soa_tensor(t)) return pm355_row_stride(t->type, t->ne[0]) * (size_t) (t->ne[1] * t->ne[2] * t->ne[3]);
    return ggml_nbytes(t);
}
bool buf_cpy_tensor(ggml_backend_buffer_t b, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    small_flush();
    if (!buffer_is_mi355(src->buffer)) return false;                 // ggml falls back to get + set through the host
    buf_ctx * sc = (buf_ctx *) src->buffer->context, * dc = (buf_ctx *) b->context;
    if (sc->device != dc->device || src->type != dst->type || !ggml_is_contiguous(src) || !ggml_is_contiguous(dst)) return false;
    dsetdev(dc->device);
    buf_drain(sc); buf_drain(dc);
    MI355_CHECK(d2d(dst->data, src->data, hbm_bytes(src), nullptr));
    MI355_CHECK(dsync(nullptr));
    return true;
}
void buf_clear(ggml_backend_buffer_t b, uint8_t v) {
    small_flush();
    buf_ctx * c = (buf_ctx *) b->context;
    dsetdev(c->device);
    buf_drain(c);
    MI355_CHECK(dset(c->base, v, c->size, nullptr));
    MI355_CHECK(dsync(nullptr));
}
const struct ggml_backend_buffer_i buffer_iface = {
    /* .get_name      = */ buf_get_name,
    /* .free_buffer   = */ buf_free,
    /* .get_base      = */ buf_get_base,
    /* .init_tensor   = */ buf_init_tensor,
    /* .memset_tensor = */ buf_memset_tensor,
    /* .set_tensor    = */ buf_set_tensor,
    /* .get_tensor    = */ buf_get_tensor,
    /* .cpy_tensor    = */ buf_cpy_tensor,
    /* .clear         = */ buf_clear,
    /* .reset         = */ nullptr,
};

// ------------------------------------------------------------------------------------------------ buffer type
const char * buft_get_name(ggml_backend_buffer_type_t t) { return ((buft_ctx *) t->context)->name.c_str(); }
bool buft_is_mi355(ggml_backend_buffer_type_t t) { return t && t->iface.get_name == buft_get_name; }

ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    buft_ctx * c = (buft_ctx *) t->context;
    dsetdev(c->device);
    size = size ? size : 1;
    void * p = dmalloc(size + 256);              // tail slack for 16-byte vector reads
    if (!p) { fprintf(stderr, "ggml-mi355: allocating %.2f MiB on device %d failed\n", size / 1048576.0, c->device); return nullptr; }
    return ggml_backend_buffer_init(t, buffer_iface, new buf_ctx{c->device, p, size, c->name}, size);
}
size_t buft_alignment(ggml_backend_buffer_type_t) { return 128; }
size_t buft_alloc_size(ggml_backend_buffer_type_t, const struct ggml_tensor * t) { return hbm_bytes(t); }
bool buft_is_host(ggml_backend_buffer_type_t) { return false; }

const struct ggml_backend_buffer_type_i buft_iface = {
    /* .get_name       = */ buft_get_name,
    /* .alloc_buffer   = */ buft_alloc,
    /* .get_alignment  = */ buft_alignment,
    /* .get_max_size   = */ nullptr,
    /* .get_alloc_size = */ buft_alloc_size,
    /* .is_host        = */ buft_is_host,
};

// pinned host buffers: a CPU buffer over hipHostMalloc memory (the scheduler puts CPU-side compute buffers here)
const char * host_buft_name(ggml_backend_buffer_type_t) { return GGML_MI355_NAME "_Host"; }
void host_buf_free(ggml_backend_buffer_t b) { if (plan_only()) free(b->context); else pm355_host_free(b->context); }
ggml_backend_buffer_t host_buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    void * p = plan_only() ? aligned_alloc(256, ((size ? size : 1) + 255) & ~(size_t) 255) : pm355_host_malloc(size ? size : 1);
    if (!p) return ggml_backend_buft_alloc_buffer(ggml_backend_cpu_buffer_type(), size);
    ggml_backend_buffer_t b = ggml_backend_cpu_buffer_from_ptr(p, size);
    b->buft = t;
    b->iface.free_buffer = host_buf_free;
    return b;
}

// ------------------------------------------------------------------------------------------------ backend (stream)
const char * backend_name(ggml_backend_t b) { return ((backend_ctx *) b->context)->name.c_str(); }
void backend_free(ggml_backend_t b) {
    backend_ctx * c = (backend_ctx *) b->context;
    dsetdev(c->device);
    dsync(c->stream);
    if (env_on("GGML_MI355_STATS"))
        fprintf(stderr, "ggml-mi355 stats: graph_compute %llu, fingerprint hits %llu, graph-pointer hits %llu, plans built %llu, hipGraph replays %llu, captures %llu, eager runs %llu\n",
                (unsigned long long) c->n_compute, (unsigned long long) c->n_fp_hit, (unsigned long long) c->n_ptr_hit, (unsigned long long) c->n_plan, (unsigned long long) c->n_replay,
                (unsigned long long) c->n_capture, (unsigned long long) c->n_eager);
    if (env_on("GGML_MI355_STATS"))
        fprintf(stderr, "ggml-mi355 host time: graph_compute %.3f ms total (%.1f us per call), set_tensor %.3f ms in %llu calls, get_tensor %.3f ms in %llu calls, "
                        "synchronize %.3f ms in %llu calls\n", g_ht.ns_compute / 1e6, c->n_compute ? g_ht.ns_compute / 1e3 / (double) c->n_compute : 0.0,
                g_ht.ns_set / 1e6, (unsigned long long) g_ht.n_set.load(), g_ht.ns_get / 1e6, (unsigned long long) g_ht.n_get.load(),
                g_ht.ns_sync / 1e6, (unsigned long long) g_ht.n_sync.load());
    if (env_on("GGML_MI355_STATS") && c->n_compute)
        fprintf(stderr, "ggml-mi355 graph_compute phases, us per call: order behind uploads %.1f, fingerprint %.1f, plan lookup / build %.1f, KV cell lookup + patch %.1f, launch %.1f\n",
                g_ht.ns_order / 1e3 / (double) c->n_compute, g_ht.ns_fp / 1e3 / (double) c->n_compute, g_ht.ns_plan / 1e3 / (double) c->n_compute,
                g_ht.ns_dyn / 1e3 / (double) c->n_compute, g_ht.ns_launch / 1e3 / (double) c->n_compute);
    // (the snapshot and the phase timers are process-wide: with more than one backend they mix counters of different backends - the steady-state
    //  split is printed only when this process had a single one)
    if (env_on("GGML_MI355_STATS") && g_n_backends.load() == 1 && g_snap.taken && c->n_replay > g_snap.v[11] + 1) {
        // from the first replay to the last: one replay per token, so (last - first) spans n - 1 whole tokens
        uint64_t now[12]; host_counters(now);
        const double n = (double) (c->n_replay - g_snap.v[11] - 1);
        const double wall = std::chrono::duration_cast<std::chrono::nanoseconds>(g_snap.t_last - g_snap.t0).count() / 1e3 / n;
        auto d = [&](int i) { return (double) (now[i] - g_snap.v[i]) / 1e3 / n; };
        fprintf(stderr, "ggml-mi355 steady state, us per token over %.0f tokens: wall %.1f = graph_compute %.1f (fingerprint %.1f, plan lookup %.1f, KV patch %.1f, launch %.1f) "
                        "+ input uploads %.1f + output download (waits for the device) %.1f + synchronize %.1f + outside the plug-in (libllama graph build / scheduler, sampling) %.1f\n",
                n, wall, d(0), d(5), d(6), d(7), d(8), d(1) + d(9), d(2) + d(10), d(3), wall - d(0) - d(1) - d(9) - d(2) - d(10) - d(3));
    }
    for (graph_entry * e : c->graphs) { if (e->exec) pm355_graph_free(e->exec); delete e; }
    dfree(c->scratch); dfree(c->d_i32); dfree(c->d_dyn); dfree(c->rope_tab); dfree(c->ss_buf); dfree(c->qkv); dfree(c->split);
    if (c->null_ev) pm355_event_destroy(c->null_ev);
    if (!plan_only()) pm355_stream_destroy(c->stream);
    delete c; delete b;
}
void backend_set_async(ggml_backend_t b, struct ggml_tensor * t, const void * data, size_t off, size_t size) {
    scoped_ns tm(g_ht.ns_seta);
    backend_ctx * c = (backend_ctx *) b->context;
    dsetdev(c->device);
    drain_all_uploads();
    order_after_null_stream(c);
    if (is_soa_tensor(t)) { dsync(c->stream); buf_set_tensor(t->view_src ? t->view_src->buffer : t->buffer, t, data, off, size); return; }
    MI355_CHECK(h2d((char *) t->data + off, data, size, c->stream));
}
void backend_get_async(ggml_backend_t b, const struct ggml_tensor * t, void * data, size_t off, size_t size) {
    scoped_ns tm(g_ht.ns_geta);                        // (a copy into pageable host memory returns when the bytes have arrived: this is where a token waits for the device)
    backend_ctx * c = (backend_ctx *) b->context;
    dsetdev(c->device);
    drain_all_uploads();
    order_after_null_stream(c);
    if (is_soa_tensor(t)) { dsync(c->stream); buf_get_tensor(t->view_src ? t->view_src->buffer : t->buffer, t, data, off, size); return; }
    MI355_CHECK(d2h(data, (const char *) t->data + off, size, c->stream));
}
void backend_sync(ggml_backend_t b) {
    scoped_ns tm(g_ht.ns_sync); ++g_ht.n_sync;
    backend_ctx * c = (backend_ctx *) b->context;
    dsetdev(c->device);
    drain_all_uploads();
    // (small uploads pending on the null stream are not waited for here: their consumers are ordered behind them on the device)
    MI355_CHECK(dsync(c->stream));
}

void * scratch(backend_ctx * c, size_t bytes) {
    if (bytes > c->scratch_bytes) {
        dsync(c->stream);
        dfree(c->scratch);
        c->scratch = dmalloc(bytes + 256);
        GGML_ASSERT(c->scratch);
        c->scratch_bytes = bytes;
    }
    return c->scratch;
}

bool mul_mat_quant_ok(const struct ggml_tensor * op) {
    const struct ggml_tensor * a = op->src[0], * b = op->src[1];
    if (!is_gemv_type(a->type) || b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
    if (!ggml_is_contiguous(a) || !ggml_is_contiguous(b) || !ggml_is_contiguous(op)) return false;
    if (a->ne[2] != 1 || a->ne[3] != 1 || b->ne[2] != 1 || b->ne[3] != 1) return false;
    if (is_soa_type(a->type) && (a->view_src || !is_soa_tensor(a))) return false;   // row-SoA layout is per allocated weight matrix (not views, not the 1-D quantized KV caches)
    if (a->ne[0] % (a->type == GGML_TYPE_Q8_0 ? 32 : 256)) return false;
    return a->ne[0] <= 131072;
}

bool supports_op_impl(const struct ggml_tensor * op) {
    const struct ggml_tensor * a = op->src[0], * b = op->src[1];
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT:
            if (mul_mat_quant_ok(op)) return true;
            // K.q on a `-ctk q8_0` cache without flash attention: src0 = a view of native Q8_0 blocks (never a row-SoA weight matrix)
            if (a->type == GGML_TYPE_Q8_0 && !is_soa_tensor(a) && b->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && a->ne[0] % 32 == 0 &&
                a->nb[0] == ggml_type_size(a->type)) return true;
            return (a->type == GGML_TYPE_F16 || a->type == GGML_TYPE_F32) && b->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 &&
                   a->nb[0] == ggml_type_size(a->type) && b->nb[0] == 4;
        case GGML_OP_RMS_NORM:
            return a->type == GGML_TYPE_F32 && a->nb[0] == 4 && op->nb[0] == 4;
        case GGML_OP_ADD: case GGML_OP_MUL:
            return a->type == GGML_TYPE_F32 && b->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32;
        case GGML_OP_SCALE:
            return a->type == GGML_TYPE_F32;
        case GGML_OP_UNARY:
            return ggml_get_unary_op(op) == GGML_UNARY_OP_SILU && a->type == GGML_TYPE_F32;
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: {
            const enum ggml_type ts = a->type, td = op->type;
            // KV store into a quantized cache: f32 rows -> contiguous native Q8_0 blocks of a 1-D cache tensor (attn_q8.hip)
            if (op->op == GGML_OP_CPY && ts == GGML_TYPE_F32 && td == GGML_TYPE_Q8_0)
                return a->nb[0] == 4 && a->ne[0] % 32 == 0 && ggml_is_contiguous(op) && op->view_src && !is_soa_tensor(op);
            // defragmentation of a quantized cache: block-wise copy between two views of native Q8_0 blocks (build_defrag, src/llama.cpp:10721)
            if (ts == GGML_TYPE_Q8_0 && td == GGML_TYPE_Q8_0)
                return op->op == GGML_OP_CPY && !is_soa_tensor(a) && !is_soa_tensor(op) && a->ne[0] % 32 == 0 && ggml_are_same_shape(a, op) &&
                       a->nb[0] == ggml_type_size(a->type) && op->nb[0] == ggml_type_size(op->type);
            // K-shift of a quantized cache: dequantizing copy of native Q8_0 rows (build_k_shift, src/llama.cpp:10665)
            if (ts == GGML_TYPE_Q8_0 && td == GGML_TYPE_F32) return !is_soa_tensor(a) && a->ne[0] % 32 == 0 && a->nb[0] == ggml_type_size(a->type);
            return (ts == GGML_TYPE_F32 || ts == GGML_TYPE_F16) && (td == GGML_TYPE_F32 || td == GGML_TYPE_F16);
        }
        case GGML_OP_SOFT_MAX:
            return a->type == GGML_TYPE_F32 && a->nb[0] == 4 && ggml_is_contiguous(a) && a->ne[0] * 4 <= 150 * 1024 &&
                   (!b || b->type == GGML_TYPE_F32 || b->type == GGML_TYPE_F16);
        case GGML_OP_ROPE: {
            const int mode = ((const int32_t *) op->op_params)[2];
            // F16: the in-place K-shift of an F16 cache (build_k_shift, src/llama.cpp:10708-10713)
            return (a->type == GGML_TYPE_F32 || a->type == GGML_TYPE_F16) && op->type == a->type && (mode == 0 || mode == 2) && a->ne[0] % 2 == 0;
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            // F16 K / V (the default KV cache type), or Q8_0 K and / or V (-ctk / -ctv q8_0: native blocks, attn_q8.hip)
            const struct ggml_tensor * k = op->src[1], * v = op->src[2], * m = op->src[3];
            const int64_t D = a->ne[0];
            if (a->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && (k->type == GGML_TYPE_Q8_0 || v->type == GGML_TYPE_Q8_0)) {
                float max_bias; memcpy(&max_bias, (const float *) op->op_params + 1, 4);
                if ((k->type != GGML_TYPE_Q8_0 && k->type != GGML_TYPE_F16) || (v->type != GGML_TYPE_Q8_0 && v->type != GGML_TYPE_F16)) return false;
                if ((k->type == GGML_TYPE_Q8_0 && is_soa_tensor(k)) || (v->type == GGML_TYPE_Q8_0 && is_soa_tensor(v))) return false;
                if ((D != 64 && D != 128 && D != 256) || a->nb[0] != 4 || max_bias != 0.0f || (m && m->type != GGML_TYPE_F16)) return false;
                return (size_t) (D + ((k->ne[1] + 3) & ~3) + (256 / (D / 8)) * D) * 4 <= 150 * 1024;
            }
            if (a->type != GGML_TYPE_F32 || k->type != GGML_TYPE_F16 || v->type != GGML_TYPE_F16 || op->type != GGML_TYPE_F32) return false;
            if (D < 8 || D > 256 || a->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2 || (m && m->type != GGML_TYPE_F16)) return false;
            const int64_t Dp = (D + 7) & ~7;
            return (size_t) (Dp + ((k->ne[1] + 3) & ~3) + (256 / (Dp / 8)) * Dp) * 4 <= 150 * 1024;     // scores of one query live in LDS
        }
        case GGML_OP_GET_ROWS:
            return b->type == GGML_TYPE_I32 && ggml_is_contiguous(b) && b->ne[1] == 1 && b->ne[2] == 1 && b->ne[3] == 1 &&
                   (a->type == GGML_TYPE_F32 || (is_gemv_type(a->type) && ggml_is_contiguous(a) && a->ne[2] == 1 && a->ne[3] == 1 && !a->view_src && (!is_soa_type(a->type) || is_soa_tensor(a))));
        default:
            return false;
    }
}

bool compute_node(backend_ctx * c, struct ggml_tensor * op) {
    const struct ggml_tensor * a = op->src[0], * b = op->src[1];
    pm355_stream_t st = c->stream;
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT: {
            if (mul_mat_quant_ok(op)) {
                // one fused launch per activation column: f32 -> vec_dot_type in the kernel prologue, then the mat-vec
                const int64_t K = a->ne[0], N = a->ne[1], ncols = b->ne[1];
                // 3..32 columns (parallel sequences, speculative batches, short prompts; pm355_small_batch_max_tokens): one pass over the weights on the integer
                // matrix cores; GGML_MI355_NO_MMQ_I8=1 restores the paths around it (one mat-vec per column / F16 GEMM from 16)
                static const bool no_small = [] { const char * e = getenv("GGML_MI355_NO_MMQ_I8"); return e && e[0] == '1'; }();
                static const int small_max = pm355_small_batch_max_tokens();
                if (!no_small && ncols >= 3 && ncols <= small_max && pm355_mul_mat_q_small_check((int) a->type, K, N, ncols) == 0) {
                    MI355_CHECK(pm355_mul_mat_q_small((int) a->type, a->data, K, N, nullptr, (const float *) b->data, ncols, (float *) op->data, nullptr, nullptr, st));
                    return true;
                }
                // GGML_MI355_PROMPT_I8=1: prompt batches with the CPU backend's own arithmetic - src1 quantized to Q8_K, integer matrix cores
                // (mmq_big.hip; the CUDA plug-in's mul_mat_q design, ggml-cuda/mmq.cuh:2583) - at 0.6-0.75 of the F16 GEMM's rate (the default below)
                static const bool prompt_i8 = [] { const char * e = getenv("GGML_MI355_PROMPT_I8"); return e && e[0] == '1'; }();
                if (prompt_i8 && ncols > 64 && pm355_mul_mat_q_i8_check((int) a->type, K, N, ncols) == 0) {
                    MI355_CHECK(pm355_mul_mat_q_i8((int) a->type, a->data, K, N, (const float *) b->data, ncols, (float *) op->data, nullptr, nullptr, st));
                    return true;
                }
                if (ncols >= 16 && K % 64 == 0 && N % 4 == 0) {          // prefill: batched GEMM on the MFMA matrix cores
                    MI355_CHECK(pm355_mul_mat_q_mfma((int) a->type, a->data, K, N, (const float *) b->data, ncols, (float *) op->data, nullptr, nullptr, st));
                    return true;
                }
                for (int64_t col = 0; col < ncols; ++col) {
                    pm355_matvec_job job = {};
                    job.type = (int32_t) a->type; job.N = N; job.W = a->data; job.y = (float *) op->data + col * N;
                    MI355_CHECK(pm355_mul_mat_vec_fused(&job, 1, K, (const float *) b->data + col * K, nullptr, 0.0f, st));
                }
                return true;
            }
            pm355_tensor ta = to_pm(a), tb = to_pm(b), td = to_pm(op);
            MI355_CHECK(pm355_op_mul_mat_f(&ta, &tb, &td, st));
            return true;
        }
        case GGML_OP_RMS_NORM: {
            float eps; memcpy(&eps, op->op_params, sizeof(float));
            pm355_tensor ta = to_pm(a), td = to_pm(op);
            MI355_CHECK(pm355_op_rms_norm(&ta, &td, eps, st));
            return true;
        }
        case GGML_OP_ADD: case GGML_OP_MUL: {
            pm355_tensor ta = to_pm(a), tb = to_pm(b), td = to_pm(op);
            MI355_CHECK(pm355_op_binary(op->op == GGML_OP_ADD ? 0 : 1, &ta, &tb, &td, st));
            return true;
        }
        case GGML_OP_SCALE: {
            float s; memcpy(&s, op->op_params, sizeof(float));
            pm355_tensor ta = to_pm(a), td = to_pm(op);
            MI355_CHECK(pm355_op_unary(0, &ta, &td, s, st));
            return true;
        }
        case GGML_OP_UNARY: {
            pm355_tensor ta = to_pm(a), td = to_pm(op);
            MI355_CHECK(pm355_op_unary(1, &ta, &td, 0.0f, st));
            return true;
        }
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: {
            // CPY writes into src[1]'s storage (dst is a view of it); DUP/CONT into dst
            pm355_tensor ta = to_pm(a), td = to_pm(op);
            MI355_CHECK(pm355_op_cpy(&ta, &td, st));
            return true;
        }
        case GGML_OP_SOFT_MAX: {
            float scale, max_bias;
            memcpy(&scale, (const float *) op->op_params + 0, sizeof(float));
            memcpy(&max_bias, (const float *) op->op_params + 1, sizeof(float));
            pm355_tensor ta = to_pm(a), td = to_pm(op), tm;
            if (b) tm = to_pm(b);
            MI355_CHECK(pm355_op_soft_max(&ta, b ? &tm : nullptr, &td, scale, max_bias, st));
            return true;
        }
        case GGML_OP_ROPE: {
            const int32_t * prm = (const int32_t *) op->op_params;
            pm355_rope_params rp;
            rp.n_dims = prm[1]; rp.mode = prm[2]; rp.n_ctx_orig = prm[4];
            memcpy(&rp.freq_base, prm + 5, 4); memcpy(&rp.freq_scale, prm + 6, 4); memcpy(&rp.ext_factor, prm + 7, 4);
            memcpy(&rp.attn_factor, prm + 8, 4); memcpy(&rp.beta_fast, prm + 9, 4); memcpy(&rp.beta_slow, prm + 10, 4);
            pm355_tensor ta = to_pm(a), td = to_pm(op);
            const struct ggml_tensor * ff = op->src[2];
            MI355_CHECK(pm355_op_rope(&ta, (const int32_t *) b->data, ff ? (const float *) ff->data : nullptr, &td, &rp, st));
            return true;
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            float scale, max_bias, softcap;
            memcpy(&scale, (const float *) op->op_params + 0, 4); memcpy(&max_bias, (const float *) op->op_params + 1, 4);
            memcpy(&softcap, (const float *) op->op_params + 2, 4);
            pm355_tensor tq = to_pm(a), tk = to_pm(op->src[1]), tv = to_pm(op->src[2]), td = to_pm(op), tm;
            if (op->src[3]) tm = to_pm(op->src[3]);
            MI355_CHECK(pm355_op_flash_attn_ext(&tq, &tk, &tv, op->src[3] ? &tm : nullptr, &td, scale, max_bias, softcap, st));
            return true;
        }
        case GGML_OP_GET_ROWS: {
            if (a->type == GGML_TYPE_F32) {
                pm355_tensor ta = to_pm(a), td = to_pm(op);
                MI355_CHECK(pm355_op_get_rows_f32(&ta, (const int32_t *) b->data, b->ne[0], &td, st));
            } else {
                MI355_CHECK(pm355_get_rows((int) a->type, a->data, a->ne[0], (const int32_t *) b->data, (int) b->ne[0], (float *) op->data, st));
            }
            return true;
        }
        default:
            return false;
    }
}

// scratch callbacks of the planner: stable pointers while large enough (a pointer that moves changes the plan -> new capture)
float * plan_qkv_scratch(void * user, size_t n_q, size_t n_kv) {
    backend_ctx * c = (backend_ctx *) user;
    const size_t need = n_q + 2 * n_kv;
    if (need > c->qkv_floats) {
        dsync(c->stream);
        dfree(c->qkv);
        c->qkv = (float *) dmalloc(need * 4 + 256);
        c->qkv_floats = c->qkv ? need : 0;
    }
    return c->qkv;
}
float * plan_split_scratch(void * user, size_t n) {
    backend_ctx * c = (backend_ctx *) user;
    if (n > c->split_floats) {
        dsync(c->stream);
        dfree(c->split);
        c->split = (float *) dmalloc(n * 4 + 256);
        c->split_floats = c->split ? n : 0;
        if (c->split) { dset(c->split, 0, n * 4 + 256, nullptr); dsync(nullptr); }      // the flash-decoding tickets start at zero
    }
    return c->split;
}

// two small device arrays with the same bytes? (plan time only: the per-layer copies of rope_freqs.weight)
bool plan_same_bytes(void * user, const void * a, const void * b, size_t n) {
    backend_ctx * c = (backend_ctx *) user;
    if (n > 4096) return false;
    char ha[4096], hb[4096];
    if (plan_only()) { memcpy(ha, a, n); memcpy(hb, b, n); }
    else {
        dsync(c->stream); dsync(nullptr);
        if (pm355_memcpy_d2h(ha, a, n, nullptr) || pm355_memcpy_d2h(hb, b, n, nullptr)) return false;
        dsync(nullptr);
    }
    return memcmp(ha, hb, n) == 0;
}

// the launch sequence of a plan on the backend's stream (also what gets captured into a hipGraph)
bool run_plan(backend_ctx * c, struct ggml_cgraph * g, const mi355::plan & p) {
    for (const mi355::step & s : p.steps) {
        switch (s.kind) {
            case mi355::STEP_GEMV:
                MI355_CHECK(pm355_mul_mat_vec_fused_ss(s.job, s.njobs, s.K, s.x, s.norm_w, s.eps, s.ss_out, s.ss_in, s.n_ss, c->stream));
                break;
            case mi355::STEP_ATTN:
                MI355_CHECK(pm355_attn_token(&s.attn, &s.rope, c->stream));
                break;
            case mi355::STEP_ROPE_TAB:
                MI355_CHECK(pm355_rope_table(&s.rope, s.attn.d_pos, s.attn.freq_factors, (float *) s.qs.rope_table, c->stream));
                break;
            case mi355::STEP_QKV:
                MI355_CHECK(pm355_mul_mat_vec_qkv_ss(s.job, s.K, s.x, s.norm_w, s.eps, &s.qs, s.ss_in, s.n_ss, c->stream));
                break;
            case mi355::STEP_ATTN_CACHED:
                if (s.attn.split) {
                    MI355_CHECK(pm355_attn_cached_long(s.attn.q, s.attn.k_cache, s.attn.v_cache, nullptr, s.attn.d_cell_nkv, s.attn.mask, s.attn.out, s.attn.scratch,
                                                       s.attn.n_head, s.attn.n_head_kv, s.attn.head_dim, s.attn.n_ctx, s.attn.kq_scale, s.attn.max_keys, s.attn.flags, c->stream));
                    break;
                }
                MI355_CHECK(pm355_attn_cached(s.attn.q, s.attn.k_cache, s.attn.v_cache, nullptr, s.attn.d_cell_nkv, s.attn.mask, s.attn.out, s.attn.n_head,
                                              s.attn.n_head_kv, s.attn.head_dim, s.attn.n_ctx, s.attn.kq_scale, s.attn.max_keys, s.attn.flags, c->stream));
                break;
            case mi355::STEP_ATTN_BATCH:
                MI355_CHECK(pm355_attn_prefill_masked_ex(s.ab.q, s.ab.kc, s.ab.vc, s.ab.mask, s.ab.mask_stride, s.ab.out, s.ab.n_tokens, s.ab.n_head,
                                                         s.ab.n_head_kv, s.ab.head_dim, s.ab.n_ctx, s.ab.n_kv, s.ab.scale, s.ab.flags, c->stream));
                break;
            default: {
                struct ggml_tensor * node = ggml_graph_node(g, s.node);
                if (!compute_node(c, node)) {
                    fprintf(stderr, "ggml-mi355: op %s not supported (supports_op should have rejected it)\n", ggml_op_name(node->op));
                    return false;
                }
            }
        }
    }
    return true;
}

void print_plan(const backend_ctx * c, struct ggml_cgraph * g, const mi355::plan & p, int32_t cell, int32_t n_kv) {
    if (p.n_attn_batch) fprintf(stderr, "ggml-mi355 plan: %d multi-token attention chain(s) -> MFMA masked attention\n", p.n_attn_batch);
    fprintf(stderr, "ggml-mi355 plan: %d nodes -> %zu launches (%d fused mat-vec, %d attention, %d node-equivalent; %d nodes fused) single_token=%d "
                    "cell=%d n_kv=%d graphable=%d\n", ggml_graph_n_nodes(g), p.steps.size(), p.n_gemv, p.n_attn, p.n_node, p.n_fused_nodes,
            (int) p.single_token, cell, n_kv, (int) (p.single_token && p.fast_ok));
    if (!env_on("GGML_MI355_DEBUG_PLAN_STEPS")) return;
    for (const mi355::step & s : p.steps) {
        if (s.kind == mi355::STEP_ROPE_TAB) {
            fprintf(stderr, "  [%d] rope table n_dims=%d mode=%d ff=%d\n", s.node_lo, s.rope.n_dims, s.rope.mode, s.attn.freq_factors != nullptr);
        } else if (s.kind == mi355::STEP_GEMV || s.kind == mi355::STEP_QKV) {
            fprintf(stderr, "  [%d,%d) matvec%s K=%lld norm=%d%s%s jobs=%d:", s.node_lo, s.node_hi, s.kind == mi355::STEP_QKV ? " + rope + KV store" : "", (long long) s.K, s.norm_w != nullptr,
                    s.ss_in ? " sumsq<-producer" : "", s.ss_out ? " sumsq->consumer" : "", s.njobs);
            for (int j = 0; j < s.njobs; ++j) fprintf(stderr, " {%s N=%lld%s%s%s}", ggml_type_name((enum ggml_type) s.job[j].type), (long long) s.job[j].N,
                                                     s.job[j].W2 ? " pair" : "", s.job[j].bias ? " +bias" : "", s.job[j].resid ? " +resid" : "");
            fprintf(stderr, "\n");
        } else if (s.kind == mi355::STEP_ATTN_BATCH) {
            fprintf(stderr, "  [%d,%d) batch attention T=%d H=%d Hkv=%d dh=%d n_kv=%d (MFMA, masked)\n", s.node_lo, s.node_hi, s.ab.n_tokens, s.ab.n_head,
                    s.ab.n_head_kv, s.ab.head_dim, s.ab.n_kv);
        } else if (s.kind == mi355::STEP_ATTN || s.kind == mi355::STEP_ATTN_CACHED) {
            fprintf(stderr, "  [%d,%d) attention H=%d Hkv=%d dh=%d n_ctx=%d %s mask=%d ff=%d\n", s.node_lo, s.node_hi, s.attn.n_head, s.attn.n_head_kv, s.attn.head_dim,
                    s.attn.n_ctx, s.kind == mi355::STEP_ATTN_CACHED ? (s.attn.split ? "cached-split" : "cached") : (s.attn.split ? "split" : "fused"), s.attn.mask != nullptr, s.attn.freq_factors != nullptr);
        } else {
            fprintf(stderr, "  [%d] %s '%s'\n", s.node, ggml_op_name(ggml_graph_node(g, s.node)->op), ggml_graph_node(g, s.node)->name);
        }
    }
    (void) c;
}

// graph_compute (ggml-backend-impl.h:112; called per split by ggml_backend_sched_compute_splits, ggml-backend.cpp:2139):
//   1. per-token fingerprint of the split; a hit on a cached entry skips planning
//   2. else lower the graph to a launch plan (fusion by structural pattern matching); an equal plan of another entry is re-used
//   3. write the token's KV cell / cells attended to the device, then replay the entry's hipGraph - captured on the entry's second
//      run, after one eager warm-up run (function attributes, scratch growth) - or run the launches eagerly
enum ggml_status backend_graph_compute(ggml_backend_t b, struct ggml_cgraph * g) {
    scoped_ns tm(g_ht.ns_compute);
    backend_ctx * c = (backend_ctx *) b->context;
    dsetdev(c->device);
    { scoped_ns t1(g_ht.ns_order);
      drain_all_uploads();                             // weights staged asynchronously by set_tensor (one atomic load when none are pending)
      order_after_null_stream(c); }                    // the graph inputs uploaded since the last call
    const int n_nodes = ggml_graph_n_nodes(g);        // public accessors: struct ggml_cgraph is private to ggml (ggml-impl.h:183)
    ++c->n_compute; ++c->tick;
    if (n_nodes == 0) return GGML_STATUS_SUCCESS;
    if (!c->d_dyn) { c->d_dyn = (int32_t *) dmalloc(64); GGML_ASSERT(c->d_dyn); }
    if (!c->rope_tab && c->qkv_epi) { c->rope_tab = (float *) dmalloc(1024 + 64); GGML_ASSERT(c->rope_tab); }
    if (!c->ss_buf && c->use_ss) { c->ss_buf = (double *) dmalloc(2 * 256 * sizeof(double)); GGML_ASSERT(c->ss_buf); }

    graph_entry * e = nullptr;
    if (c->host_reuses) for (const auto & q : c->by_ptr) if (q.g == g && q.n_nodes == n_nodes && q.e->plan.fast_ok) { e = q.e; ++c->n_ptr_hit; break; }
    if (!e) { scoped_ns t1(g_ht.ns_fp); mi355::graph_fingerprint(g, c->fp_tmp); }
    const auto t_plan0 = std::chrono::steady_clock::now();
    const bool by_pointer = e != nullptr;
    if (!e) for (graph_entry * x : c->graphs) if (x->plan.fast_ok && mi355::fingerprint_equal(x->fp, c->fp_tmp)) { e = x; ++c->n_fp_hit; break; }
    if (!e) {
        mi355::plan_ctx pc = { c, plan_qkv_scratch, plan_split_scratch, c->d_dyn, c->split_min, c->attn_mfma, c->fuse, c->qkv_epi ? c->rope_tab : nullptr, plan_same_bytes, c->use_ss ? c->ss_buf : nullptr };
        mi355::plan p;
        mi355::planner(g, pc).build(p);
        ++c->n_plan;
        for (graph_entry * x : c->graphs) if (mi355::plan_equal(x->plan, p)) { e = x; break; }
        if (e) { e->fp = c->fp_tmp; e->plan.i_kcell = p.i_kcell; e->plan.i_kview = p.i_kview; e->plan.fast_ok = p.fast_ok; }
        else {
            if (c->graphs.size() >= 16) {                        // evict the least recently used entry
                size_t lru = 0;
                for (size_t i = 1; i < c->graphs.size(); ++i) if (c->graphs[i]->last_use < c->graphs[lru]->last_use) lru = i;
                dsync(c->stream);                                // its hipGraph may still be in flight
                if (c->graphs[lru]->exec) pm355_graph_free(c->graphs[lru]->exec);
                for (size_t q = c->by_ptr.size(); q-- > 0;) if (c->by_ptr[q].e == c->graphs[lru]) c->by_ptr.erase(c->by_ptr.begin() + q);
                delete c->graphs[lru];
                c->graphs.erase(c->graphs.begin() + lru);
            }
            e = new graph_entry;
            e->fp = c->fp_tmp; e->plan = std::move(p);
            c->graphs.push_back(e);
        }
    }
    e->last_use = c->tick;
    if (!by_pointer) {                                   // remember which entry this graph object lowered to (used only while the host vouches for its objects)
        bool seen = false;
        for (auto & q : c->by_ptr) if (q.g == g) { q.n_nodes = n_nodes; q.e = e; seen = true; break; }
        if (!seen) { if (c->by_ptr.size() >= 32) c->by_ptr.erase(c->by_ptr.begin()); c->by_ptr.push_back({g, n_nodes, e}); }
    }
    const auto t_dyn0 = std::chrono::steady_clock::now();
    g_ht.ns_plan += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(t_dyn0 - t_plan0).count();
    const mi355::plan & p = e->plan;
    int32_t cell = 0, n_kv = 0;
    if (!mi355::plan_dyn(g, p, cell, n_kv)) { fprintf(stderr, "ggml-mi355: cannot locate the KV cell of a planned graph\n"); return GGML_STATUS_FAILED; }
    if (c->debug_plan) print_plan(c, g, p, cell, n_kv);
    if (plan_only()) return GGML_STATUS_SUCCESS;

    if (p.has_attn) MI355_CHECK(pm355_set_i32x2(c->d_dyn, cell, n_kv, c->stream));
    g_ht.ns_dyn += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_dyn0).count();
    scoped_ns t_launch(g_ht.ns_launch);
    const bool graphable = c->use_graphs && p.single_token && p.n_gemv > 0 && p.steps.size() >= 3;
    if (graphable && e->exec) {
        if (!g_snap.taken) { g_snap.taken = true; host_counters(g_snap.v); g_snap.v[11] = c->n_replay; g_snap.t0 = std::chrono::steady_clock::now(); }
        MI355_CHECK(pm355_graph_launch(e->exec, c->stream)); ++e->runs; ++c->n_replay;
        g_snap.t_last = std::chrono::steady_clock::now();
        return GGML_STATUS_SUCCESS;
    }
    if (graphable && e->runs >= 1) {
        MI355_CHECK(pm355_capture_begin(c->stream));
        const bool ok = run_plan(c, g, p);
        e->exec = pm355_capture_end(c->stream);
        if (!ok) return GGML_STATUS_FAILED;
        if (e->exec) { MI355_CHECK(pm355_graph_launch(e->exec, c->stream)); ++e->runs; ++c->n_capture; return GGML_STATUS_SUCCESS; }
        fprintf(stderr, "ggml-mi355: hipGraph capture failed (%s) - running eagerly from now on\n", pm355_last_error());
        c->use_graphs = false;
    }
    ++e->runs; ++c->n_eager;
    return run_plan(c, g, p) ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
}

bool backend_cpy_async(ggml_backend_t bs, ggml_backend_t bd, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    if (!ggml_backend_is_mi355(bs) || !ggml_backend_is_mi355(bd)) return false;
    if (!buffer_is_mi355(src->view_src ? src->view_src->buffer : src->buffer) || !buffer_is_mi355(dst->view_src ? dst->view_src->buffer : dst->buffer)) return false;
    backend_ctx * cs = (backend_ctx *) bs->context, * cd = (backend_ctx *) bd->context;
    if (cs->device != cd->device || src->type != dst->type || !ggml_is_contiguous(src) || !ggml_is_contiguous(dst)) return false;
    dsetdev(cd->device);
    drain_all_uploads();
    order_after_null_stream(cd);
    if (cs != cd) dsync(cs->stream);
    MI355_CHECK(d2d(dst->data, src->data, hbm_bytes(src), cd->stream));
    return true;
}
void backend_event_record(ggml_backend_t b, ggml_backend_event_t e) { if (plan_only()) return; MI355_CHECK(pm355_event_record((pm355_event_t) e->context, ((backend_ctx *) b->context)->stream)); }
void backend_event_wait(ggml_backend_t b, ggml_backend_event_t e) { if (plan_only()) return; MI355_CHECK(pm355_event_wait(((backend_ctx *) b->context)->stream, (pm355_event_t) e->context)); }

const struct ggml_backend_i backend_iface = {
    /* .get_name                = */ backend_name,
    /* .free                    = */ backend_free,
    /* .get_default_buffer_type = */ [](ggml_backend_t b) { return ggml_backend_mi355_buffer_type(((backend_ctx *) b->context)->device); },
    /* .set_tensor_async        = */ backend_set_async,
    /* .get_tensor_async        = */ backend_get_async,
    /* .cpy_tensor_async        = */ backend_cpy_async,
    /* .synchronize             = */ backend_sync,
    /* .graph_plan_create       = */ nullptr,
    /* .graph_plan_free         = */ nullptr,
    /* .graph_plan_update       = */ nullptr,
    /* .graph_plan_compute      = */ nullptr,
    /* .graph_compute           = */ backend_graph_compute,
    /* .supports_op             = */ nullptr,
    /* .supports_buft           = */ nullptr,
    /* .offload_op              = */ nullptr,
    /* .event_record            = */ backend_event_record,
    /* .event_wait              = */ backend_event_wait,
};

ggml_guid_t backend_guid() {
    static ggml_guid guid = {0x6d, 0x69, 0x33, 0x35, 0x35, 0x78, 0x2d, 0x67, 0x67, 0x6d, 0x6c, 0x2d, 0x70, 0x6d, 0x33, 0x35};
    return &guid;
}

// ------------------------------------------------------------------------------------------------ device
const char * dev_name(ggml_backend_dev_t d) { return ((dev_ctx *) d->context)->name.c_str(); }
const char * dev_desc(ggml_backend_dev_t d) { return ((dev_ctx *) d->context)->desc.c_str(); }
void dev_memory(ggml_backend_dev_t d, size_t * f, size_t * t) { ggml_backend_mi355_get_device_memory(((dev_ctx *) d->context)->device, f, t); }
enum ggml_backend_dev_type dev_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU_FULL; }
void dev_props(ggml_backend_dev_t d, struct ggml_backend_dev_props * p) {
    p->name = dev_name(d); p->description = dev_desc(d); p->type = dev_type(d);
    dev_memory(d, &p->memory_free, &p->memory_total);
    p->caps = {/* async */ true, /* host_buffer */ true, /* buffer_from_host_ptr */ false, /* events */ true};
}
ggml_backend_t dev_init(ggml_backend_dev_t d, const char *) { return ggml_backend_mi355_init(((dev_ctx *) d->context)->device); }
ggml_backend_buffer_type_t dev_buft(ggml_backend_dev_t d) { return ggml_backend_mi355_buffer_type(((dev_ctx *) d->context)->device); }
ggml_backend_buffer_type_t dev_host_buft(ggml_backend_dev_t) { return ggml_backend_mi355_host_buffer_type(); }
bool dev_supports_op(ggml_backend_dev_t, const struct ggml_tensor * op) { return supports_op_impl(op); }
bool dev_supports_buft(ggml_backend_dev_t d, ggml_backend_buffer_type_t t) {
    return buft_is_mi355(t) && ((buft_ctx *) t->context)->device == ((dev_ctx *) d->context)->device;
}
bool dev_offload_op(ggml_backend_dev_t, const struct ggml_tensor * op) {
    // weights left in host memory: worth shipping to the GPU only for batched work (cf. ggml-cuda.cu:3201-3208)
    // GGML_MI355_OFFLOAD=0: never (the parity tests' reference runs at -ngl 0 must be the host's arithmetic for prompt batches too, also on a machine with a GPU)
    static const bool off = [] { const char * e = getenv("GGML_MI355_OFFLOAD"); return e && e[0] == '0'; }();
    return !off && op->op != GGML_OP_GET_ROWS && op->ne[1] >= 32;
}
ggml_backend_event_t dev_event_new(ggml_backend_dev_t d) {
    dsetdev(((dev_ctx *) d->context)->device);
    pm355_event_t e = plan_only() ? (pm355_event_t) 1 : pm355_event_create();
    if (!e) return nullptr;
    return new ggml_backend_event{d, e};
}
void dev_event_free(ggml_backend_dev_t, ggml_backend_event_t e) { if (!plan_only()) pm355_event_destroy((pm355_event_t) e->context); delete e; }
void dev_event_sync(ggml_backend_dev_t, ggml_backend_event_t e) { if (plan_only()) return; MI355_CHECK(pm355_event_sync((pm355_event_t) e->context)); }

const struct ggml_backend_device_i device_iface = {
    /* .get_name             = */ dev_name,
    /* .get_description      = */ dev_desc,
    /* .get_memory           = */ dev_memory,
    /* .get_type             = */ dev_type,
    /* .get_props            = */ dev_props,
    /* .init_backend         = */ dev_init,
    /* .get_buffer_type      = */ dev_buft,
    /* .get_host_buffer_type = */ dev_host_buft,
    /* .buffer_from_host_ptr = */ nullptr,
    /* .supports_op          = */ dev_supports_op,
    /* .supports_buft        = */ dev_supports_buft,
    /* .offload_op           = */ dev_offload_op,
    /* .event_new            = */ dev_event_new,
    /* .event_free           = */ dev_event_free,
    /* .event_synchronize    = */ dev_event_sync,
};

// ------------------------------------------------------------------------------------------------ registry
struct reg_ctx { std::vector<ggml_backend_dev_t> devices; };
const char * reg_name(ggml_backend_reg_t) { return GGML_MI355_NAME; }
size_t reg_dev_count(ggml_backend_reg_t r) { return ((reg_ctx *) r->context)->devices.size(); }
ggml_backend_dev_t reg_dev_get(ggml_backend_reg_t r, size_t i) { reg_ctx * c = (reg_ctx *) r->context; GGML_ASSERT(i < c->devices.size()); return c->devices[i]; }
extern "C" void ggml_backend_mi355_set_graph_reused(ggml_backend_t b, int reused);
// no split buffers, no n_threads (llama.cpp:3772, :21261); the one name served is the graph-reuse patch's hint
void * reg_proc(ggml_backend_reg_t, const char * name) {
    if (name && strcmp(name, "ggml_backend_mi355_set_graph_reused") == 0) return (void *) ggml_backend_mi355_set_graph_reused;
    return nullptr;
}
const struct ggml_backend_reg_i reg_iface = { reg_name, reg_dev_count, reg_dev_get, reg_proc };

} // namespace

extern "C" {

// Hint of a host that keeps its graphs between tokens (the graph-reuse patch of INTEGRATION.md section 2b calls it before every compute): reused != 0 - the ggml_cgraph
// objects passed to graph_compute until the next call are the un-rebuilt objects of the previous token (only the KV-store views moved), so the plug-in may
// look its plan up by graph pointer instead of fingerprinting every node. reused == 0 (or never called): every graph is fingerprinted.
void ggml_backend_mi355_set_graph_reused(ggml_backend_t b, int reused) {
    if (!b || !ggml_backend_is_mi355(b)) return;
    ((backend_ctx *) b->context)->host_reuses = reused != 0;
}

int ggml_backend_mi355_get_device_count(void) {
    if (plan_only()) {                                          // (GGML_MI355_PLAN_DEVICES=<n>: several pretend devices - libllama then builds its pipeline-parallel scheduler)
        static const int n = [] { const char * e = getenv("GGML_MI355_PLAN_DEVICES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > GGML_MI355_MAX_DEVICES ? GGML_MI355_MAX_DEVICES : v; }();
        return n;
    }
    int n = pm355_device_count(); return n > GGML_MI355_MAX_DEVICES ? GGML_MI355_MAX_DEVICES : n;
}

void ggml_backend_mi355_get_device_memory(int device, size_t * free, size_t * total) {
    size_t f = 0, t = 0;
    if (plan_only()) f = t = (size_t) 64 << 30;
    else pm355_device_info(device, nullptr, 0, &f, &t, nullptr);
    if (free) *free = f;
    if (total) *total = t;
}

ggml_backend_buffer_type_t ggml_backend_mi355_buffer_type(int device) {
    static std::mutex mu;
    static struct ggml_backend_buffer_type types[GGML_MI355_MAX_DEVICES];
    static bool init = false;
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= ggml_backend_mi355_get_device_count()) return nullptr;
    if (!init) {
        ggml_backend_reg_t reg = ggml_backend_mi355_reg();
        for (int i = 0; i < ggml_backend_mi355_get_device_count(); ++i)
            types[i] = { buft_iface, ggml_backend_reg_dev_get(reg, i), new buft_ctx{i, std::string(GGML_MI355_NAME "X") + std::to_string(i)} };
        init = true;
    }
    return &types[device];                           // stable singleton per device: used as a map key by llama.cpp
}

ggml_backend_buffer_type_t ggml_backend_mi355_host_buffer_type(void) {
    static struct ggml_backend_buffer_type t = {
        { host_buft_name, host_buft_alloc, ggml_backend_cpu_buffer_type()->iface.get_alignment, nullptr,
          ggml_backend_cpu_buffer_type()->iface.get_alloc_size, ggml_backend_cpu_buffer_type()->iface.is_host },
        nullptr, nullptr };
    if (!t.device && ggml_backend_mi355_get_device_count() > 0) t.device = ggml_backend_reg_dev_get(ggml_backend_mi355_reg(), 0);
    return &t;
}

ggml_backend_t ggml_backend_mi355_init(int device) {
    if (device < 0 || device >= ggml_backend_mi355_get_device_count()) { fprintf(stderr, "ggml-mi355: invalid device %d\n", device); return nullptr; }
    if (dsetdev(device)) return nullptr;
    backend_ctx * c = new backend_ctx{device, std::string(GGML_MI355_NAME "X") + std::to_string(device), plan_only() ? (pm355_stream_t) 1 : pm355_stream_create()};
    ++g_n_backends;
    if (!c->stream) { delete c; return nullptr; }
    c->fuse = !env_on("GGML_MI355_NO_FUSE");                          // node-by-node kernels only (debug / A-B)
    if (const char * am = getenv("GGML_MI355_ATTN_MFMA")) if (am[0] == '0') c->attn_mfma = false;
    if (const char * qe = getenv("GGML_MI355_QKV_EPI")) if (qe[0] == '0') c->qkv_epi = false;   // rope + KV store inside the attention kernel (round-2 form)
    if (const char * se = getenv("GGML_MI355_SS")) c->use_ss = se[0] == '1' && (plan_only() || pm355_experiments_built());   // (the partials' kernels live in the experiments library)                    // default: every rms_norm prologue reduces its own row
    c->use_graphs = !env_on("GGML_MI355_NO_GRAPH") && !plan_only();   // no hipGraph capture / replay
    c->debug_plan = env_on("GGML_MI355_DEBUG_PLAN") || plan_only();
    if (const char * sm = getenv("GGML_MI355_ATTN_SPLIT_MIN")) if (sm[0]) c->split_min = atoi(sm);
    if (c->split_min < 32) c->split_min = 32;
    return new ggml_backend{ backend_guid(), backend_iface, ggml_backend_reg_dev_get(ggml_backend_mi355_reg(), device), c };
}

int ggml_backend_is_mi355(ggml_backend_t b) { return b != nullptr && ggml_guid_matches(b->guid, backend_guid()); }

ggml_backend_reg_t ggml_backend_mi355_reg(void) {
    static struct ggml_backend_reg reg;
    static bool init = false;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!init) {
        reg_ctx * rc = new reg_ctx;
        reg = { reg_iface, rc };
        const int n = ggml_backend_mi355_get_device_count();
        for (int i = 0; i < n; ++i) {
            char nm[256] = "plan-only pseudo device (no GPU, no compute)";
            if (!plan_only()) pm355_device_info(i, nm, sizeof(nm), nullptr, nullptr, nullptr);
            dev_ctx * dc = new dev_ctx{i, std::string(GGML_MI355_NAME "X") + std::to_string(i), nm};
            rc->devices.push_back(new ggml_backend_device{ device_iface, &reg, dc });
        }
        init = true;
    }
    return &reg;
}

} // extern "C"
