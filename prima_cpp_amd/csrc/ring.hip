// ring.hip — the piped-ring transport in C over RCCL (C ABI part (C) of include/prima_mi355.h).
//
// Replaces llama_send_tensors / llama_recv_tensors (src/llama.cpp:18031-18077: ZeroMQ multipart messages carrying the window's
// output activation + positions) and the D2H / H2D bounce around them in llama_decode_internal's ring loop
// (src/llama.cpp:18503-18564, :17306-17310). One rank per MI355X; the activation row(s) stay in HBM and travel neighbour to
// neighbour with ncclSend / ncclRecv over xGMI. Per micro-step ONE grouped exchange (ncclGroupStart ... ncclGroupEnd: send this
// step's output to the next rank, receive the next step's input from the previous rank) is enqueued on a dedicated communication
// stream; compute stream and communication stream hand over with HIP events, so the host never waits inside the token loop.
//
// RCCL is resolved at run time (dlopen("librccl.so.1")): the library keeps loading on machines without RCCL, and inside a PyTorch
// process the loader hands back the copy torch already mapped (same SONAME), never a second one.
#include "../../include/prima_mi355.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <mutex>

namespace {

// the slice of rccl.h this file needs (rccl/rccl.h:40-43, :187, :220, :260, :339, :700, :722, :923, :933)
typedef struct ncclComm * ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7 };
struct Rccl {
    void * h = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char * (*GetErrorString)(int) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok = false;
};
static void rccl_load(Rccl & r) {
    for (const char * name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.h) break;
    }
    if (!r.h) return;
    r.GetUniqueId = (decltype(r.GetUniqueId)) dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank)) dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy)) dlsym(r.h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString)) dlsym(r.h, "ncclGetErrorString");
    r.Send = (decltype(r.Send)) dlsym(r.h, "ncclSend");
    r.Recv = (decltype(r.Recv)) dlsym(r.h, "ncclRecv");
    r.GroupStart = (decltype(r.GroupStart)) dlsym(r.h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd)) dlsym(r.h, "ncclGroupEnd");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
}
Rccl & rccl() {                              // resolved once, whoever comes first (a launcher may run one rank per thread)
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(r); });
    return r;
}

thread_local char g_ring_err[256] = "";
int rfail(int code, const char * what, int nccl_rc = 0) {
    Rccl & R = rccl();
    snprintf(g_ring_err, sizeof(g_ring_err), "%s%s%s", what, nccl_rc ? ": " : "", nccl_rc && R.GetErrorString ? R.GetErrorString(nccl_rc) : "");
    return code;
}

} // namespace

struct pm355_ring {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, next = 0, prev = 0;
    hipStream_t cs = nullptr;                 // communication stream
    hipEvent_t ready = nullptr;               // compute -> comm: the buffer to send is complete
    hipEvent_t done = nullptr;                // comm -> compute: the exchange (send left, input arrived) is complete
    hipEvent_t done2[2] = {nullptr, nullptr}; // staggered loop with two sequences per rank in flight: exchange of micro-step m signals done2[m % 2]
    int done_slot = -1;                       // >= 0: pm355_ring_exchange2 records done2[done_slot] instead of `done`
    bool pending = false;                     // an exchange was enqueued since the last wait
    bool pending2[2] = {false, false};
    // caller-supplied transport (pm355_ring_init_cb) instead of RCCL
    pm355_ring_exchange_fn cb_exchange = nullptr; pm355_ring_wait_fn cb_wait = nullptr; void * cb_user = nullptr;
    // prompt pipeline / single-stream buffers: [2] inputs + [2] outputs of up to buf_floats f32
    float * pin[2] = {nullptr, nullptr}, * pout[2] = {nullptr, nullptr}; size_t buf_floats = 0;
    // staggered multi-sequence decode (pm355_ring_decode_staggered): micro-step counter, output toggle, the last output (world 1: it is the next
    // step's input), rank 0's token slot on the device
    long stag_m = 0; int stag_k = 0; float * stag_last = nullptr; int32_t * d_cur = nullptr;
};

static int ring_buffers(pm355_ring * r, size_t n_floats) {
    if (n_floats <= r->buf_floats) return 0;
    (void) hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) { if (r->pin[i]) (void) hipFree(r->pin[i]); if (r->pout[i]) (void) hipFree(r->pout[i]); r->pin[i] = r->pout[i] = nullptr; }
    r->buf_floats = 0;
    r->stag_last = nullptr;                              // (it pointed into the rows just freed: the staggered loop starts from its forced tokens again)
    for (int i = 0; i < 2; ++i)
        if (hipMalloc((void **) &r->pin[i], n_floats * 4 + 256) != hipSuccess || hipMalloc((void **) &r->pout[i], n_floats * 4 + 256) != hipSuccess) return -1;
    r->buf_floats = n_floats;
    return 0;
}

extern "C" {

const char * pm355_ring_error(void) { return g_ring_err; }

int pm355_ring_unique_id(void * id128) {
    Rccl & R = rccl();
    if (!R.ok) return rfail(PM355_E_UNSUPPORTED, "ring: librccl.so.1 not found or incomplete");
    ncclUniqueId id;
    const int rc = R.GetUniqueId(&id);
    if (rc != ncclSuccess) return rfail(PM355_E_HIP, "ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof(id));
    return 0;
}

pm355_ring * pm355_ring_init(const void * id128, int rank, int world) {
    Rccl & R = rccl();
    if (!R.ok) { rfail(PM355_E_UNSUPPORTED, "ring: librccl.so.1 not found or incomplete"); return nullptr; }
    if (!id128 || world < 1 || rank < 0 || rank >= world) { rfail(PM355_E_RANGE, "ring_init: rank / world"); return nullptr; }
    pm355_ring * r = new pm355_ring();
    r->rank = rank; r->world = world; r->next = (rank + 1) % world; r->prev = (rank + world - 1) % world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const int rc = R.CommInitRank(&r->comm, world, id, rank);
    if (rc != ncclSuccess) { rfail(PM355_E_HIP, "ncclCommInitRank", rc); delete r; return nullptr; }
    if (hipStreamCreateWithFlags(&r->cs, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->done2[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->done2[1], hipEventDisableTiming) != hipSuccess) {
        rfail(PM355_E_HIP, "ring_init: stream / events");
        pm355_ring_free(r);
        return nullptr;
    }
    return r;
}

pm355_ring * pm355_ring_init_cb(int rank, int world, pm355_ring_exchange_fn exchange, pm355_ring_wait_fn wait, void * user) {
    if (!exchange || !wait || world < 1 || rank < 0 || rank >= world) { rfail(PM355_E_RANGE, "ring_init_cb: rank / world / callbacks"); return nullptr; }
    pm355_ring * r = new pm355_ring();
    r->rank = rank; r->world = world; r->next = (rank + 1) % world; r->prev = (rank + world - 1) % world;
    r->cb_exchange = exchange; r->cb_wait = wait; r->cb_user = user;
    return r;
}

// world size 1 without a communicator: the schedules below run on one window (no exchange is ever enqueued)
pm355_ring * pm355_ring_init_local(void) {
    pm355_ring * r = new pm355_ring();
    r->rank = 0; r->world = 1; r->next = r->prev = 0;
    return r;
}

void pm355_ring_free(pm355_ring * r) {
    if (!r) return;
    if (r->cs) { (void) hipStreamSynchronize(r->cs); }
    for (int i = 0; i < 2; ++i) { if (r->pin[i]) (void) hipFree(r->pin[i]); if (r->pout[i]) (void) hipFree(r->pout[i]); }
    if (r->d_cur) (void) hipFree(r->d_cur);
    if (r->comm) (void) rccl().CommDestroy(r->comm);
    if (r->cs) (void) hipStreamDestroy(r->cs);
    if (r->ready) (void) hipEventDestroy(r->ready);
    if (r->done) (void) hipEventDestroy(r->done);
    for (int i = 0; i < 2; ++i) if (r->done2[i]) (void) hipEventDestroy(r->done2[i]);
    delete r;
}

// One grouped exchange on the communication stream: send `send` (n floats, may be NULL) to the next rank and receive `recv`
// (n floats, may be NULL) from the previous rank. The communication stream first waits for everything enqueued so far on
// `compute_stream` (the producer of `send`); the completion is published through an event that pm355_ring_wait hands to the
// compute stream. Nothing here blocks the host.
int pm355_ring_exchange(pm355_ring * r, const float * send, float * recv, int64_t n, pm355_stream_t compute_stream) {
    return pm355_ring_exchange2(r, send, send ? n : 0, recv, recv ? n : 0, compute_stream);
}
int pm355_ring_exchange2(pm355_ring * r, const float * send, int64_t n_send, float * recv, int64_t n_recv, pm355_stream_t compute_stream) {
    if (!r) return rfail(PM355_E_SHAPE, "ring_exchange: no ring");
    if (!n_send) send = nullptr;
    if (!n_recv) recv = nullptr;
    if (!send && !recv) return 0;
    if (r->cb_exchange) {
        const int rc = r->cb_exchange(r->cb_user, send, send ? n_send : 0, recv, recv ? n_recv : 0, compute_stream);
        if (rc) return rfail(PM355_E_HIP, "ring_exchange: transport callback failed");
        if (r->done_slot >= 0) r->pending2[r->done_slot] = true; else r->pending = true;
        return 0;
    }
    Rccl & R = rccl();
    if (!R.ok) return rfail(PM355_E_SHAPE, "ring_exchange: no ring");
    hipStream_t st = (hipStream_t) compute_stream;
    if (hipEventRecord(r->ready, st) != hipSuccess || hipStreamWaitEvent(r->cs, r->ready, 0) != hipSuccess) return rfail(PM355_E_HIP, "ring_exchange: event hand-off");
    int rc = R.GroupStart();
    if (rc == ncclSuccess && send) rc = R.Send(send, (size_t) n_send, ncclFloat32, r->next, r->comm, r->cs);
    if (rc == ncclSuccess && recv) rc = R.Recv(recv, (size_t) n_recv, ncclFloat32, r->prev, r->comm, r->cs);
    const int rc2 = R.GroupEnd();
    if (rc != ncclSuccess || rc2 != ncclSuccess) return rfail(PM355_E_HIP, "ring_exchange: ncclSend / ncclRecv", rc != ncclSuccess ? rc : rc2);
    if (hipEventRecord(r->done_slot >= 0 ? r->done2[r->done_slot] : r->done, r->cs) != hipSuccess) return rfail(PM355_E_HIP, "ring_exchange: event record");
    if (r->done_slot >= 0) r->pending2[r->done_slot] = true; else r->pending = true;
    return 0;
}

// the compute stream waits for the exchange that signalled done2[slot] (two-deep staggered loop). A caller-supplied transport knows only "everything
// enqueued so far" - its wait also covers the younger exchange, which costs the overlap and nothing else.
static int ring_wait_slot(pm355_ring * r, int slot, pm355_stream_t compute_stream) {
    if (!r->pending2[slot]) return 0;
    if (r->cb_wait) {
        if (r->cb_wait(r->cb_user, compute_stream)) return rfail(PM355_E_HIP, "ring_wait: transport callback failed");
        r->pending2[0] = r->pending2[1] = false;
        return 0;
    }
    if (hipStreamWaitEvent((hipStream_t) compute_stream, r->done2[slot], 0) != hipSuccess) return rfail(PM355_E_HIP, "ring_wait");
    r->pending2[slot] = false;
    return 0;
}

// The compute stream waits (on the device) for the last exchange: the previous send has left its buffer, the input has arrived.
// (also for everything a two-deep staggered loop still has in flight)
int pm355_ring_wait(pm355_ring * r, pm355_stream_t compute_stream) {
    if (!r) return rfail(PM355_E_SHAPE, "ring_wait: no ring");
    for (int sl = 0; sl < 2; ++sl) { const int rc = ring_wait_slot(r, sl, compute_stream); if (rc) return rc; }
    if (!r->pending) return 0;
    if (r->cb_wait) {
        if (r->cb_wait(r->cb_user, compute_stream)) return rfail(PM355_E_HIP, "ring_wait: transport callback failed");
        r->pending = false;
        return 0;
    }
    if (hipStreamWaitEvent((hipStream_t) compute_stream, r->done, 0) != hipSuccess) return rfail(PM355_E_HIP, "ring_wait");
    r->pending = false;
    return 0;
}

// One micro-step of a rank in the body of the reference's ring loop (src/llama.cpp:18503-18564), all enqueued, no host wait:
//   wait for the previous exchange (this step's input) -> window step (pm355_model_step_ex: [head on x_in] -> embed -> layers) ->
//   grouped exchange {send x_out to the next rank, receive the NEXT step's input into recv_next}.
// x_in may be NULL (rank 0 while the pipeline fills / forced tokens); send / recv_next NULL = nothing to send / receive.
int pm355_ring_step(pm355_ring * r, pm355_model * m, const int32_t * d_token, const float * x_in, float * x_out, float * d_logits,
                    int32_t * d_argmax, int advance, int rotate, int head_first, int use_graph, int do_send, float * recv_next,
                    int64_t n_embd, pm355_stream_t compute_stream) {
    int rc = pm355_ring_wait(r, compute_stream);
    if (rc) return rc;
    rc = pm355_model_step_ex(m, d_token, x_in, x_out, d_logits, d_argmax, advance, rotate, head_first, use_graph, compute_stream);
    if (rc) return rfail(rc, pm355_model_error(m));
    return pm355_ring_exchange(r, do_send ? x_out : nullptr, recv_next, n_embd, compute_stream);
}

int pm355_ring_step_tokens(pm355_ring * r, pm355_model * m, int seq, const int32_t * d_tokens, const float * x_in, float * x_out, int n_tokens,
                           int pos0, const float * send_ptr, int64_t n_send, float * recv_next, int64_t n_recv, pm355_stream_t compute_stream) {
    int rc = pm355_ring_wait(r, compute_stream);
    if (rc) return rc;
    rc = pm355_model_decode_seq(m, seq, d_tokens, d_tokens ? nullptr : x_in, n_tokens, pos0, x_out, nullptr, nullptr, compute_stream);
    if (rc) return rfail(rc, pm355_model_error(m));
    return pm355_ring_exchange2(r, send_ptr, n_send, recv_next, n_recv, compute_stream);
}

int pm355_ring_prefill(pm355_ring * r, pm355_model * m, int n_seq, const int32_t * d_tokens, int n_prompt, int ubatch, float * final_rows,
                       pm355_stream_t compute_stream) {
    if (!r || !m || n_seq < 1 || n_prompt < 1 || ubatch < 1) return rfail(PM355_E_RANGE, "ring_prefill: arguments");
    const int W = r->world, rank = r->rank;
    const int64_t E = pm355_model_n_embd(m);
    const int C = (n_prompt + ubatch - 1) / ubatch, G = n_seq * C;
    if (rank == 0 && (!d_tokens || !final_rows)) return rfail(PM355_E_SHAPE, "ring_prefill: rank 0 needs the tokens and final_rows");
    if (ring_buffers(r, (size_t) ubatch * E)) return rfail(PM355_E_NOMEM, "ring_prefill: activation buffers");
    hipStream_t st = (hipStream_t) compute_stream;
    auto chunk_len = [&](int g) { const int c = g % C; return c == C - 1 ? n_prompt - c * ubatch : ubatch; };
    if (W == 1) {                                             // no hand-off: chunk after chunk, keep each prompt's last row
        for (int g = 0; g < G; ++g) {
            const int seq = g / C, c = g % C, T = chunk_len(g);
            int rc = pm355_model_decode_seq(m, seq, d_tokens + (size_t) seq * n_prompt + (size_t) c * ubatch, nullptr, T, c * ubatch, r->pout[0], nullptr, nullptr, compute_stream);
            if (rc) return rfail(rc, pm355_model_error(m));
            if (c == C - 1 && hipMemcpyAsync(final_rows + (size_t) seq * E, r->pout[0] + (size_t) (T - 1) * E, E * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return rfail(PM355_E_HIP, "ring_prefill: copy of the final row");
        }
        for (int seq = 0; seq < n_seq; ++seq) { int rc = pm355_model_set_seq_pos(m, seq, n_prompt, compute_stream); if (rc) return rfail(rc, pm355_model_error(m)); }
        return 0;
    }
    for (int s = 0; s <= G + W - 2; ++s) {
        const int g = s - rank;
        const bool valid = g >= 0 && g < G;
        int rc = pm355_ring_wait(r, compute_stream);          // the previous step's exchange: this chunk's input is here, the last output has left
        if (rc) return rc;
        float * out = r->pout[g & 1];
        int T = 0;
        if (valid) {
            const int seq = g / C, c = g % C;
            T = chunk_len(g);
            rc = pm355_model_decode_seq(m, seq, rank == 0 ? d_tokens + (size_t) seq * n_prompt + (size_t) c * ubatch : nullptr,
                                        rank == 0 ? nullptr : r->pin[g & 1], T, c * ubatch, out, nullptr, nullptr, compute_stream);
            if (rc) return rfail(rc, pm355_model_error(m));
        }
        // what leaves: a window's whole ubatch to the next rank; from the last rank only the last row of a prompt's final chunk, to rank 0
        const float * snd = nullptr; int64_t n_snd = 0;
        if (valid) {
            if (rank < W - 1) { snd = out; n_snd = (int64_t) T * E; }
            else if (g % C == C - 1) { snd = out + (size_t) (T - 1) * E; n_snd = E; }
        }
        // what arrives for the NEXT step: rank > 0 its next chunk; rank 0 the row the last rank sends at the end of THIS step
        float * rcv = nullptr; int64_t n_rcv = 0;
        if (rank > 0) {
            const int gn = g + 1;
            if (gn >= 0 && gn < G) { rcv = r->pin[gn & 1]; n_rcv = (int64_t) chunk_len(gn) * E; }
        } else {
            const int gl = s - (W - 1);                       // the chunk the last rank is working on in this step
            if (gl >= 0 && gl < G && gl % C == C - 1) { rcv = final_rows + (size_t) (gl / C) * E; n_rcv = E; }
        }
        rc = pm355_ring_exchange2(r, snd, n_snd, rcv, n_rcv, compute_stream);
        if (rc) return rc;
    }
    for (int seq = 0; seq < n_seq; ++seq) { int rc = pm355_model_set_seq_pos(m, seq, n_prompt, compute_stream); if (rc) return rfail(rc, pm355_model_error(m)); }
    return 0;
}

int pm355_ring_single_token(pm355_ring * r, pm355_model * m, int seq, int32_t * d_token, float * d_logits, pm355_stream_t compute_stream) {
    if (!r || !m) return rfail(PM355_E_SHAPE, "ring_single_token: no ring / window");
    const int W = r->world, rank = r->rank;
    const int64_t E = pm355_model_n_embd(m);
    int rc = pm355_model_set_seq(m, seq, compute_stream);
    if (rc) return rfail(rc, pm355_model_error(m));
    if (W == 1) {
        rc = pm355_model_step_ex(m, d_token, nullptr, nullptr, d_logits, d_token, 1, 0, 0, 1, compute_stream);
        return rc ? rfail(rc, pm355_model_error(m)) : 0;
    }
    if (ring_buffers(r, (size_t) E)) return rfail(PM355_E_NOMEM, "ring_single_token: buffers");
    if (rank == 0) {
        if (!d_token) return rfail(PM355_E_SHAPE, "ring_single_token: rank 0 needs d_token");
        rc = pm355_ring_wait(r, compute_stream);
        if (!rc) { rc = pm355_model_step_ex(m, d_token, nullptr, r->pout[0], nullptr, nullptr, 1, 0, 0, 1, compute_stream); if (rc) return rfail(rc, pm355_model_error(m)); }
        if (!rc) rc = pm355_ring_exchange2(r, r->pout[0], E, r->pin[0], E, compute_stream);      // out to rank 1; the last rank's row comes back
        if (!rc) rc = pm355_ring_wait(r, compute_stream);
        if (rc) return rc;
        rc = pm355_model_head(m, r->pin[0], d_logits, d_token, compute_stream);
        return rc ? rfail(rc, pm355_model_error(m)) : 0;
    }
    rc = pm355_ring_wait(r, compute_stream);
    if (!rc) rc = pm355_ring_exchange2(r, nullptr, 0, r->pin[0], E, compute_stream);
    if (!rc) rc = pm355_ring_wait(r, compute_stream);
    if (rc) return rc;
    rc = pm355_model_step_ex(m, nullptr, r->pin[0], r->pout[0], nullptr, nullptr, 1, 0, 0, 1, compute_stream);
    if (rc) return rfail(rc, pm355_model_error(m));
    return pm355_ring_exchange2(r, r->pout[0], E, nullptr, 0, compute_stream);
}

// The staggered multi-sequence decode loop of the ring, in C (was RingDriver.micro_step in prima_cpp_amd/ring.py: one interpreter round trip per
// ~1 ms micro-step and rank). The reference keeps ONE batch in flight - rank 0 blocks in recv until the token has been round the ring
// (llama_decode_internal, src/llama.cpp:18503-18564) - so a layer split cannot speed it up; here `world` sequences are in flight, one rank apart:
// at micro-step m rank r works on sequence (m - r) mod world, every rank's window holds `world` KV slabs and rotates its sequence counter in the
// captured graph. Per micro-step and rank, all enqueued, no host wait:
//   wait for the previous exchange -> window step (rank 0: head on the activation the last rank returned -> argmax -> embed -> its window, ONE
//   graph; other ranks: their window on the incoming row) -> grouped exchange {send the output row on, receive the next step's input}.
// forced (host, rank 0 only, may be NULL): token id to feed at micro-step i of this call instead of the head's argmax (prompt tokens, the first
// token after a prompt pass; < 0 = none); mandatory while no activation has come back yet (the first `world` micro-steps). d_tokens_out (device,
// rank 0, may be NULL): slot i receives the token fed at micro-step i of this call. reset != 0: the schedule starts again at micro-step 0.
int pm355_ring_decode_staggered(pm355_ring * r, pm355_model * m, int n_micro, const int32_t * forced, int32_t * d_tokens_out, int reset, int use_graph,
                                pm355_stream_t compute_stream) {
    if (!r || !m || n_micro < 0) return rfail(PM355_E_SHAPE, "ring_decode_staggered: arguments");
    const int W = r->world, rank = r->rank;
    const int64_t E = pm355_model_n_embd(m);
    hipStream_t st = (hipStream_t) compute_stream;
    if (ring_buffers(r, (size_t) E)) return rfail(PM355_E_NOMEM, "ring_decode_staggered: buffers");
    if (!r->d_cur && hipMalloc((void **) &r->d_cur, 64) != hipSuccess) return rfail(PM355_E_NOMEM, "ring_decode_staggered: token slot");
    if (reset) {
        // the schedule restarts at micro-step 0 = sequence 0 on EVERY rank: the model's device-side sequence counter (it selects the KV slab and advances
        // with every step) restarts with it - after n micro-steps the ranks sit at different sequence indices
        r->stag_m = 0; r->stag_k = 0; r->stag_last = nullptr;
        const int rs = pm355_model_set_seq(m, 0, compute_stream);
        if (rs) return rfail(rs, pm355_model_error(m));
    }
    // D = sequences per rank in flight = hop latency in micro-steps: a window finalized for n_seq = 2 x world runs TWO interleaved rounds of sequences - the row
    // a rank sends after micro-step m is consumed by its successor at micro-step m + 2, so the exchange of step m travels while step m + 1 (the other round's
    // sequence) computes and no hop is exposed (the reference has one batch in flight, src/llama.cpp:18509). D = 1: the lock-step schedule.
    const int n_seq = pm355_model_n_seq(m);
    const int D = (W > 1 && n_seq == 2 * W) ? 2 : 1;
    if (n_seq != D * W && !(W == 1 && n_seq >= 1)) return rfail(PM355_E_SHAPE, "ring_decode_staggered: the window must be finalized for world or 2 x world sequences");
    auto need_recv = [&](long mm) { return W > 1 && (rank == 0 ? mm >= (long) D * W : mm >= (long) D * rank); };
    for (int i = 0; i < n_micro; ++i) {
        const long mm = r->stag_m++;
        const bool active = mm >= (long) D * rank;
        // the exchange enqueued D micro-steps ago: our row of then has left its buffer, this step's input is here
        int rc = D == 1 ? pm355_ring_wait(r, compute_stream) : ring_wait_slot(r, (int) (mm & 1), compute_stream);
        if (rc) return rc;
        float * out = nullptr;
        if (active) {
            r->stag_k ^= 1;
            out = r->pout[r->stag_k];
            const float * x_in = W == 1 ? r->stag_last : (need_recv(mm) ? r->pin[mm & 1] : nullptr);
            if (rank == 0) {
                const bool have_forced = forced && forced[i] >= 0;
                if (have_forced || !x_in) {
                    if (!have_forced && mm < (long) D * W) return rfail(PM355_E_SHAPE, "ring_decode_staggered: the first micro-steps of rank 0 (one per sequence in flight) need forced tokens");
                    if (have_forced && pm355_set_i32x2(r->d_cur, forced[i], 0, compute_stream)) return rfail(PM355_E_HIP, "ring_decode_staggered: token upload");
                    rc = pm355_model_step_ex(m, r->d_cur, nullptr, out, nullptr, nullptr, 1, 1, 0, use_graph, compute_stream);
                } else {
                    rc = pm355_model_step_ex(m, r->d_cur, x_in, out, nullptr, r->d_cur, 1, 1, 1, use_graph, compute_stream);     // head -> argmax -> embed -> window
                }
                if (!rc && d_tokens_out && hipMemcpyAsync(d_tokens_out + i, r->d_cur, 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return rfail(PM355_E_HIP, "ring_decode_staggered: token copy");
            } else {
                rc = pm355_model_step_ex(m, nullptr, x_in, out, nullptr, nullptr, 1, 1, 0, use_graph, compute_stream);
            }
            if (rc) return rfail(rc, pm355_model_error(m));
            r->stag_last = out;
        }
        if (W > 1) {
            // send this step's row, receive the input of step mm + D (same buffer parity as this step's input, which the window has just consumed)
            r->done_slot = D == 1 ? -1 : (int) (mm & 1);
            rc = pm355_ring_exchange2(r, active ? out : nullptr, E, need_recv(mm + D) ? r->pin[(mm + D) & 1] : nullptr, E, compute_stream);
            r->done_slot = -1;
            if (rc) return rc;
        }
    }
    return 0;
}
// the last output row of this rank's staggered loop (device pointer; NULL before the first active micro-step)
const float * pm355_ring_decode_last_output(const pm355_ring * r) { return r ? r->stag_last : nullptr; }

int pm355_ring_rank(const pm355_ring * r) { return r ? r->rank : -1; }
int pm355_ring_world(const pm355_ring * r) { return r ? r->world : 0; }

} // extern "C"
