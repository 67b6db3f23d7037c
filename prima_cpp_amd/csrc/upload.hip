// upload.hip — asynchronous weight staging from host DRAM: pageable (mmap'd GGUF) bytes -> pinned ring -> hipMemcpyAsync -> HBM layout.
//
// North-star row "async weight staging from host DRAM via pinned hipMemcpyAsync in place of mmap prefetch". Replaces, for tensors in
// our buffers, what the reference's loader does per weight tensor: ggml_backend_tensor_set(cur, mmap pointer, 0, n_size)
// (src/llama.cpp:5580-5600; its own pinned async-upload path, :5432-5520, is disabled in this fork: "async uploads is not supported
// now") -> a synchronous pageable cudaMemcpy in the CUDA plug-in (ggml-cuda.cu:503-511). A pageable hipMemcpy of a page-cache mapping
// runs at ~11 GB/s on the MI355X box because the runtime's own staging copy is single-threaded and not overlapped with the DMA.
// Here:
//   * NBUF pinned chunks (hipHostMalloc) form a ring; a small pool of copier threads fills chunk i (page faults + memcpy, split in
//     equal slices) while the DMA engine sends chunk i-1 (hipMemcpyAsync on a private non-blocking stream) and the repack kernel
//     re-orders chunk i-2 into the row-SoA HBM layout (repack.hip) on the same stream;
//   * the caller returns as soon as the last chunk is enqueued; pm355_uploader_sync() waits for the stream (the plug-in calls it from
//     the buffer functions that must observe the data: get_tensor, cpy, clear, and from graph_compute through the device-wide order
//     of the null stream... see ggml_backend_mi355.cpp).
#include "../../include/prima_mi355.h"
#include "pm355_kernels.h"
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <mutex>
#include <string.h>
#include <thread>
#include <vector>

namespace {

// persistent copier threads: run(fn over slices) and wait
struct CopyPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    char * dst = nullptr; const char * src = nullptr; size_t n = 0;
    int gen = 0, pending = 0; bool stop = false;
    explicit CopyPool(int nthreads) {
        for (int i = 0; i < nthreads; ++i) th.emplace_back([this, i, nthreads] { worker(i, nthreads); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; ++gen; }
        cv_go.notify_all();
        for (auto & t : th) t.join();
    }
    void worker(int i, int nt) {
        int seen = 0;
        for (;;) {
            char * d; const char * s; size_t bytes;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (stop) return;
                d = dst; s = src; bytes = n;
            }
            // slice i of nt, 4-KiB aligned boundaries (page faults stay inside one thread's slice)
            size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t) 4095;
            const size_t b = (size_t) i * per, e = b + per < bytes ? b + per : bytes;
            if (b < e) memcpy(d + b, s + b, e - b);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    void copy(char * d, const char * s, size_t bytes) {
        if (th.empty() || bytes < (1u << 20)) { memcpy(d, s, bytes); return; }
        std::unique_lock<std::mutex> lk(mu);
        dst = d; src = s; n = bytes; pending = (int) th.size(); ++gen;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

constexpr int NBUF = 3;

} // namespace

struct pm355_uploader {
    size_t chunk = 0;
    void * pin[NBUF] = {};
    hipEvent_t ev[NBUF] = {};
    void * dstage = nullptr;                 // NBUF device chunks for the repack source
    hipStream_t st = nullptr;
    int next = 0;
    CopyPool * pool = nullptr;
    uint64_t bytes = 0;
};

extern "C" {

pm355_uploader * pm355_uploader_new(size_t chunk_bytes, int copy_threads) {
    pm355_uploader * u = new pm355_uploader();
    u->chunk = chunk_bytes ? chunk_bytes : (size_t) 32 << 20;
    if (copy_threads < 0) {
        const char * e = getenv("PM355_UPLOAD_THREADS");
        copy_threads = e ? atoi(e) : 4;
    }
    if (copy_threads > 16) copy_threads = 16;
    if (hipStreamCreateWithFlags(&u->st, hipStreamNonBlocking) != hipSuccess) { delete u; return nullptr; }
    for (int i = 0; i < NBUF; ++i) {
        if (hipHostMalloc(&u->pin[i], u->chunk, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&u->ev[i], hipEventDisableTiming) != hipSuccess) { pm355_uploader_free(u); return nullptr; }
    }
    if (copy_threads > 1) u->pool = new CopyPool(copy_threads);
    return u;
}

void pm355_uploader_free(pm355_uploader * u) {
    if (!u) return;
    if (u->st) (void) hipStreamSynchronize(u->st);
    delete u->pool;
    for (int i = 0; i < NBUF; ++i) { if (u->pin[i]) (void) hipHostFree(u->pin[i]); if (u->ev[i]) (void) hipEventDestroy(u->ev[i]); }
    if (u->dstage) (void) hipFree(u->dstage);
    if (u->st) (void) hipStreamDestroy(u->st);
    delete u;
}

// host (pageable or pinned) GGUF-order bytes -> device. type < 0 or a type whose HBM layout is the GGUF layout: plain bytes;
// row-SoA types (Q4_K/Q6_K/Q8_0 matrices): `nbytes` must be whole rows of K weights, dev = first destination row in the HBM layout.
// Returns after the last chunk is ENQUEUED (host memory may be reused / unmapped: every byte has been copied into the pinned ring).
int pm355_upload(pm355_uploader * u, int type, int64_t K, const void * host, void * dev, size_t nbytes, int repack) {
    if (!u || !host || !dev) return PM355_E_SHAPE;
    (void) hipGetLastError();
    const bool rp = repack && type >= 0 && pm_type_is_repacked(type);
    const size_t rb = rp ? pm_weight_row_bytes(type, K) : 0, rs = rp ? pm_weight_row_stride(type, K) : 0;
    if (rp && (!rb || nbytes % rb)) return PM355_E_SHAPE;
    if (rp && rb > u->chunk) return PM355_E_RANGE;
    const size_t chunk = rp ? (u->chunk / rb) * rb : u->chunk;
    if (rp && !u->dstage && hipMalloc(&u->dstage, (size_t) NBUF * u->chunk) != hipSuccess) return PM355_E_NOMEM;
    size_t off = 0;
    while (off < nbytes) {
        const size_t n = nbytes - off < chunk ? nbytes - off : chunk;
        const int b = u->next;
        if (hipEventSynchronize(u->ev[b]) != hipSuccess) return PM355_E_HIP;        // previous use of this pinned chunk has left it
        if (u->pool) u->pool->copy((char *) u->pin[b], (const char *) host + off, n);
        else memcpy(u->pin[b], (const char *) host + off, n);
        if (rp) {
            char * ds = (char *) u->dstage + (size_t) b * u->chunk;
            if (hipMemcpyAsync(ds, u->pin[b], n, hipMemcpyHostToDevice, u->st) != hipSuccess) return PM355_E_HIP;
            pm_launch_repack(type, ds, (char *) dev + (off / rb) * rs, K, (int64_t) (n / rb), 1, u->st);
        } else {
            if (hipMemcpyAsync((char *) dev + off, u->pin[b], n, hipMemcpyHostToDevice, u->st) != hipSuccess) return PM355_E_HIP;
        }
        if (hipEventRecord(u->ev[b], u->st) != hipSuccess) return PM355_E_HIP;
        off += n; u->next = (b + 1) % NBUF;
    }
    u->bytes += nbytes;
    return hipGetLastError() == hipSuccess ? 0 : PM355_E_HIP;
}

int pm355_uploader_sync(pm355_uploader * u) {
    if (!u) return 0;
    return hipStreamSynchronize(u->st) == hipSuccess ? 0 : PM355_E_HIP;
}

uint64_t pm355_uploader_bytes(const pm355_uploader * u) { return u ? u->bytes : 0; }

} // extern "C"
