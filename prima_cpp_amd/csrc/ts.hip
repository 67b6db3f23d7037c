// ts.hip — seam anatomy, MEASUREMENT BUILDS ONLY (compiled to nothing unless -DPM_TS; tools/seam_anatomy.py builds ab/ts.so with it).
// Hands every instrumented launch (mat-vec, single-token attention) a slot of PM_TS_WGS x 8 timestamps; see pm355_device.h.
#include "pm355_device.h"
#ifdef PM_TS
#include <atomic>

static unsigned long long * g_ts_base = nullptr;
static int g_ts_slots = 0;
static std::atomic<int> g_ts_next{0};

unsigned long long * pm_ts_next_slot() {
    if (!g_ts_base) return nullptr;
    const int s = g_ts_next.fetch_add(1);
    return s < g_ts_slots ? g_ts_base + (size_t) s * PM_TS_WGS * 8 : nullptr;     // launches beyond the buffer go un-instrumented
}

extern "C" {
// n_slots launches can be recorded; returns 0. pm355_ts_reset(): the next launch gets slot 0 again (call before capturing a graph).
__attribute__((visibility("default"))) int pm355_ts_enable(int n_slots) {
    if (g_ts_base) { (void) hipFree(g_ts_base); g_ts_base = nullptr; }
    g_ts_slots = 0; g_ts_next = 0;
    if (n_slots <= 0) return 0;
    const size_t nb = (size_t) n_slots * PM_TS_WGS * 8 * sizeof(unsigned long long);
    if (hipMalloc((void **) &g_ts_base, nb) != hipSuccess) { g_ts_base = nullptr; return -1; }
    (void) hipMemset(g_ts_base, 0, nb);
    g_ts_slots = n_slots;
    return 0;
}
__attribute__((visibility("default"))) int pm355_ts_reset(void) { const int n = g_ts_next.exchange(0); return n; }
__attribute__((visibility("default"))) int pm355_ts_used(void) { const int n = g_ts_next.load(); return n < g_ts_slots ? n : g_ts_slots; }
// copies the first n_slots slots to host memory (n_slots * PM_TS_WGS * 8 u64); synchronizes the device
__attribute__((visibility("default"))) int pm355_ts_read(unsigned long long * host, int n_slots) {
    if (!g_ts_base || n_slots > g_ts_slots) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpy(host, g_ts_base, (size_t) n_slots * PM_TS_WGS * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
}
#endif
