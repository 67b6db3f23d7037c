// probe.hip — measurement helpers (not on the product path): the HBM read ceiling this chip/box actually delivers for
// the access pattern of the mat-vec (every wave streams its own contiguous span with 16-byte non-temporal loads, one
// 1024-thread workgroup per CU), reported next to the 8 TB/s spec peak by bench.py ("confirm on the box").
#include "pm355_device.h"
#include "pm355_probe.h"

namespace {

template <int UNROLL, bool NT = true>
__global__ __launch_bounds__(1024) void stream_read_kernel(const uint8_t * __restrict__ src, long bytes_per_wave, uint32_t * sink) {
    const int lane = threadIdx.x & 63;
    const long wave = (long) blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t * p = src + wave * bytes_per_wave;
    u32x4 acc = {0, 0, 0, 0};
    const long steps = bytes_per_wave / (1024 * UNROLL);
    for (long s = 0; s < steps; ++s) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) v[i] = NT ? ld_nt16(p + (uint32_t) (lane * 16 + i * 1024)) : *(const PM_G u32x4 *) (p + (uint32_t) (lane * 16 + i * 1024));
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) acc ^= v[i];
        p += 1024 * UNROLL;
    }
    const uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (r == 0x9E3779B9u && sink) sink[0] = r;          // practically never: keeps the loads alive
}

// Strided-chunk read (tools/access_pattern_probe.py): what HBM delivers for the access patterns of the long-context attention. A wave
// instruction reads 1 KB as 1024 / chunk pieces of `chunk` contiguous bytes, `piece_stride` apart; instruction i of a wave goes to
// (i / n_inner) * outer_stride + (i % n_inner) * inner_stride; two groups of n_inner instructions in flight per wave.
struct ChunkP { const uint8_t * src; long wgx_stride, wgy_stride, wave_stride, outer_stride, inner_stride, piece_stride; int nx, chunk, n_outer; uint32_t * sink; };
template <int NI>
__global__ __launch_bounds__(256, 2) void chunk_read_kernel(ChunkP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpc = p.chunk >> 4;                      // lanes per piece
    const uint8_t * b = p.src + (long) (blockIdx.x % p.nx) * p.wgx_stride + (long) (blockIdx.x / p.nx) * p.wgy_stride + (long) wave * p.wave_stride
                              + (long) (lane / lpc) * p.piece_stride + (lane % lpc) * 16;
    u32x4 acc = {0, 0, 0, 0}, v0[NI], v1[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) v0[k] = *(const PM_G u32x4 *) (b + k * p.inner_stride);
    for (int t = 0; t < p.n_outer; t += 2) {
        if (t + 1 < p.n_outer) {
#pragma unroll
            for (int k = 0; k < NI; ++k) v1[k] = *(const PM_G u32x4 *) (b + (t + 1) * p.outer_stride + k * p.inner_stride);
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) acc ^= v0[k];
        if (t + 1 >= p.n_outer) break;
        if (t + 2 < p.n_outer) {
#pragma unroll
            for (int k = 0; k < NI; ++k) v0[k] = *(const PM_G u32x4 *) (b + (t + 2) * p.outer_stride + k * p.inner_stride);
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) acc ^= v1[k];
    }
    const uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (r == 0x9E3779B9u && p.sink) p.sink[0] = r;
}

} // namespace

int pm_launch_chunk_read(const void * src, int n_wg, int nx, long wgx_stride, long wgy_stride, long wave_stride, long outer_stride, long inner_stride,
                         long piece_stride, int chunk, int n_outer, void * sink, hipStream_t st) {
    if (chunk < 16 || chunk > 1024 || (chunk & (chunk - 1))) return -1;
    ChunkP p = {(const uint8_t *) src, wgx_stride, wgy_stride, wave_stride, outer_stride, inner_stride, piece_stride, nx, chunk, n_outer, (uint32_t *) sink};
    hipLaunchKernelGGL(chunk_read_kernel<8>, dim3(n_wg), dim3(256), 0, st, p);
    return 0;
}

// reads `bytes` (rounded down to whole wave spans) once; grid = one 1024-thread workgroup per CU x wg_per_cu
int pm_launch_stream_read(const void * src, size_t bytes, int wg_per_cu, int unroll, void * sink, hipStream_t st) {
    hipDeviceProp_t pr; int dev = 0; (void) hipGetDevice(&dev);
    const int cus = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256;
    const int grid = wg_per_cu < 0 ? -wg_per_cu : cus * (wg_per_cu > 0 ? wg_per_cu : 1);
    const long waves = (long) grid * 16;
    long per_wave = (long) (bytes / waves);
    per_wave -= per_wave % (1024 * 8);
    if (per_wave <= 0) return -1;
    // unroll < 0: |unroll| = 8 default-policy (cached) loads instead of non-temporal ones (prefetch experiments: does a plain read leave
    // the lines in the infinity cache for a later nt read?); wg_per_cu < 0: -wg_per_cu workgroups in TOTAL (a partial grid)
    if (unroll < 0) { hipLaunchKernelGGL((stream_read_kernel<8, false>), dim3(grid), dim3(1024), 0, st, (const uint8_t *) src, per_wave, (uint32_t *) sink); return 0; }
    if (unroll == 4) hipLaunchKernelGGL(stream_read_kernel<4>, dim3(grid), dim3(1024), 0, st, (const uint8_t *) src, per_wave, (uint32_t *) sink);
    else             hipLaunchKernelGGL(stream_read_kernel<8>, dim3(grid), dim3(1024), 0, st, (const uint8_t *) src, per_wave, (uint32_t *) sink);
    return 0;
}
