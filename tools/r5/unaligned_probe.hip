// does a global_load_dwordx2 / dwordx4 at a 2-byte-aligned address return the right bytes on gfx950? (Q8_0 blocks are 34 bytes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t * src, u32x2 * out2, u32x4 * out4, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t * p = src + 34 * i + 2;
    u32x2 a; u32x4 b;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(b) : "v"(p + 8) : "memory");
    out2[i] = a; out4[i] = b;
}
int main() {
    const int n = 4096;
    std::vector<uint8_t> h(34 * n + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t) (i * 131 + 7);
    uint8_t * d; u32x2 * o2; u32x4 * o4;
    hipMalloc(&d, h.size()); hipMalloc(&o2, n * 8); hipMalloc(&o4, n * 16);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o2, o4, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("UNALIGNED_PROBE fault: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<uint8_t> r2(n * 8), r4(n * 16);
    hipMemcpy(r2.data(), o2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r4.data(), o4, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) { if (memcmp(&r2[8 * i], &h[34 * i + 2], 8)) ++bad; if (memcmp(&r4[16 * i], &h[34 * i + 10], 16)) ++bad; }
    printf("UNALIGNED_PROBE bad=%d of %d\n", bad, 2 * n);
    return bad != 0;
}
