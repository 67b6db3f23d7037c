// stand-alone check of the LDS-DMA forms the decode engine relies on (hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((address_space(3))) void * lds_vp;
typedef __attribute__((address_space(1))) const void * g_vp;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(128) void k(const uint8_t * src, uint8_t * out, int base_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    char * dst = smem + base_off;
    if (wave == 1) {
        __builtin_amdgcn_global_load_lds((g_vp) (src + lane * 16), (lds_vp) dst, 16, 0, 2);                    // 1 KiB, all lanes
        __builtin_amdgcn_global_load_lds((g_vp) (src + 4096 + (63 - lane) * 16), (lds_vp) (dst + 1024), 16, 0, 2);   // gathered: lane l <- chunk 63 - l
        if (lane < 16) __builtin_amdgcn_global_load_lds((g_vp) (src + 8192 + lane * 16), (lds_vp) (dst + 2048), 16, 0, 2);   // 16 lanes
        if (lane < 8) __builtin_amdgcn_global_load_lds((g_vp) (src + 12288 + lane * 4), (lds_vp) (dst + 2304), 4, 0, 2);      // dword form, 8 lanes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned) (uintptr_t) (__attribute__((address_space(3))) unsigned *) &flag), "v"(1u) : "memory");
        return;
    }
    while (__hip_atomic_load((__attribute__((address_space(3))) unsigned *) &flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    for (int i = lane; i < 2336 / 4; i += 64) ((uint32_t *) out)[i] = ((const uint32_t *) dst)[i];
}
int main() {
    const int N = 16384;
    std::vector<uint8_t> h(N), o(4096, 0xEE);
    for (int i = 0; i < N; ++i) h[i] = (uint8_t) ((i * 131 + (i >> 8) * 7) & 0xFF);
    uint8_t * d, * dout;
    hipMalloc(&d, N); hipMalloc(&dout, 4096);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int bad_total = 0;
    for (int base : {0, 60000, 70000 / 16 * 16, 122880 - 4096}) {
        hipMemset(dout, 0xEE, 4096);
        hipLaunchKernelGGL(k, dim3(1), dim3(128), base + 4096, 0, d, dout, base);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
        int bad[4] = {0, 0, 0, 0};
        for (int i = 0; i < 1024; ++i) if (o[i] != h[i]) ++bad[0];
        for (int l = 0; l < 64; ++l) for (int b = 0; b < 16; ++b) if (o[1024 + l * 16 + b] != h[4096 + (63 - l) * 16 + b]) ++bad[1];
        for (int i = 0; i < 256; ++i) if (o[2048 + i] != h[8192 + i]) ++bad[2];
        for (int i = 0; i < 32; ++i) if (o[2304 + i] != h[12288 + i]) ++bad[3];
        printf("base %6d: err %d  mismatches: full %d gathered %d 16-lane %d dword-8-lane %d\n", base, (int) e, bad[0], bad[1], bad[2], bad[3]);
        bad_total += bad[0] + bad[1] + bad[2] + bad[3];
    }
    printf(bad_total ? "DMA_PROBE FAILED\n" : "DMA_PROBE OK\n");
    return bad_total != 0;
}
