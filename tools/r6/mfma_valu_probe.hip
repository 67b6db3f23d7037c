// mfma_valu_probe — does a VALU instruction of wave B slow the MFMA stream of wave A on the same SIMD?  (MI355X, gfx950)
// One workgroup per CU, 512 threads: waves 0-3 (one per SIMD) issue 32x32x16 F16 MFMAs back to back, waves 4-7 (their SIMD partners) run a loop of ONE kind
// of vector instruction. Reports cycles per MFMA of wave 0 (s_memtime) for every kind. Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void probe(float * out, unsigned long long * cyc, int iters, int same_wave) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float16v acc[8];
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16) (lane * 0.01f + e); b[e] = (_Float16) (e * 0.5f - lane * 0.02f); }
    uint32_t v0 = lane * 2654435761u, v1 = lane + 77, v2 = 0x3c003c00u, v3 = 0x38003800u;
    float f0 = lane * 0.5f, f1 = 1.0001f, f2 = 0.5f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
            if (KIND == 20) {          // fresh pseudo-random operand bits every iteration (bounded exponents): full toggle rate in the multipliers
                uint32_t * au = (uint32_t *) &a, * bu = (uint32_t *) &b;
#pragma unroll
                for (int q = 0; q < 4; ++q) { v0 = v0 * 1664525u + 1013904223u; au[q] = (v0 & 0x83ff83ffu) | 0x38003800u; bu[q] = ((v0 >> 3) & 0x83ff83ffu) | 0x38003800u; }
            }
            if (same_wave) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (KIND == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v0) : "v"(v2), "v"(v3));
                    if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                    if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
                }
            }
        }
    } else if (KIND != 0) {
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (KIND == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v0) : "v"(v2), "v"(v3));
                if (KIND == 2) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(v0) : "v"(v2));
                if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                if (KIND == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v0) : "v"(v1));
                if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
                if (KIND == 6) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f0) : "v"(v0));
                if (KIND == 7) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(v0) : "v"(f1), "v"(f2));
                if (KIND == 8) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v0) : "v"(f1), "v"(f2));
                if (KIND == 9) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v0) : "v"(v2));
                if (KIND == 10) asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(v0));
                if (KIND == 11) asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(v0) : "v"(v2), "v"(v3));
                if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(unsigned long long *) &cyc[0]) : "v"(0ull), "v"(0ull));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
    float s = f0 + (float) v0;
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND> void run(const char * name, float * out, unsigned long long * cyc, int same) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, out, cyc, iters, same);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, out, cyc, iters, same);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-28s %s: wave 0 %.1f cycles per MFMA   (partner wave 4 busy for %.0f %% of wave 0's time)\n", name, same ? "SAME wave, 16 per 8 MFMAs" : "partner wave",
           (double) h[0] / (iters * 8.0), 100.0 * h[4] / h[0]);
    printf("    kernel %.1f us = %.2f ns per MFMA of wave 0; counter ticks at %.3f GHz\n", ms * 1e3, ms * 1e6 / (iters * 8.0), (double) h[0] / (ms * 1e6));
}

int main() {
    float * out; unsigned long long * cyc;
    hipMalloc((void **) &out, 256 * 512 * 4); hipMalloc((void **) &cyc, 64);
    run<0>("nothing", out, cyc, 0);
    run<20>("random operands each iter", out, cyc, 0);
    run<1>("v_pk_fma_f16", out, cyc, 0);
    run<2>("v_pk_add_f16", out, cyc, 0);
    run<9>("v_pk_mul_f16", out, cyc, 0);
    run<11>("v_fma_f16", out, cyc, 0);
    run<3>("v_perm_b32", out, cyc, 0);
    run<4>("v_and_b32", out, cyc, 0);
    run<10>("v_lshrrev_b32", out, cyc, 0);
    run<5>("v_fma_f32", out, cyc, 0);
    run<6>("v_cvt_f32_ubyte0", out, cyc, 0);
    run<7>("v_fma_mixlo_f16", out, cyc, 0);
    run<8>("v_cvt_pk_f16_f32", out, cyc, 0);
    run<1>("v_pk_fma_f16", out, cyc, 1);
    run<3>("v_perm_b32", out, cyc, 1);
    run<5>("v_fma_f32", out, cyc, 1);
    return 0;
}
