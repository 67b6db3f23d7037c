// repack.hip — row-local, bijective re-ordering of GGUF quant blocks into the HBM layout the GEMV reads.
//
// Why: block_q6_K is 210 B and block_q8_0 is 34 B (reference: ggml/src/ggml-common.h:321-327, :187-191):
// 2-byte aligned only, so the native array-of-blocks layout cannot be streamed with aligned 16-byte
// loads. Inside each ROW we store the block fields as separate, individually aligned streams
// ("row-SoA"). The payload bytes per row are unchanged (= ggml_row_size); the row STRIDE is the payload
// rounded so that the trailing fp16 scale stream ends on a 16-B boundary (pm_weight_row_stride: identical
// to ggml_row_size whenever K % 2048 == 0, i.e. for every Llama-3 shape; <= 14 B per row otherwise), so
// rows and field streams are 16-B aligned for ANY K. Algorithmic bytes per weight are identical.
//   Q6_K row (nb blocks): ql[nb][128] | qh[nb][64] | scales[nb][16] | d[nb]
//   Q8_0 row (nb blocks): qs[nb][32]  | d[nb]
// Q4_K (144 = 9x16 B) and Q5_K (176 = 11x16 B) are already 16-B granular: identity.
// to_device = 1: GGUF order -> HBM order (used by set_tensor / the weight loader);
// to_device = 0: inverse (get_tensor). Pure byte permutation; tests check round trips bit-exactly.
#include "pm355_device.h"
#include "pm355_kernels.h"

__global__ __launch_bounds__(256) void repack_kernel(const uint16_t * __restrict__ src, uint16_t * __restrict__ dst,
                                                     int type, long nb, long row_halfs, long stride_halfs, long nrows, int to_device) {
    const long i = (long) blockIdx.x * 256 + threadIdx.x;           // one 2-byte element
    if (i >= row_halfs * nrows) return;
    const long row = i / row_halfs;
    const long o = (i - row * row_halfs) * 2;                         // byte offset inside the row, GGUF order
    long so;                                                          // byte offset inside the row, SoA order
    if (type == PM_Q6_K) {
        const long b = o / PM_BS_Q6_K, f = o - b * PM_BS_Q6_K;
        if (f < 128)      so = b * 128 + f;
        else if (f < 192) so = nb * 128 + b * 64 + (f - 128);
        else if (f < 208) so = nb * 192 + b * 16 + (f - 192);
        else              so = nb * 208 + b * 2;
    } else {                                                          // PM_Q8_0
        const long b = o / PM_BS_Q8_0, f = o - b * PM_BS_Q8_0;
        so = f < 2 ? nb * 32 + b * 2 : b * 32 + (f - 2);
    }
    // GGUF side: rows are row_halfs apart; HBM side: stride_halfs apart (16-B aligned rows)
    if (to_device) dst[row * stride_halfs + so / 2] = src[row * row_halfs + o / 2];
    else           dst[row * row_halfs + o / 2] = src[row * stride_halfs + so / 2];
}

void pm_launch_repack(int type, const void * src, void * dst, int64_t K, int64_t nrows, int to_device, hipStream_t st) {
    const size_t rb = pm_weight_row_bytes(type, K);
    if (type != PM_Q6_K && type != PM_Q8_0) {
        if (src != dst) hipMemcpyAsync(dst, src, rb * nrows, hipMemcpyDeviceToDevice, st);
        return;
    }
    const long nb = type == PM_Q6_K ? K / 256 : K / 32;
    const long halfs = (long) rb / 2;
    const long n = halfs * nrows;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st,
                       (const uint16_t *) src, (uint16_t *) dst, type, nb, halfs, (long) pm_weight_row_stride(type, K) / 2,
                       (long) nrows, to_device);
}
