// repack.hip — row-local, bijective re-ordering of GGUF quant blocks into the HBM layout the GEMV reads.
//
// Why: block_q6_K is 210 B and block_q8_0 is 34 B (reference: ggml/src/ggml-common.h:321-327, :187-191):
// 2-byte aligned only, so the native array-of-blocks layout cannot be streamed with aligned 16-byte
// loads. Inside each ROW we store the block fields as separate, individually aligned streams
// ("row-SoA"). The payload bytes per row are unchanged (= ggml_row_size); the row STRIDE is the payload
// rounded so that the trailing fp16 scale stream ends on a 16-B boundary (pm_weight_row_stride: identical
// to ggml_row_size whenever K % 2048 == 0, i.e. for every Llama-3 shape; <= 14 B per row otherwise), so
// rows and field streams are 16-B aligned for ANY K. Algorithmic bytes per weight are identical.
// Second rule (measured, DESIGN.md "GEMV"): every wave-level 16-byte load of the mat-vec must cover ONE contiguous span and
// no cache line may be touched by two different load instructions - the L1<->L2 line-request rate, not HBM, was the limit
// with native Q4_K blocks (header + 8 data pieces per 144 B). So each 16-byte "piece" a lane loads is its own stream
// (U = units per row, a unit = the 64 weights (Q4_K/Q6_K) or 32 weights (Q8_0) one lane decodes at a time):
//   Q4_K row (nb blocks, U = 4 nb): qa[U][16] | qb[U][16] | hdr[nb][16]           unit (b, j): qs[32j, +16) | qs[32j+16, +16)
//   Q6_K row (nb blocks, U = 4 nb): la[U][16] | lb[U][16] | qh[U][16] | per group of 8 blocks: scales[8][16] | d[8]   (round 5, pm355_device.h;
//                                   rounds 1-4 and -DPM_Q6K_SCD=0: scales[nb][16] | d[nb])
//                                   unit (b, hh, v): ql[64hh+16v, +16) | ql[64hh+32+16v, +16) | qh[32hh+16v, +16)
//   Q8_0 row (nb blocks, U = nb)  : qa[U][16] | qb[U][16] | d[nb]
// Q5_K (176 = 11x16 B) stays native (identity): it only serves attn_v of the 70B mixture (2 % of the bytes).
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
        if (f < 128) {                                                // ql: unit u = 4b + 2hh + v, first / second 16-byte piece
            const long hh = f >> 6, r = f & 63, v = (r >> 4) & 1, second = r >> 5;
            so = second * nb * 64 + (4 * b + 2 * hh + v) * 16 + (r & 15);
        }
        else if (f < 192) so = nb * 128 + b * 64 + (f - 128);
        else if (f < 208) so = pm_q6k_sc_off((uint32_t) nb, (uint32_t) b) + (f - 192);
        else              so = pm_q6k_d_off((uint32_t) nb, (uint32_t) b);
    } else if (type == PM_Q4_K) {
        const long b = o / PM_BS_Q4_K, f = o - b * PM_BS_Q4_K;
        if (f < 16) so = nb * 128 + b * 16 + f;                       // header: d, dmin, scales[12]
        else { const long q = f - 16, j = q >> 5, r = q & 31; so = (r >> 4) * nb * 64 + (4 * b + j) * 16 + (r & 15); }
    } else {                                                          // PM_Q8_0
        const long b = o / PM_BS_Q8_0, f = o - b * PM_BS_Q8_0;
        if (f < 2) so = nb * 32 + b * 2;
        else { const long r = f - 2; so = (r >> 4) * nb * 16 + b * 16 + (r & 15); }
    }
    // GGUF side: rows are row_halfs apart; HBM side: stride_halfs apart (16-B aligned rows)
    if (to_device) dst[row * stride_halfs + so / 2] = src[row * row_halfs + o / 2];
    else           dst[row * row_halfs + o / 2] = src[row * stride_halfs + so / 2];
}

void pm_launch_repack(int type, const void * src, void * dst, int64_t K, int64_t nrows, int to_device, hipStream_t st) {
    const size_t rb = pm_weight_row_bytes(type, K);
    if (!pm_type_is_repacked(type)) {
        if (src != dst) hipMemcpyAsync(dst, src, rb * nrows, hipMemcpyDeviceToDevice, st);
        return;
    }
    const long nb = type == PM_Q8_0 ? K / 32 : K / 256;
    const long halfs = (long) rb / 2;
    const long n = halfs * nrows;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st,
                       (const uint16_t *) src, (uint16_t *) dst, type, nb, halfs, (long) pm_weight_row_stride(type, K) / 2,
                       (long) nrows, to_device);
}
