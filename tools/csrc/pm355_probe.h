// pm355_probe.h — measurement helpers, built into libprima_mi355_probe.so (tools/csrc/, NOT part of the product library):
// the HBM streaming-read ceiling of the box (bench.py "measured_stream_read_peak") and the skeleton of a persistent
// loader-wave / consumer-wave decode layer (tools/engine_probe.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

int pm_launch_stream_read(const void * src, size_t bytes, int wg_per_cu, int unroll, void * sink, hipStream_t st);
int pm_launch_chunk_read(const void * src, int n_wg, int nx, long wgx_stride, long wgy_stride, long wave_stride, long outer_stride, long inner_stride,
                         long piece_stride, int chunk, int n_outer, void * sink, hipStream_t st);
int pm_launch_engine_probe(const void * w, long region_stride, int n_regions, int n_layers, int nph, const int * chunks, const int * act_n,
                           const int * out_n, int attn_ph, float attn_us, float * act, long act_stride, void * ctr, int nw, int ns, int nt,
                           int thin, hipStream_t st);

extern "C" {
/* streams `bytes` from HBM exactly once with the mat-vec's access pattern (one 1024-thread workgroup per CU x wg_per_cu, every wave
 * reads its own contiguous span with `unroll` (4 or 8) 16-byte non-temporal loads in flight per lane); `sink` = 4 writable bytes */
__attribute__((visibility("default"))) int pm355_probe_stream_read(const void * src, size_t bytes, int wg_per_cu, int unroll, void * sink, void * stream);
__attribute__((visibility("default"))) int pm355_probe_tr16(const uint16_t * in, uint16_t * out, void * stream);    /* ds_read_b64_tr_b16 semantics (tools/tr16_probe.py) */
/* strided-chunk read (probe.hip): n_wg workgroups of 4 waves; see tools/access_pattern_probe.py */
__attribute__((visibility("default"))) int pm355_probe_chunk_read(const void * src, int n_wg, int nx, int64_t wgx_stride, int64_t wgy_stride, int64_t wave_stride,
                       int64_t outer_stride, int64_t inner_stride, int64_t piece_stride, int chunk, int n_outer, void * sink, void * stream);
/* n_layers decode layers as ONE persistent launch on a run-ahead LDS-DMA weight loader: real byte counts and seams, stand-in consumer */
__attribute__((visibility("default"))) int pm355_probe_engine(const void * w, int64_t region_stride, int n_regions, int n_layers, int nph, const int * chunks,
                       const int * act_n, const int * out_n, int attn_ph, float attn_us, float * act, int64_t act_stride,
                       void * ctr, int nw, int ns, int nt, int thin, float * us, int * err_out, void * stream);
}
