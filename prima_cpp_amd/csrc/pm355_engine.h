// pm355_engine.h — host interface of the persistent decode engine (decode_engine.hip): ONE launch runs every phase of every layer of a
// single-token step (wq | wk | wv + RoPE + KV store, attention, wo, ffn_gate | ffn_up, ffn_down), device-wide barriers in place of kernel boundaries.
// Opt-in (PM355_ENGINE=1): bit-identical to the launches and measured slower than them (profiles/r05_engine_measured.txt).
#pragma once
#include "pm355_kernels.h"

struct pm_eng_plan;                        // a phase list in device memory + the launch's control words

pm_eng_plan * pm_eng_plan_new();
void pm_eng_plan_free(pm_eng_plan * p);
// append a mat-vec phase (same argument block as pm_launch_gemv_fused; f.xf is the f32 activation row, f.norm_w != null needs f.ss_in - the engine's
// rms_norm always takes its sum of squares from producer-side partials). 0, or < 0 when the engine does not serve the job list (types, K, row counts)
int pm_eng_plan_add_matvec(pm_eng_plan * p, const pm_gemv_fused & f);
// append the attention phase over cached cells (what pm_launch_attn_cached computes): q = rotated F16-rounded rows the previous phase stored
int pm_eng_plan_add_attention(pm_eng_plan * p, const float * q, void * kc, void * vc, const int32_t * pos0, const int32_t * seq, long seq_stride, float * out,
                              int H, int Hkv, int dh, int n_ctx, float scale, int max_keys);
int pm_eng_plan_finish(pm_eng_plan * p);   // upload the table; < 0: not launchable on this device
int pm_eng_plan_phases(const pm_eng_plan * p);
int pm_eng_plan_launch(pm_eng_plan * p, hipStream_t st);
// after a synchronize: the launch's watchdog word (0 = clean; else the code of the first wait that gave up) - also re-arms the plan
int pm_eng_plan_status(pm_eng_plan * p);
// out[0] = f64 sum of the f32-rounded squares of x[0 .. K): the one-partial form of the producer-side sum of squares, for a row nobody produced
void pm_launch_sumsq_row(const float * x, int K, double * out, hipStream_t st);
