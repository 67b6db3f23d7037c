// attn_flash_mfma.hip — long-context single-token attention on the MATRIX CORES (round 3), for the decode form whose wq | wk | wv launch
// has already rotated q (F16-rounded) and stored the token's K row / V column (QkvEpi): every attended cell is in the cache.
//
// Why: attn_flash.hip walks its keys with VALU dot products - per KV head 8 query heads x n_kv keys x 128 dims x 2 (K.q, P.V) multiply-adds
// plus an F16 -> f32 conversion per operand: at 32k cells 110 us per 70B layer against 19 us of KV bytes (57.5 tok/s, profiles/
// r02_long_context.txt). The GQA group's scores are a [keys x head_dim] . [head_dim x heads] product: on v_mfma_f32_32x32x16_f16 a 32-key
// tile costs 8 + 8 matrix instructions and the kernel is bound by the KV bytes again.
//
// Everything is computed TRANSPOSED like attn_prefill.hip: S^T = K Q^T (keys x heads) and O^T = V^T P^T (head_dim x heads) with
// v_mfma_f32_16x16x32_f16: a query head is a COLUMN of the 16 x 16 accumulators (columns >= n_head / n_head_kv are padding: with the 8
// heads of a Llama-3 GQA group half of the matrix work is useful; the 32 x 32 shape would waste three quarters and needs twice the
// accumulator registers), so the online-softmax statistics of a head are in-lane reductions over 8 keys plus two lane exchanges, and
// the rescaling of O^T is one scalar per lane.
//   * key permutation: a 32-key tile is two 16-row score tiles A and B; row i of A stands for key 8 (i >> 2) + (i & 3), row i of B for
//     that + 4. Accumulator register r of lane group g (rows 4 g + r) is then key 8 g + r in A and 8 g + 4 + r in B: the lane's eight
//     probabilities are keys 8 g .. 8 g + 7 = exactly k-slots 8 g .. + 7 of the B operand of the P.V product, and the matching A operand
//     is ONE 16-byte load of a V^T row (transposed V cache, keys contiguous; the four lane groups of a row read 64 contiguous bytes).
//     No LDS, no lane exchange for P.
//   * a WAVE owns whole 32-key tiles (tiles wave, wave + 4, ... of the workgroup's span) with its own running (max, sum, O^T): no
//     workgroup barrier inside the key loop, and both K and V^T of the next tile are in flight (second register set) while a tile is
//     computed; the four waves' partials are merged once per span through LDS, published write-through, and the last EIGHT workgroups of
//     the KV head to arrive (ticket) merge the spans, each an eighth of the outputs (see the comment at the tickets).
// Rounding points: q, K, V F16; the probabilities exp(s - m) are rounded to F16 as MFMA operands (the default graph rounds the normalised
// p to F16 for the V^T.p product, the flash-attention op keeps p in f32 and rounds its accumulator: both within the reference's backend
// tolerance for these ops, NMSE 5e-4 - tests compare against float64 on the same cache values at 2e-3 of max |out|).
#include "attn_device.h"
#include "pm355_layer_ops.h"
#include "pm355_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

struct FlashM {
    const float * q; const uint16_t * kc, * vc; const int32_t * pos0_ptr, * seq_ptr; long seq_stride; float * out;
    float * M, * L, * P; unsigned * ticket;          // scratch as in attn_flash.hip: M / L [H][nspan], P [nspan][H][dh], ticket [Hkv]
    int H, Hkv, n_ctx, nspan, span; float scale;
    const int32_t * dyn; const void * mask; int mask_f16;
    unsigned long long * ts;                         // (-DPM_TS measurement builds: tools/attn_long_probe.py --ts)
};

constexpr int NMERGE = 8;                            // workgroups per KV head that share the merge of the spans
constexpr int RMAXM = 16;                            // query heads per KV head served (columns 0 .. R-1 of the tiles)

// 16-byte agent-scope load (global_load_dwordx4 sc1): ISSUE only - the caller waits (s_waitcnt vmcnt(0)) and then touches the value
// through an empty asm, which is what orders its uses behind the wait (the compiler does not count loads issued from asm)
__device__ __forceinline__ float4v ld_coh4_issue(const float4 * p) {
    float4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// 16-byte agent-scope write-through store (what four st_act<true> do with one instruction)
__device__ __forceinline__ void st_coh4(float * p, float4v v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

// VROW: the V cache is ROW-major ([cell][n_head_kv * head_dim], the --flash-attn layout of the reference: llm_build_kv, src/llama.cpp:9705) - the 8 keys
// of an operand's k-slots are 8 cache rows apart. A wave then fetches its 32-key tile as whole rows (coalesced 16-byte pieces, same register count as the
// transposed operands), parks it in a private LDS image of DH / 16 sub-tiles [32 keys][16 dims] and reads the operands back with the transposing LDS read
// (ds_read_b64_tr_b16: lane c of a 16-lane group receives column c of the [4 keys][16 dims] block whose rows the group's lanes address - checked on the hardware by
// tools/tr16_probe.py). Keys 8 g .. 8 g + 3 of lane group g sit in block g, keys 8 g + 4 .. + 7 in block 4 + g: each of the two reads of an operand covers 512
// contiguous bytes; sub-tiles are 32 bytes apart from a multiple of 1 KiB so that the 16 lanes that write one key's row hit different banks.
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4v * lds_s4;

// ---- Q8_0 K / V (round 5): the caches are 1-D tensors of NATIVE 34-byte blocks (half d | 32 int8, attn_q8.hip) - a head's row is DH / 32 blocks. The
// operands of the F16 matrix instructions are the DEQUANTIZED values q_i * d rounded to F16 (one rounding per value; the reference dequantizes V the
// same way - v_to_float, ggml.c:15990 - and multiplies K block-wise with the Q8_0-quantized query: attn_q8.hip's prep kernel hands the query over as
// q_i * d_q rounded to F16, so the only difference to ggml_vec_dot_q8_0_q8_0 is that rounding of the two factors, 2^-11 relative each).
// An operand slice = 8 consecutive int8 of one block (8 bytes at a 2-byte-aligned address: gfx950 serves them, tools/r5/unaligned_probe) + the block's d.
constexpr int QB8 = 34;
typedef uint32_t u32x2q __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
struct Q8Piece { u32x2q q; uint32_t d; };                  // raw bytes + the F16 scale's bits (zero-extended)
__device__ __forceinline__ Q8Piece q8_fetch(const uint8_t * blk, int off8) {
    Q8Piece r;
    r.q = *(const PM_G u32x2q *) (blk + 2 + off8);
    r.d = *(const PM_G uint16_t *) blk;
    return r;
}
__device__ __forceinline__ half8 q8_deq(const Q8Piece & r) {
    // int8 b -> F16: (b ^ 0x80) is b + 128 as an unsigned byte; the F16 bit pattern 0x6400 | x is the number 1024 + x; minus 1152 leaves b, exactly
    const uint32_t lo = r.q.x ^ 0x80808080u, hi = r.q.y ^ 0x80808080u;
    const half2v off = {(_Float16) 1152.0f, (_Float16) 1152.0f};
    const _Float16 dh_ = __builtin_bit_cast(_Float16, (uint16_t) r.d);
    const half2v d2 = {dh_, dh_};
    const half2v h0 = (__builtin_bit_cast(half2v, __builtin_amdgcn_perm(0x64646464u, lo, 0x04010400u)) - off) * d2;
    const half2v h1 = (__builtin_bit_cast(half2v, __builtin_amdgcn_perm(0x64646464u, lo, 0x04030402u)) - off) * d2;
    const half2v h2 = (__builtin_bit_cast(half2v, __builtin_amdgcn_perm(0x64646464u, hi, 0x04010400u)) - off) * d2;
    const half2v h3 = (__builtin_bit_cast(half2v, __builtin_amdgcn_perm(0x64646464u, hi, 0x04030402u)) - off) * d2;
    return half8{h0.x, h0.y, h1.x, h1.y, h2.x, h2.y, h3.x, h3.y};
}
// one operand slice of a tile as it travels: F16 (16 bytes) or a Q8_0 piece (8 + 2 bytes)
template <bool Q8> struct Opnd;
template <> struct Opnd<false> { half8 v;   __device__ __forceinline__ half8 get() const { return v; } };
template <> struct Opnd<true>  { Q8Piece v; __device__ __forceinline__ half8 get() const { return q8_deq(v); } };

template <int DH, bool VROW, bool KQ8 = false, bool VQ8 = false>
__global__ __launch_bounds__(256, 2) void attn_flash_mfma_kernel(FlashM p) {
    static_assert(!VQ8 || VROW, "a quantized V cache is row-major (flash-attention graphs)");
    constexpr int KK = DH / 32;                      // k-steps of the S^T product
    constexpr int DT = DH / 16;                      // 16-row tiles of O^T
    constexpr int SUB = 32 * 32 + 32, VW = DT * SUB; // bytes of a V sub-tile image / of a wave's staging area (VROW)
    constexpr int OBYTES = 4 * RMAXM * DH * 4, SBYTES = VROW ? 4 * VW : 0;
    __shared__ float wm[4][16], wl[4][16];
    // the four waves' O^T columns, scaled to the workgroup maximum (after the key loop) / VROW: the waves' V staging images (during it)
    __shared__ __attribute__((aligned(16))) uint8_t obuf[OBYTES > SBYTES ? OBYTES : SBYTES];
    float (* ored)[RMAXM][DH] = (float (*)[RMAXM][DH]) obuf;
    __shared__ float msc[RMAXM][64], lsc[RMAXM][64]; // merge of the spans (<= 64 per KV head)
    __shared__ int last_flag;
    unsigned long long tsv[6] = {PM_TS_NOW(), 0, 0, 0, 0, 0};
    // KV heads fastest in the grid: the workgroups that read the 8 x 256 B pieces of the same 2 KB cell rows start together (9 % at 32k cells)
    const int c = blockIdx.y, g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, lg = lane >> 4;
    const int R = p.H / p.Hkv;
    const int seq = p.seq_ptr ? *p.seq_ptr : 0;
    const unsigned base = p.ticket[16 + g];          // (see the tickets below)
    const int k_begin = c * p.span;
    const long krow = (long) p.Hkv * DH;
    const uint16_t * kc = p.kc + (long) seq * p.seq_stride + (long) g * DH;
    const uint16_t * vc = p.vc + (long) seq * p.seq_stride + (VROW ? (long) g * DH : (long) g * DH * p.n_ctx);
    // Q8_0 caches: byte addressing, a cell row = n_head_kv * DH / 32 blocks
    constexpr int HB8 = DH / 32 * QB8;                 // bytes of a head's row
    const long krow8 = (long) p.Hkv * HB8;
    const uint8_t * kc8 = (const uint8_t *) p.kc + (long) g * HB8, * vc8 = (const uint8_t *) p.vc + (long) g * HB8;
    // Everything below up to the first use of n_kv is issued BEFORE the position arrives (addresses are clamped into the cache): the
    // scalar round trip for the position overlaps the first tile's loads instead of preceding them.
    // Q^T as B operand: lane (col = head, lg) holds q[head][32 kk + 8 lg .. +8] (already rotated and F16-rounded by the QKV epilogue)
    half8 qf[KK];
    {
        const float * qr = p.q + ((long) g * R + min(col, R - 1)) * DH + 8 * lg;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const float4 a = *(const float4 *) (qr + 32 * kk), b = *(const float4 *) (qr + 32 * kk + 4);
            qf[kk] = col < R ? half8{(_Float16) a.x, (_Float16) a.y, (_Float16) a.z, (_Float16) a.w, (_Float16) b.x, (_Float16) b.y, (_Float16) b.z, (_Float16) b.w}
                             : half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // the key this lane's A-operand row stands for in score tile A (tile B: + 4), see the header
    const int krow_i = 8 * (col >> 2) + (col & 3);
    typedef Opnd<KQ8> KOp;
    typedef Opnd<VQ8> VOp;
    KOp ka0[2][KK], ka1[2][KK]; VOp va0[DT], va1[DT];
    auto fetch = [&](KOp (&ka)[2][KK], VOp (&va)[DT], int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long cell = min(kt + krow_i + 4 * t, p.n_ctx - 1);
            if constexpr (KQ8) {
                const uint8_t * kr = kc8 + cell * krow8;                       // k-step kk of the S^T product = block kk of the head's row
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) ka[t][kk].v = q8_fetch(kr + kk * QB8, 8 * lg);
            } else {
                const uint16_t * kr = kc + cell * krow + 8 * lg;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) ka[t][kk].v = *(const half8 *) (kr + 32 * kk);
            }
        }
        if constexpr (!VROW) {
            const uint16_t * vr = vc + (long) col * p.n_ctx + min(kt + 8 * lg, p.n_ctx - 8);
#pragma unroll
            for (int d = 0; d < DT; ++d) va[d].v = *(const half8 *) (vr + (long) 16 * d * p.n_ctx);
        } else {
            // raw rows: load n covers 64 / CPK keys, lane l = (key l / CPK, 16-byte piece l % CPK of the head's DH halves)
            constexpr int CPK = DH / 8;
#pragma unroll
            for (int n = 0; n < DT; ++n) {
                const int key = n * (64 / CPK) + lane / CPK, ch = lane % CPK;
                const long cell = min(kt + key, p.n_ctx - 1);
                if constexpr (VQ8) va[n].v = q8_fetch(vc8 + cell * krow8 + (ch >> 2) * QB8, 8 * (ch & 3));     // dims 8 ch .. + 7 = a quarter of block ch / 4
                else va[n].v = *(const half8 *) (vc + cell * krow + 8 * ch);
            }
        }
    };
    constexpr int KSTEP = 128;                        // (a wave's tiles interleave with the other waves': contiguous quarter spans measured 3 % slower)
    int kt = k_begin + 32 * wave;
    fetch(ka0, va0, kt);
    const int n_kv = p.dyn ? p.dyn[1] : p.pos0_ptr[seq] + 1;
    if (k_begin >= n_kv) return;                     // (the ticket target is the number of ACTIVE spans)
    const int nact = (n_kv + p.span - 1) / p.span;
    const int k_end = min(k_begin + p.span, n_kv);
    float m = -INFINITY, l = 0.0f;                   // running statistics of THIS lane's column over the wave's tiles (the four lane groups agree)
    float4v o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    // one 32-key tile whose K and V^T are in (kc_, vc_); the next tile's loads go to the other register set first
    auto tile = [&](int kt_, KOp (&kc_)[2][KK], VOp (&vc_)[DT], KOp (&kn_)[2][KK], VOp (&vn_)[DT]) __attribute__((always_inline)) {
        if (kt_ + KSTEP < k_end) fetch(kn_, vn_, kt_ + KSTEP);
        float4v acc[2] = {float4v{0.0f, 0.0f, 0.0f, 0.0f}, float4v{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc_[0][kk].get(), qf[kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc_[1][kk].get(), qf[kk], acc[1], 0, 0, 0);
        }
        float s[8], mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int key = kt_ + 8 * lg + r;
            float v = key < n_kv ? acc[r >> 2][r & 3] * p.scale : -INFINITY;
            if (p.mask && key < n_kv) v += attn_mask_at(p.mask, p.mask_f16, key);
            s[r] = v; mt = fmaxf(mt, v);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mn = fmaxf(m, mt);
        half8 pf = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (mn != -INFINITY) {                       // (else nothing visible so far for this head: masked cells only)
            const float f = m == -INFINITY ? 0.0f : __expf(m - mn);
            float lt = 0.0f, e[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) { e[r] = s[r] == -INFINITY ? 0.0f : __expf(s[r] - mn); lt += e[r]; }
            lt += __shfl_xor(lt, 16);
            lt += __shfl_xor(lt, 32);
            l = l * f + lt; m = mn;
            if (__builtin_amdgcn_ballot_w64(f != 1.0f)) {        // the maximum moved for some head of this wave: rescale the accumulators
#pragma unroll
                for (int d = 0; d < DT; ++d) o[d] *= f;
            }
            pf = half8{(_Float16) e[0], (_Float16) e[1], (_Float16) e[2], (_Float16) e[3], (_Float16) e[4], (_Float16) e[5], (_Float16) e[6], (_Float16) e[7]};
        }
        if constexpr (!VROW) {
#pragma unroll
            for (int d = 0; d < DT; ++d) o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vc_[d].get(), pf, o[d], 0, 0, 0);
        } else {
            constexpr int CPK = DH / 8;
            uint8_t * wb = obuf + wave * VW;
#pragma unroll
            for (int n = 0; n < DT; ++n) {
                const int kl = n * (64 / CPK) + lane / CPK, ch = lane % CPK;                 // key inside the tile, 16-byte piece of its row
                const int pos = ((kl & 4) ? 16 + 4 * (kl >> 3) : 4 * (kl >> 3)) + (kl & 3);    // row of the sub-tile image: block (k / 8 or 4 + k / 8), row k % 4
                *(half8 *) (wb + (ch >> 1) * SUB + pos * 32 + (ch & 1) * 16) = vc_[n].get();
            }
            const uint8_t * rb = wb + (4 * lg + (col >> 2)) * 32 + (col & 3) * 8;             // block lg, the lane's 4 halves of it
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4) (rb + d * SUB));          // keys 8 lg .. + 3 of dim 16 d + col
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4) (rb + d * SUB + 512));    // keys 8 lg + 4 .. + 7
                typedef short short8v __attribute__((ext_vector_type(8)));
                const short8v v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, v8), pf, o[d], 0, 0, 0);
            }
        }
    };
    for (; kt < k_end; kt += 2 * KSTEP) {
        tile(kt, ka0, va0, ka1, va1);
        if (tsv[1] == 0) tsv[1] = PM_TS_NOW();
        if (kt + KSTEP >= k_end) break;
        tile(kt + KSTEP, ka1, va1, ka0, va0);
    }
    tsv[2] = PM_TS_NOW();
    // ---- merge the four waves: workgroup maximum per head, O^T columns scaled to it
    if (lg == 0) { wm[wave][col] = m; wl[wave][col] = l; }
    __syncthreads();
    const float mw = fmaxf(fmaxf(wm[0][col], wm[1][col]), fmaxf(wm[2][col], wm[3][col]));
    const float fw = (m == -INFINITY || mw == -INFINITY) ? 0.0f : __expf(m - mw);
    if (col < R) {
#pragma unroll
        for (int d = 0; d < DT; ++d) *(float4v *) &ored[wave][col][16 * d + 4 * lg] = o[d] * fw;
    }
    __syncthreads();
    for (int i = tid; i < R * DH / 4; i += 256) {    // publish: 16-byte write-through stores (agent scope: readable by the mergers on other XCDs)
        const int h = i / (DH / 4), e = 4 * (i - h * (DH / 4));
        const float4v ov = (*(const float4v *) &ored[0][h][e] + *(const float4v *) &ored[1][h][e]) + (*(const float4v *) &ored[2][h][e] + *(const float4v *) &ored[3][h][e]);
        st_coh4(p.P + ((long) c * p.H + g * R + h) * DH + e, ov);
    }
    if (tid < R) {
        const float mh = fmaxf(fmaxf(wm[0][tid], wm[1][tid]), fmaxf(wm[2][tid], wm[3][tid]));
        float lh = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) lh += (wm[w][tid] == -INFINITY || mh == -INFINITY) ? 0.0f : wl[w][tid] * __expf(wm[w][tid] - mh);
        st_act<true>(p.M + (g * R + tid) * p.nspan + c, mh);
        st_act<true>(p.L + (g * R + tid) * p.nspan + c, lh);
    }
    // ---- tickets: the last NM workgroups of this KV head to arrive merge the spans, each a slice of the outputs.
    // One workgroup reading all partials of a KV head (64 x 4 KB with 8 query heads) is bound by what ONE CU pulls from memory another
    // XCD wrote (~25 GB/s: 10 us, measured - more than the whole key loop at 8k cells); the late arrivals are running anyway, so NM of
    // them share the reads. A merger waits (bounded spin of its thread 0 on the arrival counter) for the few workgroups still behind it:
    // they are resident or will get a slot as non-mergers leave (NM x n_head_kv <= 128 workgroups spin at most, checked by the launcher).
    tsv[3] = PM_TS_NOW();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int NM = min(NMERGE, nact);
    if (tid == 0) {
        // arrivals are counted by a counter that is never reset; `base` = its value when this launch began (read at entry; moved on by
        // the last arrival, which is after every workgroup of the launch has read it): no second counter, nobody resets under a spinner
        const unsigned t = __hip_atomic_fetch_add((PM_G unsigned *) (p.ticket + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int idx = (int) (t - base);
        if (idx == nact - 1) __hip_atomic_store((PM_G unsigned *) (p.ticket + 16 + g), t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int me = idx - (nact - NM);
        int me_ = me;
        if (me >= 0) {
            int spin = 0;
            for (; spin < (1 << 22); ++spin) {
                if (__hip_atomic_load((PM_G unsigned *) (p.ticket + g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base >= (unsigned) nact) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (spin == (1 << 22)) me_ = me + (1 << 20);          // never seen; if it ever happens the merger writes NaN instead of a wrong sum
        }
        last_flag = me_;
    }
    __syncthreads();
    tsv[4] = PM_TS_NOW();
    const bool timed_out = last_flag >= (1 << 20);
    const int me = timed_out ? last_flag - (1 << 20) : last_flag;
#ifdef PM_TS
    auto ts_out = [&]() {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (p.ts && tid == 0 && lin < 4 * PM_TS_WGS) {      // (the probe enables 4 slots and records ONE launch: room for 1024 workgroups)
            unsigned long long * o_ = p.ts + (size_t) lin * 8;
            tsv[5] = PM_TS_NOW();
            for (int i = 0; i < 6; ++i) o_[i] = tsv[i];
            o_[6] = (unsigned long long) (me >= 0); o_[7] = (unsigned long long) (gridDim.x * gridDim.y);
        }
    };
    if (me < 0) { ts_out(); return; }
#endif
    if (me < 0) return;
    // (no acquire fence - buffer_inv sc1 walks the L2: the partials and statistics are read with agent-scope loads instead)
    // this merger's outputs: float4 items [i0, i0 + cnt) of the R x DH / 4 of the KV head; a thread = (item, slice of the partials):
    // all its loads (statistics and partial rows) are issued before anything is computed - one round trip
    const int items = R * DH / 4, ipm = (items + NM - 1) / NM, i0 = me * ipm, cnt = max(0, min(ipm, items - i0));
    int SL = min(8, nact);                           // slices of the partials: <= 8 partials each
    if (cnt * SL > 512) SL = 512 / cnt;              // (few spans, many heads: 2 x 256 units at most)
    const long stride4 = (long) p.H * DH / 4;
    float4v pv[2][8];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int u = tid + 256 * it;                // (cnt x SL <= 512 units)
        const int item = u % max(cnt, 1), sl = u / max(cnt, 1);
        const int c_lo = sl * nact / SL;
        const float4 * src = (const float4 *) (p.P + (long) g * R * DH) + min(i0 + item, items - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            pv[it][j] = ld_coh4_issue(src + min(c_lo + j, nact - 1) * stride4);      // (beyond its slice / no unit: re-reads a valid row, weight 0 or unused)
    }
    for (int i = tid; i < R * nact; i += 256) {
        const int h = i / nact, cc = i - h * nact;
        msc[h][cc] = ld_act<true>(p.M + (g * R + h) * p.nspan + cc);
        lsc[h][cc] = ld_act<true>(p.L + (g * R + h) * p.nspan + cc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(pv[it][j]));
    __syncthreads();
    for (int h = wave; h < R; h += 4) {              // weights of head h: lane = span
        const float mc = lane < nact ? msc[h][lane] : -INFINITY;
        float mm = mc;
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o_));
        const float f = mc == -INFINITY ? 0.0f : __expf(mc - mm);
        const float den = wave_sum(lane < nact ? f * lsc[h][lane] : 0.0f);
        if (lane < nact) msc[h][lane] = den > 0.0f ? f / den : 0.0f;
    }
    __syncthreads();
    float4 * red = (float4 *) &ored[0][0][0];        // [SL][cnt] partial sums (<= 512 float4)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int u = tid + 256 * it;
        if (u < cnt * SL) {
            const int item = u % cnt, sl = u / cnt;
            const int c_lo = sl * nact / SL, c_hi = (sl + 1) * nact / SL;
            const int h = (i0 + item) / (DH / 4);
            float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = c_lo + j < c_hi ? msc[h][c_lo + j] : 0.0f;
                a.x += w * pv[it][j][0]; a.y += w * pv[it][j][1]; a.z += w * pv[it][j][2]; a.w += w * pv[it][j][3];
            }
            red[u] = a;
        }
    }
    __syncthreads();
    if (tid < cnt) {
        float4 a = red[tid];
        for (int sl = 1; sl < SL; ++sl) { const float4 b = red[sl * cnt + tid]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        if (timed_out) a.x = a.y = a.z = a.w = __builtin_nanf("");      // (a workgroup of this KV head never arrived: fail loudly downstream)
        *((float4 *) (p.out + (long) g * R * DH) + i0 + tid) = a;
    }
#ifdef PM_TS
    ts_out();
#endif
}

} // namespace

// q = rotated, F16-rounded query rows; the caches already hold this token (transposed V cache). scratch: pm_attn_flash_scratch_floats()
// zeroed once. -1: shape not served (head_dim 64 / 128, <= 16 query heads per KV head, n_ctx % 8 == 0).
// Shapes the matrix-core kernel serves: head_dim 64 / 128, <= 16 query heads per KV head, n_ctx % 8 == 0, and <= 16 KV heads - the arrival
// counters live at ticket[g] / ticket[16 + g] and the bounded spin of the merging workgroups assumes that the whole grid (n_head_kv x <= 64 spans
// <= 2 x CUs workgroups of 4 waves) is resident. MHA models (Llama-2-7B / 13B: 32 / 40 KV heads) keep the round-2 kernel (attn_flash.hip).
int pm_attn_flash_cached_ok(int H, int Hkv, int dh, int n_ctx) {
    if ((dh != 64 && dh != 128) || Hkv < 1 || H % Hkv || H / Hkv > RMAXM || n_ctx % 8 || Hkv > 16) return -1;
    int parts = 256 / Hkv;
    parts = parts < 16 ? 16 : parts > 64 ? 64 : parts;
    return Hkv * parts <= 2 * pm_device_cus() ? 0 : -1;
}

int pm_launch_attn_flash_cached(const float * q, void * kc, void * vc, const int32_t * pos0, const int32_t * seq, long seq_stride, float * out,
                                float * scratch, int H, int Hkv, int dh, int n_ctx, float scale, hipStream_t st, const int32_t * dyn,
                                const void * mask, int mask_f16, int max_cells, int v_rowmajor, int k_q8, int v_q8) {
    if (!scratch || pm_attn_flash_cached_ok(H, Hkv, dh, n_ctx)) return -1;
    if ((k_q8 || v_q8) && (!v_rowmajor || seq)) return -1;          // Q8_0 caches: flash-attention layout (row-major V), one sequence slab
    if (!pos0) pos0 = dyn;
    if (!pos0) return -1;
    const int CK = 128;
    const int nspan_max = (n_ctx + CK - 1) / CK;
    const int cells = max_cells > 0 && max_cells < n_ctx ? max_cells : n_ctx;
    // spans: 128 keys each until the KV heads together fill the CUs (256 workgroups: 32 spans per head at 8 KV heads - measured against 64
    // and 16 at 8k .. 32k cells), then longer spans; never more than 64 partials per head (the mergers' LDS tables)
    int span = CK, parts = 256 / Hkv;
    parts = parts < 16 ? 16 : parts > 64 ? 64 : parts;
    // (measurement only: PM355_FLASH_PARTS = spans per KV head, 1 .. 64 - profiles/r05_long_context_spans.txt)
    static const int parts_env = [] { const char * e = getenv("PM355_FLASH_PARTS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? v : 0; }();
    if (parts_env) parts = parts_env;
    while ((cells + span - 1) / span > parts) span *= 2;
    FlashM p = {};
    p.q = q; p.kc = (const uint16_t *) kc; p.vc = (const uint16_t *) vc; p.pos0_ptr = pos0; p.seq_ptr = seq; p.seq_stride = seq_stride; p.out = out;
    p.M = scratch; p.L = p.M + (size_t) H * nspan_max; p.P = p.L + (size_t) H * nspan_max; p.ticket = (unsigned *) (p.P + (size_t) nspan_max * H * dh) + Hkv + 16;    // (behind attn_flash.hip's tickets, which must read 0 between its launches)
    p.H = H; p.Hkv = Hkv; p.n_ctx = n_ctx; p.nspan = nspan_max; p.span = span; p.scale = scale;
    p.dyn = dyn; p.mask = mask; p.mask_f16 = mask_f16; p.ts = pm_ts_next_slot();
    const dim3 grid(Hkv, (cells + span - 1) / span);
    if (k_q8 || v_q8) {
#define PM_FQ8(DH_) do { \
            if (k_q8 && v_q8) hipLaunchKernelGGL((attn_flash_mfma_kernel<DH_, true, true, true>), grid, dim3(256), 0, st, p); \
            else if (k_q8)    hipLaunchKernelGGL((attn_flash_mfma_kernel<DH_, true, true, false>), grid, dim3(256), 0, st, p); \
            else              hipLaunchKernelGGL((attn_flash_mfma_kernel<DH_, true, false, true>), grid, dim3(256), 0, st, p); } while (0)
        if (dh == 128) PM_FQ8(128); else PM_FQ8(64);
#undef PM_FQ8
    } else if (v_rowmajor) {
        if (dh == 128) hipLaunchKernelGGL((attn_flash_mfma_kernel<128, true>), grid, dim3(256), 0, st, p);
        else           hipLaunchKernelGGL((attn_flash_mfma_kernel<64, true>), grid, dim3(256), 0, st, p);
    } else {
        if (dh == 128) hipLaunchKernelGGL((attn_flash_mfma_kernel<128, false>), grid, dim3(256), 0, st, p);
        else           hipLaunchKernelGGL((attn_flash_mfma_kernel<64, false>), grid, dim3(256), 0, st, p);
    }
    return 0;
}
