"""Op-level host mirror of the reference interface for the hot path (ggml op names and argument
meaning; see include/prima_mi355.h for the reference function each one replaces).

Tensors are torch CUDA tensors used purely as device memory handles; all compute happens in
libprima_mi355.so on the current torch stream.
"""
import numpy as np
import torch

from . import lib as L
from .lib import F16, F32, Q4_K, Q5_K, Q6_K, Q8_0, Q8_K, check, ptr, stream_ptr  # noqa: F401


def _dev():
    if not torch.cuda.is_available():
        raise L.PM355Error("no HIP device visible; prima_cpp_amd has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


class QWeight:
    """A quantized weight matrix [N rows, K cols] resident in HBM in the library's layout."""

    def __init__(self, qtype, K, N, data):
        self.type, self.K, self.N, self.data = qtype, K, N, data

    @property
    def nbytes(self):
        return self.data.numel()


def upload_weight(qtype, blocks, K, N):
    """ggml_backend_tensor_set semantics: host GGUF-order bytes -> HBM (H2D copy + row-local repack)."""
    lib = L.load()
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1)
    assert blocks.size == L.row_size(qtype, K) * N, (blocks.size, L.row_size(qtype, K) * N)
    raw = torch.from_numpy(blocks).to(_dev())
    if qtype in (Q4_K, Q6_K, Q8_0):
        dst = torch.zeros(N * lib.pm355_row_stride(qtype, K), dtype=torch.uint8, device=raw.device)
        check(lib.pm355_repack_rows(qtype, ptr(raw), ptr(dst), K, N, 1, stream_ptr()), "repack_rows")
        return QWeight(qtype, K, N, dst)
    return QWeight(qtype, K, N, raw)


def download_weight(w):
    """ggml_backend_tensor_get semantics: HBM layout -> host GGUF-order bytes."""
    lib = L.load()
    if w.type in (Q4_K, Q6_K, Q8_0):
        tmp = torch.empty(w.N * L.row_size(w.type, w.K), dtype=torch.uint8, device=w.data.device)
        check(lib.pm355_repack_rows(w.type, ptr(w.data), ptr(tmp), w.K, w.N, 0, stream_ptr()), "repack_rows")
        return tmp.cpu().numpy()
    return w.data.cpu().numpy()


def quantize_act(x, act_type=Q8_K):
    """x: f32 [rows, K] (or [K]) -> device buffer of quantized rows (library row-SoA layout)."""
    lib = L.load()
    x = x.contiguous().view(-1, x.shape[-1])
    rows, K = x.shape
    rb = lib.pm355_q8_K_row_size(K) if act_type == Q8_K else lib.pm355_q8_0_row_size(K)
    out = torch.empty(rows * rb, dtype=torch.uint8, device=x.device)
    fn = lib.pm355_quantize_q8_K if act_type == Q8_K else lib.pm355_quantize_q8_0
    check(fn(ptr(x), ptr(out), K, rows, stream_ptr()), "quantize")
    return out


def act_to_ggml_blocks(yq, act_type, K, rows):
    """Quantized activation rows -> numpy bytes in the reference's block_q8_K / block_q8_0 layout (tests)."""
    lib = L.load()
    out = torch.empty(rows * L.row_size(act_type, K), dtype=torch.uint8, device=yq.device)
    check(lib.pm355_act_to_ggml_blocks(act_type, ptr(yq), ptr(out), K, rows, stream_ptr()), "act_to_ggml_blocks")
    return out.cpu().numpy()


def rms_norm(x, w, eps, want_f32=True, want_q8=False):
    lib = L.load()
    x2 = x.contiguous().view(-1, x.shape[-1])
    rows, K = x2.shape
    y = torch.empty_like(x2) if want_f32 else None
    yq = torch.empty(rows * lib.pm355_q8_K_row_size(K), dtype=torch.uint8, device=x.device) if want_q8 else None
    check(lib.pm355_rms_norm(ptr(x2), ptr(w), ptr(y), ptr(yq), K, rows, float(eps), stream_ptr()), "rms_norm")
    if want_f32 and want_q8:
        return y.view_as(x), yq
    return y.view_as(x) if want_f32 else yq


def vec_dot_act_type(qtype):
    return Q8_0 if qtype == Q8_0 else Q8_K


def mul_mat_vec(w, x=None, xq=None, w2=None, bias=None, resid=None, ncols=1):
    """ggml_mul_mat(W, x) for ncols <= 8 f32 activation columns x [ncols, K] (quantized on device to the
    reference's vec_dot_type first), or pre-quantized xq. Returns f32 [ncols, N]."""
    lib = L.load()
    if xq is None:
        x2 = x.contiguous().view(-1, w.K)
        ncols = x2.shape[0]
        xq = quantize_act(x2, vec_dot_act_type(w.type))
    y = torch.empty((ncols, w.N), dtype=torch.float32, device=w.data.device)
    check(lib.pm355_mul_mat_vec_q(w.type, ptr(w.data), ptr(w2.data) if w2 is not None else None, w.K, w.N, ptr(xq),
                                  ncols, ptr(y), w.N, ptr(bias), ptr(resid), stream_ptr()), "mul_mat_vec_q")
    return y


def mul_mat_vec_dbg(w, xq):
    """Returns (y f32 [N], int partials int32 [N, units, 2])."""
    import ctypes as C
    lib = L.load()
    upr = C.c_int64(0)
    check(lib.pm355_mul_mat_vec_q_dbg(w.type, None, w.K, w.N, None, None, None, C.addressof(upr), None), "units query")
    units = int(upr.value)
    y = torch.empty(w.N, dtype=torch.float32, device=w.data.device)
    ip = torch.zeros((w.N, units, 2), dtype=torch.int32, device=w.data.device)
    check(lib.pm355_mul_mat_vec_q_dbg(w.type, ptr(w.data), w.K, w.N, ptr(xq), ptr(y), ptr(ip), C.addressof(upr),
                                      stream_ptr()), "mul_mat_vec_q_dbg")
    assert upr.value == units
    return y, ip


class RopeParams(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("n_dims", _C.c_int32), ("mode", _C.c_int32), ("n_ctx_orig", _C.c_int32),
                ("freq_base", _C.c_float), ("freq_scale", _C.c_float), ("ext_factor", _C.c_float),
                ("attn_factor", _C.c_float), ("beta_fast", _C.c_float), ("beta_slow", _C.c_float)]


def rope_kv_store(q, k, v, k_cache, v_cache, pos0, n_head, n_head_kv, head_dim, n_ctx, freq_factors=None, mode=0,
                  n_ctx_orig=8192, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0,
                  beta_fast=32.0, beta_slow=1.0, n_dims=None):
    """ggml_rope_ext on Q and K + llm_build_kv_store. q [T, H*dh], k/v [T, Hkv*dh] f32; caches int16-typed
    tensors holding F16 bits. Returns (q_rot, k_rot_f32)."""
    import ctypes as C
    lib = L.load()
    T = q.shape[0]
    rp = RopeParams(n_dims or head_dim, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow)
    q_out = torch.empty_like(q)
    k_out = torch.empty_like(k)
    pos = torch.tensor([pos0], dtype=torch.int32, device=q.device)
    check(lib.pm355_rope_kv_store(ptr(q), ptr(k), ptr(v), ptr(q_out), ptr(k_out), ptr(k_cache), ptr(v_cache), ptr(pos),
                                  ptr(freq_factors), T, n_head, n_head_kv, head_dim, n_ctx, C.addressof(rp), stream_ptr()),
          "rope_kv_store")
    return q_out, k_out


def attn_decode(q, k_cache, v_cache, pos0, n_head, n_head_kv, head_dim, n_ctx, scale):
    lib = L.load()
    T = q.shape[0]
    out = torch.empty_like(q)
    pos = torch.tensor([pos0], dtype=torch.int32, device=q.device)
    check(lib.pm355_attn_decode(ptr(q), ptr(k_cache), ptr(v_cache), ptr(pos), ptr(out), T, n_head, n_head_kv, head_dim,
                                n_ctx, float(scale), stream_ptr()), "attn_decode")
    return out


def attn_decode_split(q_rot, k_cache, v_cache, pos, n_head, n_head_kv, head_dim, n_ctx, scale):
    """Long-context single-token attention (keys split over workgroups). q_rot [1, H*dh]: rotated queries; pos = token index."""
    import ctypes as C
    lib = L.load()
    lib.pm355_attn_split_scratch_floats.restype = C.c_size_t
    lib.pm355_attn_split_scratch_floats.argtypes = [C.c_int] * 3
    lib.pm355_attn_decode_split.restype = C.c_int
    lib.pm355_attn_decode_split.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_float, C.c_void_p]
    scratch = torch.empty(lib.pm355_attn_split_scratch_floats(n_head, head_dim, n_ctx), dtype=torch.float32, device=q_rot.device)
    out = torch.empty_like(q_rot)
    p = torch.tensor([pos], dtype=torch.int32, device=q_rot.device)
    check(lib.pm355_attn_decode_split(ptr(q_rot), ptr(k_cache), ptr(v_cache), ptr(p), ptr(out), ptr(scratch), n_head, n_head_kv,
                                      head_dim, n_ctx, float(scale), stream_ptr()), "attn_decode_split")
    return out


def attn_split_scratch(n_head, head_dim, n_ctx, device="cuda"):
    """Zeroed scratch of the long-context attention kernels (partials + tickets; the kernels leave the tickets at zero)."""
    import ctypes as C
    lib = L.load()
    lib.pm355_attn_split_scratch_floats.restype = C.c_size_t
    lib.pm355_attn_split_scratch_floats.argtypes = [C.c_int] * 3
    return torch.zeros(lib.pm355_attn_split_scratch_floats(n_head, head_dim, n_ctx), dtype=torch.float32, device=device)


def attn_prefill(q, k_cache, v_cache, pos0, n_head, n_head_kv, head_dim, n_ctx, scale):
    """Causal multi-token attention on MFMA (same contract as attn_decode)."""
    import ctypes as C
    lib = L.load()
    lib.pm355_attn_prefill.restype = C.c_int
    lib.pm355_attn_prefill.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_float, C.c_void_p]
    T = q.shape[0]
    out = torch.empty_like(q)
    pos = torch.tensor([pos0], dtype=torch.int32, device=q.device)
    check(lib.pm355_attn_prefill(ptr(q), ptr(k_cache), ptr(v_cache), ptr(pos), ptr(out), T, n_head, n_head_kv, head_dim,
                                 n_ctx, float(scale), stream_ptr()), "attn_prefill")
    return out


def argmax(x):
    lib = L.load()
    idx = torch.empty(1, dtype=torch.int32, device=x.device)
    check(lib.pm355_argmax(ptr(x), x.numel(), ptr(idx), None, stream_ptr()), "argmax")
    return idx


def get_rows(w, tokens):
    lib = L.load()
    out = torch.empty((tokens.numel(), w.K), dtype=torch.float32, device=tokens.device)
    check(lib.pm355_get_rows(w.type, ptr(w.data), w.K, ptr(tokens), tokens.numel(), ptr(out), stream_ptr()), "get_rows")
    return out


def attn_rope_fused(q, k, v, k_cache, v_cache, pos0, n_head, n_head_kv, head_dim, n_ctx, scale, freq_factors=None, mode=0,
                    n_ctx_orig=8192, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0,
                    beta_fast=32.0, beta_slow=1.0, n_dims=None):
    """Single-token rope + KV store + attention in one launch. q [H*dh], k/v [Hkv*dh] raw projections."""
    import ctypes as C
    lib = L.load()
    rp = RopeParams(n_dims or head_dim, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow)
    out = torch.empty_like(q)
    pos = torch.tensor([pos0], dtype=torch.int32, device=q.device)
    check(lib.pm355_attn_rope_fused(ptr(q), ptr(k), ptr(v), ptr(k_cache), ptr(v_cache), ptr(pos), ptr(freq_factors), ptr(out),
                                    n_head, n_head_kv, head_dim, n_ctx, float(scale), C.addressof(rp), stream_ptr()),
          "attn_rope_fused")
    return out


class MatvecJob(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("type", _C.c_int32), ("pad_", _C.c_int32), ("N", _C.c_int64), ("W", _C.c_void_p), ("W2", _C.c_void_p),
                ("y", _C.c_void_p), ("bias", _C.c_void_p), ("resid", _C.c_void_p)]


def mul_mat_vec_fused(ws, x, norm_w=None, eps=0.0, w2s=None, biases=None, resids=None):
    """One launch: y_j = W_j . q(x) for up to 3 matrices sharing the f32 row x (optionally rms_norm(x)*norm_w first).
    Returns the list of f32 outputs [N_j]."""
    import ctypes as C
    lib = L.load()
    lib.pm355_mul_mat_vec_fused.restype = C.c_int
    lib.pm355_mul_mat_vec_fused.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    n = len(ws)
    jobs = (MatvecJob * n)()
    ys = []
    for j, w in enumerate(ws):
        y = torch.empty(w.N, dtype=torch.float32, device=w.data.device)
        ys.append(y)
        jobs[j] = MatvecJob(w.type, 0, w.N, ptr(w.data), ptr(w2s[j].data) if w2s else None, ptr(y),
                            ptr(biases[j]) if biases and biases[j] is not None else None,
                            ptr(resids[j]) if resids and resids[j] is not None else None)
    check(lib.pm355_mul_mat_vec_fused(C.addressof(jobs), n, ws[0].K, ptr(x), ptr(norm_w), float(eps), stream_ptr()),
          "mul_mat_vec_fused")
    return ys


def mul_mat_vec_fused_ss(ws, x, norm_w=None, eps=0.0, resids=None, want_sumsq=False, sumsq_in=None):
    """pm355_mul_mat_vec_fused_ss: like mul_mat_vec_fused; want_sumsq -> also returns the per-workgroup f64 partial sums of squares of the
    (single) output row; sumsq_in (f64 tensor) -> the rms_norm takes its sum of squares from those partials."""
    import ctypes as C
    lib = L.load()
    lib.pm355_mul_mat_vec_fused_ss.restype = C.c_int
    lib.pm355_mul_mat_vec_fused_ss.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.pm355_mul_mat_vec_fused_grid.restype = C.c_int
    lib.pm355_mul_mat_vec_fused_grid.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    n = len(ws)
    jobs = (MatvecJob * n)()
    ys = []
    for j, w in enumerate(ws):
        y = torch.empty(w.N, dtype=torch.float32, device=w.data.device)
        ys.append(y)
        jobs[j] = MatvecJob(w.type, 0, w.N, ptr(w.data), None, ptr(y), None, ptr(resids[j]) if resids and resids[j] is not None else None)
    ss = None
    if want_sumsq:
        g = lib.pm355_mul_mat_vec_fused_grid(C.addressof(jobs), n, ws[0].K)
        assert g > 0, g
        ss = torch.zeros(g, dtype=torch.float64, device=x.device)
    check(lib.pm355_mul_mat_vec_fused_ss(C.addressof(jobs), n, ws[0].K, ptr(x), ptr(norm_w), float(eps), ptr(ss), ptr(sumsq_in),
                                         0 if sumsq_in is None else sumsq_in.numel(), stream_ptr()), "mul_mat_vec_fused_ss")
    return (ys, ss) if want_sumsq else ys


def mul_mat_small(w, x=None, xq=None, n_tokens=None, bias=None, resid=None):
    """1..32 tokens on the integer matrix cores (mmq_i8.hip): x f32 [T, K] (quantized to Q8_K on device) or pre-quantized xq -> f32 [T, N]."""
    import ctypes as C
    lib = L.load()
    lib.pm355_mul_mat_q_small.restype = C.c_int
    lib.pm355_mul_mat_q_small.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if xq is None:
        x = x.contiguous().view(-1, w.K)
        n_tokens = x.shape[0]
    y = torch.empty((n_tokens, w.N), dtype=torch.float32, device=w.data.device)
    check(lib.pm355_mul_mat_q_small(w.type, ptr(w.data), w.K, w.N, ptr(xq), ptr(x) if xq is None else None, n_tokens, ptr(y), ptr(bias), ptr(resid),
                                    stream_ptr()), "mul_mat_q_small")
    return y


def mul_mat_small_multi(ws, xq, n_tokens, biases=None):
    """2 or 3 matrices of one type and K sharing the pre-quantized activations, one launch (mmq_i8.hip multi-job kernels) -> list of f32 [T, N_j]."""
    import ctypes as C
    lib = L.load()
    n = len(ws)
    lib.pm355_mul_mat_q_small_multi.restype = C.c_int
    lib.pm355_mul_mat_q_small_multi.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    ys = [torch.empty((n_tokens, w.N), dtype=torch.float32, device=w.data.device) for w in ws]
    Wp = (C.c_void_p * n)(*[ptr(w.data) for w in ws]); Np = (C.c_int64 * n)(*[w.N for w in ws]); Yp = (C.c_void_p * n)(*[ptr(y) for y in ys])
    Bp = (C.c_void_p * n)(*[ptr(b) for b in biases]) if biases else None
    check(lib.pm355_mul_mat_q_small_multi(ws[0].type, n, Wp, Np, Yp, Bp, ptr(xq), ws[0].K, n_tokens, stream_ptr()), "mul_mat_q_small_multi")
    return ys


def mul_mat_small_mixed(ws, xq, n_tokens, biases=None):
    """wq | wk of one type and wv of another sharing the pre-quantized activations, one grid (mmq_i8.hip mmq_i8_dual_kernel) -> list of f32 [T, N_j]."""
    import ctypes as C
    lib = L.load()
    n = len(ws)
    lib.pm355_mul_mat_q_small_mixed.restype = C.c_int
    lib.pm355_mul_mat_q_small_mixed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    ys = [torch.empty((n_tokens, w.N), dtype=torch.float32, device=w.data.device) for w in ws]
    Tp = (C.c_int * n)(*[w.type for w in ws])
    Wp = (C.c_void_p * n)(*[ptr(w.data) for w in ws]); Np = (C.c_int64 * n)(*[w.N for w in ws]); Yp = (C.c_void_p * n)(*[ptr(y) for y in ys])
    Bp = (C.c_void_p * n)(*[ptr(b) for b in biases]) if biases else None
    check(lib.pm355_mul_mat_q_small_mixed(Tp, n, Wp, Np, Yp, Bp, ptr(xq), ws[0].K, n_tokens, stream_ptr()), "mul_mat_q_small_mixed")
    return ys


def mul_mat_i8(w, x, bias=None, resid=None):
    """Prompt-sized batches on the integer matrix cores (mmq_big.hip): x f32 [T, K] -> Q8_K on device -> f32 [T, N]."""
    import ctypes as C
    lib = L.load()
    lib.pm355_mul_mat_q_i8.restype = C.c_int
    lib.pm355_mul_mat_q_i8.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    x2 = x.contiguous().view(-1, w.K)
    y = torch.empty((x2.shape[0], w.N), dtype=torch.float32, device=x.device)
    check(lib.pm355_mul_mat_q_i8(w.type, ptr(w.data), w.K, w.N, ptr(x2), x2.shape[0], ptr(y), ptr(bias), ptr(resid), stream_ptr()), "mul_mat_q_i8")
    return y


def mul_mat_mfma(w, x, bias=None, resid=None):
    """Batched GEMM on MFMA: x f32 [T, K] -> f32 [T, N]."""
    import ctypes as C
    lib = L.load()
    lib.pm355_mul_mat_q_mfma.restype = C.c_int
    lib.pm355_mul_mat_q_mfma.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    x2 = x.contiguous().view(-1, w.K)
    y = torch.empty((x2.shape[0], w.N), dtype=torch.float32, device=x.device)
    check(lib.pm355_mul_mat_q_mfma(w.type, ptr(w.data), w.K, w.N, ptr(x2), x2.shape[0], ptr(y), ptr(bias), ptr(resid), stream_ptr()),
          "mul_mat_q_mfma")
    return y


class GemmJob(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("type", _C.c_int32), ("N", _C.c_int32), ("W", _C.c_void_p), ("y", _C.c_void_p), ("bias", _C.c_void_p), ("resid", _C.c_void_p)]


def mul_mat_mfma_multi(ws, x, biases=None, resids=None):
    """Several matrices over the same activations as ONE launch of the prompt GEMM (mmq_pf.hip): x f32 [T, K] -> list of f32 [T, N_j]."""
    import ctypes as C
    lib = L.load()
    lib.pm355_mul_mat_q_mfma_multi.restype = C.c_int
    lib.pm355_mul_mat_q_mfma_multi.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    K = ws[0].K
    x2 = x.contiguous().view(-1, K)
    ys = [torch.empty((x2.shape[0], w.N), dtype=torch.float32, device=x.device) for w in ws]
    jobs = (GemmJob * len(ws))()
    for j, w in enumerate(ws):
        jobs[j].type, jobs[j].N, jobs[j].W, jobs[j].y = w.type, w.N, ptr(w.data), ptr(ys[j])
        jobs[j].bias = ptr(biases[j]) if biases else None
        jobs[j].resid = ptr(resids[j]) if resids else None
    check(lib.pm355_mul_mat_q_mfma_multi(C.addressof(jobs), len(ws), K, ptr(x2), x2.shape[0], stream_ptr()), "mul_mat_q_mfma_multi")
    return ys


def mul_mat_mfma_pair(w_gate, w_up, x):
    """silu(W_gate . x) * (W_up . x) as one launch of pair tiles (mmq_pf.hip): x f32 [T, K] -> f32 [T, N]."""
    import ctypes as C
    lib = L.load()
    lib.pm355_mul_mat_q_mfma_pair.restype = C.c_int
    lib.pm355_mul_mat_q_mfma_pair.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    x2 = x.contiguous().view(-1, w_gate.K)
    y = torch.empty((x2.shape[0], w_gate.N), dtype=torch.float32, device=x.device)
    check(lib.pm355_mul_mat_q_mfma_pair(w_gate.type, ptr(w_gate.data), ptr(w_up.data), w_gate.K, w_gate.N, ptr(x2), x2.shape[0], ptr(y), stream_ptr()),
          "mul_mat_q_mfma_pair")
    return y


class QkvStore(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("rope_table", _C.c_void_p), ("d_pos", _C.c_void_p), ("d_cell_nkv", _C.c_void_p), ("k_cache", _C.c_void_p),
                ("v_cache", _C.c_void_p), ("n_head_kv", _C.c_int32), ("head_dim", _C.c_int32), ("n_ctx", _C.c_int32),
                ("n_rot", _C.c_int32), ("v_rowmajor", _C.c_int32), ("pad_", _C.c_int32)]


def rope_table(pos, head_dim, freq_factors=None, mode=0, n_ctx_orig=8192, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0,
               attn_factor=1.0, beta_fast=32.0, beta_slow=1.0, n_dims=None):
    """This token's (cos, sin) per rotation pair: f32 [n_dims] = c0, s0, c1, s1, ... (ggml_rope_cache_init). pos: int32 device tensor [1]."""
    import ctypes as C
    lib = L.load()
    lib.pm355_rope_table.restype = C.c_int
    lib.pm355_rope_table.argtypes = [C.c_void_p] * 5
    rp = RopeParams(n_dims or head_dim, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow)
    tab = torch.empty(n_dims or head_dim, dtype=torch.float32, device=pos.device)
    check(lib.pm355_rope_table(C.addressof(rp), ptr(pos), ptr(freq_factors), ptr(tab), stream_ptr()), "rope_table")
    return tab


def mul_mat_vec_qkv(ws, x, tab, pos, k_cache, v_cache, n_head_kv, head_dim, n_ctx, norm_w=None, eps=0.0, biases=None, n_rot=None,
                    v_rowmajor=False, cell_nkv=None, neox=False):
    """wq | wk | wv mat-vecs with RoPE + F16 KV store in the epilogue: returns the rotated, F16-rounded query row [N_q] (f32);
    the K row / V column of cache cell pos[0] (or cell_nkv[0]) are written."""
    import ctypes as C
    lib = L.load()
    lib.pm355_mul_mat_vec_qkv.restype = C.c_int
    lib.pm355_mul_mat_vec_qkv.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    jobs = (MatvecJob * 3)()
    q = torch.empty(ws[0].N, dtype=torch.float32, device=x.device)
    for j, w in enumerate(ws):
        jobs[j] = MatvecJob(w.type, 0, w.N, ptr(w.data), None, ptr(q) if j == 0 else None,
                            ptr(biases[j]) if biases and biases[j] is not None else None, None)
    s = QkvStore(ptr(tab), ptr(pos), ptr(cell_nkv), ptr(k_cache), ptr(v_cache), n_head_kv, head_dim, n_ctx, n_rot or head_dim,
                 int(v_rowmajor), int(neox))
    check(lib.pm355_mul_mat_vec_qkv(C.addressof(jobs), ws[0].K, ptr(x), ptr(norm_w), float(eps), C.addressof(s), stream_ptr()),
          "mul_mat_vec_qkv")
    return q


ATTN_V_ROWMAJOR, ATTN_MASK_F16, ATTN_K_Q8_0, ATTN_V_Q8_0 = 1, 2, 4, 8


class AttnTokenArgs(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("q", _C.c_void_p), ("k", _C.c_void_p), ("v", _C.c_void_p), ("k_cache", _C.c_void_p), ("v_cache", _C.c_void_p), ("d_pos", _C.c_void_p),
                ("d_cell_nkv", _C.c_void_p), ("mask", _C.c_void_p), ("freq_factors", _C.c_void_p), ("out", _C.c_void_p), ("scratch", _C.c_void_p),
                ("n_head", _C.c_int32), ("n_head_kv", _C.c_int32), ("head_dim", _C.c_int32), ("n_ctx", _C.c_int32), ("split", _C.c_int32), ("max_keys", _C.c_int32),
                ("kq_scale", _C.c_float), ("flags", _C.c_int32)]


def attn_token(q, k, v, k_cache, v_cache, pos, cell_nkv, n_head, n_head_kv, head_dim, n_ctx, scale, flags=0, mask=None, freq_factors=None, scratch=None,
               max_keys=0, mode=0, freq_base=10000.0, n_dims=None):
    """pm355_attn_token: one token's RoPE + KV store + attention (the plug-in's ggml-graph mode: cell_nkv = int32 {cache cell, cells attended}); raw q / k / v
    projections in, attention output [n_head * head_dim] out. scratch != None: the long-context (split) form."""
    import ctypes as C
    lib = L.load()
    lib.pm355_attn_token.restype = C.c_int
    lib.pm355_attn_token.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    out = torch.empty(n_head * head_dim, dtype=torch.float32, device=q.device)
    a = AttnTokenArgs(ptr(q), ptr(k), ptr(v), ptr(k_cache), ptr(v_cache), ptr(pos), ptr(cell_nkv), ptr(mask), ptr(freq_factors), ptr(out), ptr(scratch),
                      n_head, n_head_kv, head_dim, n_ctx, 1 if scratch is not None else 0, max_keys, float(scale), flags)
    rp = RopeParams(n_dims or head_dim, mode, 8192, freq_base, 1.0, 0.0, 1.0, 32.0, 1.0)
    check(lib.pm355_attn_token(C.addressof(a), C.addressof(rp), stream_ptr()), "attn_token")
    return out


class QkvAttn(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("out", _C.c_void_p), ("ticket", _C.c_void_p), ("watchdog", _C.c_void_p), ("kq_scale", _C.c_float), ("n_head", _C.c_int32), ("max_keys", _C.c_int32)]


def mul_mat_vec_qkv_attn(ws, x, tab, pos, k_cache, v_cache, n_head, n_head_kv, head_dim, n_ctx, scale, ticket, norm_w=None, eps=0.0, biases=None, n_rot=None,
                         neox=False, max_keys=0, watchdog=None):
    """mul_mat_vec_qkv with the attention over the cached cells in the launch's tail (pm355_mul_mat_vec_qkv_attn). ticket: zeroed uint32-sized int32 tensor
    [n_head_kv] that lives across calls. Returns (q [N_q], attention output [N_q])."""
    import ctypes as C
    lib = L.load()
    lib.pm355_mul_mat_vec_qkv_attn.restype = C.c_int
    lib.pm355_mul_mat_vec_qkv_attn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    jobs = (MatvecJob * 3)()
    q = torch.empty(ws[0].N, dtype=torch.float32, device=x.device)
    out = torch.empty(ws[0].N, dtype=torch.float32, device=x.device)
    for j, w in enumerate(ws):
        jobs[j] = MatvecJob(w.type, 0, w.N, ptr(w.data), None, ptr(q) if j == 0 else None,
                            ptr(biases[j]) if biases and biases[j] is not None else None, None)
    s = QkvStore(ptr(tab), ptr(pos), None, ptr(k_cache), ptr(v_cache), n_head_kv, head_dim, n_ctx, n_rot or head_dim, 0, int(neox))
    at = QkvAttn(ptr(out), ptr(ticket), ptr(watchdog), float(scale), n_head, max_keys)
    check(lib.pm355_mul_mat_vec_qkv_attn(C.addressof(jobs), ws[0].K, ptr(x), ptr(norm_w), float(eps), C.addressof(s), None, 0, C.addressof(at), stream_ptr()),
          "mul_mat_vec_qkv_attn")
    return q, out


def attn_cached(q_rot, k_cache, v_cache, pos, n_head, n_head_kv, head_dim, n_ctx, scale, cell_nkv=None, mask=None, max_keys=0, flags=0, scratch=None):
    """Single-token attention over cells that are all in the cache (q rotated and F16-rounded). pos: int32 device tensor [1].
    scratch (zero-initialised f32 tensor of attn_split_scratch_floats): the long-context matrix-core kernel, max_keys = grid cells."""
    import ctypes as C
    lib = L.load()
    if scratch is not None:
        lib.pm355_attn_cached_long.restype = C.c_int
        lib.pm355_attn_cached_long.argtypes = [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_int, C.c_void_p]
        out = torch.empty_like(q_rot)
        check(lib.pm355_attn_cached_long(ptr(q_rot), ptr(k_cache), ptr(v_cache), ptr(pos), ptr(cell_nkv), ptr(mask), ptr(out), ptr(scratch), n_head,
                                         n_head_kv, head_dim, n_ctx, float(scale), max_keys, flags, stream_ptr()), "attn_cached_long")
        return out
    lib.pm355_attn_cached.restype = C.c_int
    lib.pm355_attn_cached.argtypes = [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_int, C.c_void_p]
    out = torch.empty_like(q_rot)
    check(lib.pm355_attn_cached(ptr(q_rot), ptr(k_cache), ptr(v_cache), ptr(pos), ptr(cell_nkv), ptr(mask), ptr(out), n_head, n_head_kv,
                                head_dim, n_ctx, float(scale), max_keys, flags, stream_ptr()), "attn_cached")
    return out


class EnginePhase(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("kind", _C.c_int32), ("njobs", _C.c_int32), ("K", _C.c_int64),
                ("jobs", _C.c_void_p), ("x_f32", _C.c_void_p), ("norm_w", _C.c_void_p), ("eps", _C.c_float), ("n_sumsq_in", _C.c_int32),
                ("sumsq_out", _C.c_void_p), ("sumsq_in", _C.c_void_p), ("qkv", _C.c_void_p),
                ("q_rot", _C.c_void_p), ("k_cache", _C.c_void_p), ("v_cache", _C.c_void_p), ("d_pos", _C.c_void_p), ("out", _C.c_void_p),
                ("n_head", _C.c_int32), ("n_head_kv", _C.c_int32), ("head_dim", _C.c_int32), ("n_ctx", _C.c_int32), ("max_keys", _C.c_int32),
                ("kq_scale", _C.c_float)]


class EngineRun:
    """A phase list for pm355_engine_run (the persistent decode engine, csrc/decode_engine.hip). matvec() / attention() append phases and return the
    output tensors; run() executes them as ONE launch."""

    def __init__(self):
        self.phases, self.keep = [], []

    def matvec(self, ws, x, norm_w=None, eps=0.0, w2s=None, biases=None, resids=None, sumsq_in=None, want_sumsq=False, qkv=None):
        """qkv = dict(tab=, pos=, k_cache=, v_cache=, n_head_kv=, head_dim=, n_ctx=, n_rot=, neox=) for the wq | wk | wv phase with RoPE + KV store."""
        import ctypes as C
        n = len(ws)
        jobs = (MatvecJob * n)()
        ys = []
        for j, w in enumerate(ws):
            y = torch.zeros(w.N, dtype=torch.float32, device=w.data.device)
            ys.append(y)
            jobs[j] = MatvecJob(w.type, 0, w.N, ptr(w.data), ptr(w2s[j].data) if w2s else None, ptr(y),
                                ptr(biases[j]) if biases and biases[j] is not None else None, ptr(resids[j]) if resids and resids[j] is not None else None)
        ss = torch.zeros(256, dtype=torch.float64, device=x.device) if want_sumsq else None
        st = None
        if qkv is not None:
            st = QkvStore(ptr(qkv["tab"]), ptr(qkv["pos"]), None, ptr(qkv["k_cache"]), ptr(qkv["v_cache"]), qkv["n_head_kv"], qkv["head_dim"], qkv["n_ctx"],
                          qkv.get("n_rot") or qkv["head_dim"], 0, int(qkv.get("neox", False)))
        ph = EnginePhase(0, n, ws[0].K, C.addressof(jobs), ptr(x), ptr(norm_w), float(eps), 0 if sumsq_in is None else sumsq_in.numel(), ptr(ss), ptr(sumsq_in),
                         C.addressof(st) if st is not None else None, None, None, None, None, None, 0, 0, 0, 0, 0, 0.0)
        self.phases.append(ph)
        self.keep += [jobs, ys, ss, st, x, norm_w, sumsq_in, biases, resids]
        return (ys, ss) if want_sumsq else ys

    def attention(self, q_rot, k_cache, v_cache, pos, n_head, n_head_kv, head_dim, n_ctx, scale, max_keys=0):
        out = torch.zeros(n_head * head_dim, dtype=torch.float32, device=q_rot.device)
        self.phases.append(EnginePhase(1, 0, 0, None, None, None, 0.0, 0, None, None, None, ptr(q_rot), ptr(k_cache), ptr(v_cache), ptr(pos), ptr(out),
                                       n_head, n_head_kv, head_dim, n_ctx, max_keys, float(scale)))
        self.keep += [out, q_rot, k_cache, v_cache, pos]
        return out

    def run(self):
        import ctypes as C
        lib = L.load()
        lib.pm355_engine_run.restype = C.c_int
        lib.pm355_engine_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        arr = (EnginePhase * len(self.phases))(*self.phases)
        check(lib.pm355_engine_run(C.addressof(arr), len(self.phases), stream_ptr()), "engine_run")
