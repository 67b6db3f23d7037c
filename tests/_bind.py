"""ctypes bindings used by the tests ONLY: the CPU oracle (oracle/liboracle.so, our restatement)
and, when present, the unmodified reference compiled under oracle/_ref/ (see oracle/Makefile).

Nothing in prima_cpp_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

F32, F16, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 8, 12, 13, 14, 15
TYPE_NAMES = {F32: "f32", F16: "f16", Q8_0: "q8_0", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", Q8_K: "q8_K"}
BLOCK = {Q8_0: (32, 34), Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210), Q8_K: (256, 292),
         F16: (1, 2), F32: (1, 4)}
QUANT_TYPES = [Q4_K, Q5_K, Q6_K, Q8_0]


def row_size(t, k):
    n, b = BLOCK[t]
    assert k % n == 0
    return k // n * b


def vec_dot_type(t):
    return {F32: F32, F16: F16, Q8_0: Q8_0}.get(t, Q8_K)


_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None else None


class TensorT(C.Structure):
    _fields_ = [("type", C.c_int32), ("pad_", C.c_int32), ("data", C.c_void_p)]


class ModelDesc(C.Structure):
    """Field order shared by orc_model_desc / ref_model_desc / pm355_model_desc."""
    _fields_ = [("arch", C.c_int32), ("n_layer", C.c_int32), ("n_embd", C.c_int32), ("n_head", C.c_int32),
                ("n_head_kv", C.c_int32), ("head_dim", C.c_int32), ("n_ff", C.c_int32), ("n_vocab", C.c_int32),
                ("n_ctx", C.c_int32), ("n_ctx_orig", C.c_int32),
                ("rms_eps", C.c_float), ("rope_freq_base", C.c_float), ("rope_freq_scale", C.c_float),
                ("pad_", C.c_int32),
                ("attn_norm", C.POINTER(TensorT)), ("wq", C.POINTER(TensorT)), ("wk", C.POINTER(TensorT)),
                ("wv", C.POINTER(TensorT)), ("wo", C.POINTER(TensorT)), ("ffn_norm", C.POINTER(TensorT)),
                ("ffn_gate", C.POINTER(TensorT)), ("ffn_up", C.POINTER(TensorT)), ("ffn_down", C.POINTER(TensorT)),
                ("bq", C.POINTER(TensorT)), ("bk", C.POINTER(TensorT)), ("bv", C.POINTER(TensorT)),
                ("tok_embd", TensorT), ("out_norm", TensorT), ("output", TensorT),
                ("rope_freqs", C.c_void_p)]


def build_oracle():
    """Compile oracle/liboracle.so (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True,
                   stdout=subprocess.DEVNULL)


class _Lib:
    """Common wrapper: oracle ('orc_') and reference harness ('ref_') expose the same op API."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path

    def fn(self, name, restype=None, argtypes=None):
        f = getattr(self.lib, name)
        f.restype = restype
        if argtypes is not None:
            f.argtypes = argtypes
        return f

    # --- ops with identical signatures in both libs -------------------------------------
    def mul_mat(self, t, W, K, N, x, n_threads=1):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, K)
        out = np.empty((x.shape[0], N), dtype=np.float32)
        if self.prefix == "ref_":
            f = self.fn("ref_mul_mat", C.c_int, [C.c_int, _vp, C.c_int64, C.c_int64, _vp, C.c_int64, _vp, C.c_int])
            assert f(t, _ptr(W), K, N, _ptr(x), x.shape[0], _ptr(out), n_threads) == 0
        else:
            f = self.fn("orc_mul_mat", None, [C.c_int, _vp, C.c_int64, C.c_int64, _vp, C.c_int64, _vp])
            f(t, _ptr(W), K, N, _ptr(x), x.shape[0], _ptr(out))
        return out

    def rms_norm(self, x, w, eps):
        x = np.ascontiguousarray(x, dtype=np.float32)
        rows, n = x.reshape(-1, x.shape[-1]).shape
        out = np.empty_like(x)
        f = self.fn(self.prefix + "rms_norm", None if self.prefix == "orc_" else C.c_int,
                    [_vp, _vp, C.c_int64, C.c_int64, C.c_float, _vp])
        w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
        f(_ptr(x), _ptr(w), n, rows, eps, _ptr(out))
        return out

    def rope(self, x, pos, freq_factors=None, n_dims=None, mode=0, n_ctx_orig=8192, freq_base=10000.0,
             freq_scale=1.0, ext_factor=0.0, attn_factor=1.0, beta_fast=32.0, beta_slow=1.0):
        x = np.ascontiguousarray(x, dtype=np.float32)        # [ntok, heads, d]
        ntok, heads, d = x.shape
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        ff = None if freq_factors is None else np.ascontiguousarray(freq_factors, dtype=np.float32)
        out = np.empty_like(x)
        f = self.fn(self.prefix + "rope", None if self.prefix == "orc_" else C.c_int,
                    [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, C.c_int, C.c_int, C.c_int,
                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp])
        f(_ptr(x), d, heads, ntok, _ptr(pos), _ptr(ff), n_dims or d, mode, n_ctx_orig, freq_base, freq_scale,
          ext_factor, attn_factor, beta_fast, beta_slow, _ptr(out))
        return out

    def soft_max_ext(self, x, mask, scale, max_bias=0.0):
        x = np.ascontiguousarray(x, dtype=np.float32)        # [heads, nr, nc]
        heads, nr, nc = x.shape
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.float32)
        out = np.empty_like(x)
        f = self.fn(self.prefix + "soft_max_ext", None if self.prefix == "orc_" else C.c_int,
                    [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_float, _vp])
        f(_ptr(x), _ptr(m), nc, nr, heads, scale, max_bias, _ptr(out))
        return out

    def silu_mul(self, g, u):
        g = np.ascontiguousarray(g, dtype=np.float32)
        u = None if u is None else np.ascontiguousarray(u, dtype=np.float32)
        out = np.empty_like(g)
        f = self.fn(self.prefix + "silu_mul", None if self.prefix == "orc_" else C.c_int, [_vp, _vp, C.c_int64, _vp])
        f(_ptr(g), _ptr(u), g.size, _ptr(out))
        return out

    # --- model -----------------------------------------------------------------------------
    def model_new(self, desc):
        f = self.fn(self.prefix + "model_new", _vp, [C.POINTER(ModelDesc)])
        h = f(C.byref(desc))
        assert h
        return h

    def model_free(self, h):
        self.fn(self.prefix + "model_free", None, [_vp])(h)

    def model_kv_clear(self, h):
        self.fn(self.prefix + "model_kv_clear", None, [_vp])(h)

    def model_eval(self, h, desc, tokens=None, embd=None, pos0=0, layer_lo=0, layer_hi=None, with_head=True,
                   n_threads=4):
        layer_hi = desc.n_layer if layer_hi is None else layer_hi
        if tokens is not None:
            tokens = np.ascontiguousarray(tokens, dtype=np.int32)
            T = tokens.size
        else:
            embd = np.ascontiguousarray(embd, dtype=np.float32)
            T = embd.shape[0]
        hidden = np.empty((T, desc.n_embd), dtype=np.float32)
        logits = np.empty(desc.n_vocab, dtype=np.float32) if with_head else None
        f = self.fn(self.prefix + "model_eval", C.c_int,
                    [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int])
        rc = f(h, _ptr(tokens), _ptr(embd), T, pos0, layer_lo, layer_hi, int(with_head), _ptr(hidden),
               _ptr(logits), n_threads)
        assert rc == 0, rc
        return hidden, logits

    def model_kv(self, h, desc, il, which):
        f = self.fn(self.prefix + "model_kv_ptr", _vp, [_vp, C.c_int, C.c_int])
        p = f(h, il, which)
        n = desc.head_dim * desc.n_head_kv * desc.n_ctx
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), shape=(n,)).copy()


class Oracle(_Lib):
    def __init__(self):
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        super().__init__(path, "orc_")

    def quantize_row_q8_K(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(row_size(Q8_K, x.size), dtype=np.uint8)
        self.fn("orc_quantize_row_q8_K", None, [_vp, _vp, C.c_int64])(_ptr(x), _ptr(out), x.size)
        return out

    def quantize_row_q8_0(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(row_size(Q8_0, x.size), dtype=np.uint8)
        self.fn("orc_quantize_row_q8_0", None, [_vp, _vp, C.c_int64])(_ptr(x), _ptr(out), x.size)
        return out

    def quantize_act(self, t, x):
        vdt = vec_dot_type(t)
        if vdt == Q8_K:
            return self.quantize_row_q8_K(x)
        if vdt == Q8_0:
            return self.quantize_row_q8_0(x)
        raise ValueError(t)

    def dequantize_row(self, t, blocks, k):
        out = np.empty(k, dtype=np.float32)
        self.fn("orc_dequantize_row", None, [C.c_int, _vp, _vp, C.c_int64])(t, _ptr(blocks), _ptr(out), k)
        return out

    def vec_dot(self, t, n, w, a):
        return self.fn("orc_vec_dot", C.c_float, [C.c_int, C.c_int64, _vp, _vp])(t, n, _ptr(w), _ptr(a))

    def int_partials(self, t, n, w, a):
        nb = n // BLOCK[t][0]
        isum = np.empty(nb, dtype=np.int32)
        msum = np.empty(nb, dtype=np.int32)
        self.fn("orc_vec_dot_int_partials", None, [C.c_int, C.c_int64, _vp, _vp, _vp, _vp])(
            t, n, _ptr(w), _ptr(a), _ptr(isum), _ptr(msum))
        return isum, msum

    def f32_to_f16(self, x):
        f = self.fn("orc_f32_to_f16", C.c_uint16, [C.c_float])
        return np.array([f(float(v)) for v in np.asarray(x, dtype=np.float32).ravel()], dtype=np.uint16)

    def f16_to_f32(self, h):
        f = self.fn("orc_f16_to_f32", C.c_float, [C.c_uint16])
        return np.array([f(int(v)) for v in np.asarray(h, dtype=np.uint16).ravel()], dtype=np.float32)


def ref_path(flavour):
    return os.path.join(ORACLE_DIR, "_ref", f"libggml_ref_{flavour}.so")


def have_ref(flavour="scalar"):
    return os.path.exists(ref_path(flavour))


def best_ref_flavour():
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    need512 = ["avx512f", "avx512bw", "avx512dq", "avx512vl", "avx512cd", "avx512_vnni"]
    if all(x in flags for x in need512) and have_ref("avx512"):
        return "avx512"
    if "avx2" in flags and "fma" in flags and "f16c" in flags and have_ref("avx2"):
        return "avx2"
    return "scalar" if have_ref("scalar") else None


class Ref(_Lib):
    """The unmodified reference (ggml CPU backend) + our ref_ops.c harness."""

    def __init__(self, flavour="scalar"):
        super().__init__(ref_path(flavour), "ref_")
        self.flavour = flavour
        # ggml needs its fp16 tables initialised before raw quantize calls (ggml_init does it: ggml.c:3478)
        class IP(C.Structure):
            _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]
        init = self.fn("ggml_init", _vp, [IP])
        ctx = init(IP(1 << 20, None, False))
        self.fn("ggml_free", None, [_vp])(ctx)

    def quantize_row_q8_K(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(row_size(Q8_K, x.size), dtype=np.uint8)
        self.fn("quantize_row_q8_K", None, [_vp, _vp, C.c_int64])(_ptr(x), _ptr(out), x.size)
        return out

    def quantize_row_q8_0(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(row_size(Q8_0, x.size), dtype=np.uint8)
        self.fn("quantize_row_q8_0", None, [_vp, _vp, C.c_int64])(_ptr(x), _ptr(out), x.size)
        return out

    def quantize_weights(self, t, w):
        """ggml_quantize_chunk (ggml.c:21826): f32 [nrows, k] -> packed blocks."""
        w = np.ascontiguousarray(w, dtype=np.float32)
        nrows, k = w.shape
        out = np.zeros(nrows * row_size(t, k), dtype=np.uint8)
        f = self.fn("ggml_quantize_chunk", C.c_size_t, [C.c_int, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp])
        n = f(t, _ptr(w), _ptr(out), 0, nrows, k, None)
        assert n == out.size
        return out

    def dequantize_row(self, t, blocks, k):
        out = np.empty(k, dtype=np.float32)
        name = {Q4_K: "dequantize_row_q4_K", Q5_K: "dequantize_row_q5_K", Q6_K: "dequantize_row_q6_K",
                Q8_0: "dequantize_row_q8_0"}[t]
        self.fn(name, None, [_vp, _vp, C.c_int64])(_ptr(blocks), _ptr(out), k)
        return out

    def vec_dot(self, t, n, w, a):
        name = {Q4_K: "ggml_vec_dot_q4_K_q8_K", Q5_K: "ggml_vec_dot_q5_K_q8_K", Q6_K: "ggml_vec_dot_q6_K_q8_K",
                Q8_0: "ggml_vec_dot_q8_0_q8_0"}[t]
        s = C.c_float(0)
        self.fn(name, None, [C.c_int, _fp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int])(
            n, C.byref(s), 0, _ptr(w), 0, _ptr(a), 0, 1)
        return s.value

    def fp32_to_fp16_row(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(x.size, dtype=np.uint16)
        self.fn("ggml_fp32_to_fp16_row", None, [_vp, _vp, C.c_int64])(_ptr(x), _ptr(out), x.size)
        return out

    def fp16_to_fp32_row(self, h):
        h = np.ascontiguousarray(h, dtype=np.uint16)
        out = np.empty(h.size, dtype=np.float32)
        self.fn("ggml_fp16_to_fp32_row", None, [_vp, _vp, C.c_int64])(_ptr(h), _ptr(out), h.size)
        return out

    def time_layer_matvecs(self, mats, n_threads, reps=3):
        """mats: list of (type, K, N, uint8 array). Returns seconds per pass."""
        n = len(mats)
        types = (C.c_int32 * n)(*[m[0] for m in mats])
        Ks = (C.c_int64 * n)(*[m[1] for m in mats])
        Ns = (C.c_int64 * n)(*[m[2] for m in mats])
        datas = (C.c_void_p * n)(*[m[3].ctypes.data for m in mats])
        f = self.fn("ref_time_layer_matvecs", C.c_double, [C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int])
        return f(n, types, Ks, Ns, datas, n_threads, reps)

    def build_info(self):
        return self.fn("ref_build_info", C.c_char_p, [])().decode()


# --------------------------------------------------------------------------------------------
# synthetic data
# --------------------------------------------------------------------------------------------

_POOL = None


def rand_blocks(t, nrows, k, rng, scale=None):
    """Random VALID quant blocks (every bit pattern of the packed fields, finite fp16 scales).

    Exercises all nibble/high-bit/6-bit-scale paths harder than quantized gaussians do.
    `scale` sets the magnitude of the fp16 super-block scale d (default gives |w| ~ 1/sqrt(k))."""
    nper, bs = BLOCK[t]
    nb = nrows * (k // nper)
    if nb * bs >= (1 << 24):      # big tensors (model-shaped tests): bytes from a fixed random pool, read cyclically from a random odd offset
        global _POOL
        if _POOL is None:
            _POOL = np.random.default_rng(0x9E3779B9).integers(0, 2 ** 64 - 1, size=((32 << 20) + 8072) // 8, dtype=np.uint64, endpoint=True).view(np.uint8)
        n, off = nb * bs, int(rng.integers(0, _POOL.size // 2)) | 1
        raw = np.empty(n, dtype=np.uint8)
        done = 0
        while done < n:                  # cyclic read as slice copies (np.tile on a 1-D array is ~50x slower than memcpy)
            take = min(n - done, _POOL.size - off)
            raw[done:done + take] = _POOL[off:off + take]
            done, off = done + take, 0
        raw = raw.reshape(nb, bs)
    else:
        raw = rng.integers(0, 256, size=(nb, bs), dtype=np.uint8)
    if scale is None:
        scale = 1.0 / np.sqrt(k)

    def f16(vals):
        return np.asarray(vals, dtype=np.float16).view(np.uint8).reshape(nb, 2)
    if t == Q8_0:
        raw[:, 0:2] = f16(rng.uniform(0.5, 1.5, nb) * scale / 64.0)
    elif t in (Q4_K, Q5_K):
        qmax = 15 if t == Q4_K else 31
        raw[:, 0:2] = f16(rng.uniform(0.5, 1.5, nb) * scale / (qmax * 32.0))
        raw[:, 2:4] = f16(rng.uniform(0.5, 1.5, nb) * scale / (2 * 32.0))
    elif t == Q6_K:
        raw[:, 208:210] = f16(rng.uniform(0.5, 1.5, nb) * scale / (32.0 * 64.0))
    else:
        raise ValueError(t)
    return raw.reshape(-1)


def tiny_model(rng, arch=0, n_layer=2, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64,
               types=None, quantize=None, rope_freqs=False):
    """Build a small Llama/Qwen2-shaped model with the Q4_K_M-style type mixture.

    quantize(t, w_f32[rows, k]) -> uint8 blocks; defaults to random valid blocks when None."""
    head_dim = n_embd // n_head
    types = types or {}
    keep = []

    def mk(name, k, n, il=0):
        t = types.get(name, Q4_K)
        if quantize is not None:
            w = rng.normal(0, 1.0 / np.sqrt(k), size=(n, k)).astype(np.float32)
            data = quantize(t, w)
        else:
            data = rand_blocks(t, n, k, rng)
        keep.append(data)
        return TensorT(t, 0, data.ctypes.data)

    def mkf(n, mean=1.0, std=0.02):
        data = (mean + rng.normal(0, std, size=n)).astype(np.float32)
        keep.append(data)
        return TensorT(F32, 0, data.ctypes.data)

    def arr(items):
        a = (TensorT * len(items))(*items)
        keep.append(a)
        return a

    d = ModelDesc()
    d.arch, d.n_layer, d.n_embd, d.n_head, d.n_head_kv, d.head_dim = arch, n_layer, n_embd, n_head, n_head_kv, head_dim
    d.n_ff, d.n_vocab, d.n_ctx, d.n_ctx_orig = n_ff, n_vocab, n_ctx, 8192
    d.rms_eps = 1e-5 if arch == 0 else 1e-6
    d.rope_freq_base = 500000.0 if arch == 0 else 1000000.0
    d.rope_freq_scale = 1.0
    E, Eq, Ekv = n_embd, head_dim * n_head, head_dim * n_head_kv
    d.attn_norm = arr([mkf(E) for _ in range(n_layer)])
    d.wq = arr([mk("attn_q", E, Eq) for _ in range(n_layer)])
    d.wk = arr([mk("attn_k", E, Ekv) for _ in range(n_layer)])
    vt = [types.get("attn_v", [Q6_K, Q5_K][il % 2]) for il in range(n_layer)]
    d.wv = arr([_typed(rng, vt[il], E, Ekv, keep, quantize) for il in range(n_layer)])
    d.wo = arr([mk("attn_output", Eq, E) for _ in range(n_layer)])
    d.ffn_norm = arr([mkf(E) for _ in range(n_layer)])
    d.ffn_gate = arr([mk("ffn_gate", E, n_ff) for _ in range(n_layer)])
    d.ffn_up = arr([mk("ffn_up", E, n_ff) for _ in range(n_layer)])
    dt = [types.get("ffn_down", [Q6_K, Q4_K][il % 2]) for il in range(n_layer)]
    d.ffn_down = arr([_typed(rng, dt[il], n_ff, E, keep, quantize) for il in range(n_layer)])
    if arch == 1:
        d.bq = arr([mkf(Eq, 0.0, 0.1) for _ in range(n_layer)])
        d.bk = arr([mkf(Ekv, 0.0, 0.1) for _ in range(n_layer)])
        d.bv = arr([mkf(Ekv, 0.0, 0.1) for _ in range(n_layer)])
    d.tok_embd = _typed(rng, types.get("token_embd", Q4_K), E, n_vocab, keep, quantize, scale=1.0)
    d.out_norm = mkf(E)
    d.output = _typed(rng, types.get("output", Q6_K), E, n_vocab, keep, quantize)
    if rope_freqs:
        ff = (1.0 + rng.uniform(0, 7, size=head_dim // 2)).astype(np.float32)
        keep.append(ff)
        d.rope_freqs = ff.ctypes.data
    d._keep = keep
    return d


def _typed(rng, t, k, n, keep, quantize, scale=None):
    if quantize is not None:
        w = rng.normal(0, (scale or 1.0) / np.sqrt(k), size=(n, k)).astype(np.float32)
        data = quantize(t, w)
    else:
        data = rand_blocks(t, n, k, rng, scale=None if scale is None else scale)
    keep.append(data)
    return TensorT(t, 0, data.ctypes.data)


# --------------------------------------------------------------------------------------------
# (de)serialisation of a ModelDesc so that golden fixtures are self-contained
# --------------------------------------------------------------------------------------------
_LAYER_FIELDS = ["attn_norm", "wq", "wk", "wv", "wo", "ffn_norm", "ffn_gate", "ffn_up", "ffn_down", "bq", "bk", "bv"]
_HP_FIELDS = ["arch", "n_layer", "n_embd", "n_head", "n_head_kv", "head_dim", "n_ff", "n_vocab", "n_ctx", "n_ctx_orig",
              "rms_eps", "rope_freq_base", "rope_freq_scale"]


def _tensor_shape(d, field):
    E, Eq, Ekv, F = d.n_embd, d.head_dim * d.n_head, d.head_dim * d.n_head_kv, d.n_ff
    return {"attn_norm": (E, 1), "wq": (E, Eq), "wk": (E, Ekv), "wv": (E, Ekv), "wo": (Eq, E), "ffn_norm": (E, 1),
            "ffn_gate": (E, F), "ffn_up": (E, F), "ffn_down": (F, E), "bq": (Eq, 1), "bk": (Ekv, 1), "bv": (Ekv, 1),
            "tok_embd": (E, d.n_vocab), "out_norm": (E, 1), "output": (E, d.n_vocab)}[field]


def _tensor_bytes(t, k, n):
    return np.ctypeslib.as_array(C.cast(t.data, C.POINTER(C.c_uint8)), shape=(row_size(t.type, k) * n,)).copy()


def desc_to_arrays(d):
    out = {f"hp_{f}": np.array(getattr(d, f)) for f in _HP_FIELDS}
    for f in _LAYER_FIELDS:
        arr = getattr(d, f)
        if not arr:
            continue
        for il in range(d.n_layer):
            k, n = _tensor_shape(d, f)
            out[f"t_{f}_{il}"] = _tensor_bytes(arr[il], k, n)
            out[f"y_{f}_{il}"] = np.array(arr[il].type)
    for f in ("tok_embd", "out_norm", "output"):
        t = getattr(d, f)
        k, n = _tensor_shape(d, f)
        out[f"t_{f}"] = _tensor_bytes(t, k, n)
        out[f"y_{f}"] = np.array(t.type)
    if d.rope_freqs:
        out["t_rope_freqs"] = np.ctypeslib.as_array(C.cast(d.rope_freqs, C.POINTER(C.c_float)), shape=(d.head_dim // 2,)).copy()
    return out


def desc_from_arrays(z):
    d = ModelDesc()
    keep = []
    for f in _HP_FIELDS:
        v = z[f"hp_{f}"]
        setattr(d, f, float(v) if f in ("rms_eps", "rope_freq_base", "rope_freq_scale") else int(v))

    def tt(key_t, key_y):
        data = np.ascontiguousarray(z[key_t])
        keep.append(data)
        return TensorT(int(z[key_y]), 0, data.ctypes.data)
    for f in _LAYER_FIELDS:
        if f"t_{f}_0" not in z:
            continue
        a = (TensorT * d.n_layer)(*[tt(f"t_{f}_{il}", f"y_{f}_{il}") for il in range(d.n_layer)])
        keep.append(a)
        setattr(d, f, a)
    for f in ("tok_embd", "out_norm", "output"):
        setattr(d, f, tt(f"t_{f}", f"y_{f}"))
    if "t_rope_freqs" in z:
        ff = np.ascontiguousarray(z["t_rope_freqs"], dtype=np.float32)
        keep.append(ff)
        d.rope_freqs = ff.ctypes.data
    d._keep = keep
    return d


# --------------------------------------------------------------------------------------------
# the reference's own driver (oracle/_ref/llama-ref-driver-*, built by oracle/Makefile from /root/reference with the
# committed gate patch) on GGUF files written by prima_cpp_amd/gguf.py
# --------------------------------------------------------------------------------------------
def write_gguf_from_arrays(path, z):
    """A golden fixture's model arrays (desc_to_arrays) -> GGUF file the reference loader reads."""
    import sys
    sys.path.insert(0, ROOT)
    from prima_cpp_amd import gguf as G
    hp = {f: z[f"hp_{f}"] for f in _HP_FIELDS}
    arch, L, E, H, Hkv, F, V, dh = (int(hp[k]) for k in ("arch", "n_layer", "n_embd", "n_head", "n_head_kv", "n_ff", "n_vocab", "head_dim"))
    Eq, Ekv = dh * H, dh * Hkv
    kv = G.model_kv(arch, L, E, H, Hkv, F, V, int(hp["n_ctx_orig"]), float(hp["rms_eps"]), float(hp["rope_freq_base"]))
    shapes = {"attn_norm": (E,), "wq": (E, Eq), "wk": (E, Ekv), "wv": (E, Ekv), "wo": (Eq, E), "ffn_norm": (E,),
              "ffn_gate": (E, F), "ffn_up": (E, F), "ffn_down": (F, E), "bq": (Eq,), "bk": (Ekv,), "bv": (Ekv,)}
    keys = set(z.files) if hasattr(z, "files") else set(z.keys())
    T = [("token_embd.weight", int(z["y_tok_embd"]), (E, V), z["t_tok_embd"])]
    for il in range(L):
        for f, nm in G.LAYER_TENSORS.items():
            if f"t_{f}_{il}" in keys:
                T.append((f"blk.{il}.{nm}", int(z[f"y_{f}_{il}"]), shapes[f], z[f"t_{f}_{il}"]))
    T.append(("output_norm.weight", F32, (E,), z["t_out_norm"]))
    T.append(("output.weight", int(z["y_output"]), (E, V), z["t_output"]))
    if "t_rope_freqs" in keys:
        T.append(("rope_freqs.weight", F32, (dh // 2,), z["t_rope_freqs"]))
    return G.write_gguf(path, kv, T)


def llama_driver_path(flavour=None):
    fl = flavour or best_ref_flavour()
    for f in ([fl] if fl in ("scalar", "avx2", "avx512") else []) + ([] if flavour else ["avx2"]):
        p = os.path.join(ORACLE_DIR, "_ref", f"llama-ref-driver-{f}")
        if os.path.exists(p):
            return p
    return None


def run_llama_driver(gguf_path, prompt, n_gen, ngl=0, n_ctx=64, threads=2, extra_args=(), force=None, env=None, timeout=600,
                     chunk=None, flavour=None, lout=False):
    """Greedy decode through the reference's llama_init_from_gpt_params + llama_decode. Returns (tokens, logits, stats).
    lout: also capture every layer's output row of the first single-token decode (the driver's REFDRV_LOUT, scheduler eval callback):
    stats["lout"] = {layer (-1 = result_norm): float32 row}."""
    import json
    import tempfile
    drv = llama_driver_path(flavour)
    assert drv, "oracle/_ref/llama-ref-driver-* not built"
    out = tempfile.NamedTemporaryFile(suffix=".refdrv", delete=False).name
    e = dict(os.environ)
    e.update({"REFDRV_PROMPT": ",".join(str(int(t)) for t in prompt), "REFDRV_NGEN": str(n_gen), "REFDRV_OUT": out})
    if force is not None:
        e["REFDRV_FORCE"] = ",".join(str(int(t)) for t in force)
    if chunk:
        e["REFDRV_CHUNK"] = str(chunk)
    lout_path = None
    if lout:
        lout_path = tempfile.NamedTemporaryFile(suffix=".lout", delete=False).name
        e["REFDRV_LOUT"] = lout_path
    if ngl == 0:
        # a reference run: the host's arithmetic for EVERY batch. With the plug-in linked in, ggml_backend_sched would ship the >= 32-token batches of a
        # -ngl 0 run to the GPU (offload_op, as the CUDA plug-in does, ggml-cuda.cu:3201-3208) - round 6 found the 8200-token "CPU reference" doing that on the GPU box
        e["GGML_MI355_OFFLOAD"] = "0"
    if env:
        e.update(env)
    cmd = [drv, "-m", gguf_path, "-c", str(n_ctx), "-t", str(threads), "-ngl", str(ngl)] + list(extra_args)
    r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"llama-ref-driver rc={r.returncode}\n{r.stderr[-4000:]}")
    stats = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"refdrv"')][-1])
    stats["stderr"] = r.stderr
    hdr = np.fromfile(out, dtype=np.int32, count=4)
    assert hdr[0] == 0x52444c4c
    ng, nv = int(hdr[2]), int(hdr[3])
    toks = np.fromfile(out, dtype=np.int32, offset=16, count=ng)
    logits = np.fromfile(out, dtype=np.float32, offset=16 + 4 * ng).reshape(ng, nv)
    os.unlink(out)
    if lout_path:
        h3 = np.fromfile(lout_path, dtype=np.int32, count=3)
        assert h3[0] == 0x54554f4c
        nr, ne = int(h3[1]), int(h3[2])
        layers = np.fromfile(lout_path, dtype=np.int32, offset=12, count=nr)
        rows = np.fromfile(lout_path, dtype=np.float32, offset=12 + 4 * nr).reshape(nr, ne) if nr else np.zeros((0, 0), np.float32)
        stats["lout"] = {int(l): rows[i] for i, l in enumerate(layers)}
        os.unlink(lout_path)
    return toks, logits, stats
