"""GGUF v3 container: minimal writer + reader (numpy only), written from the format as the reference reads it
(gguf_init_from_file, ggml/src/ggml.c:22327-22640: magic "GGUF", u32 version, u64 n_tensors, u64 n_kv, the KV section,
the tensor-info section, padding to `general.alignment` (default 32), then every tensor at its aligned offset).

Used for (a) synthetic Llama / Qwen2-shaped model files (`tokenizer.ggml.model = "no_vocab"`, src/llama.cpp:6610) that the
reference's unmodified loader + `llama_decode` and our engine both consume, and (b) loading a GGUF straight into the
resident engine (`engine.Model.from_gguf`). Tensor names / metadata keys are the reference's
(LLM_TENSOR_NAMES src/llama.cpp:560-600, LLM_KV_NAMES :424-520).
"""
import mmap
import struct

import numpy as np

GGUF_MAGIC = 0x46554747
GGUF_VERSION = 3
ALIGNMENT = 32

# enum gguf_type (ggml/include/ggml.h:2353-2368)
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)
_SCALAR = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f", T_BOOL: "<?",
           T_U64: "<Q", T_I64: "<q", T_F64: "<d"}

# ggml types on the path: (block elements, block bytes)   (ggml/src/ggml-common.h)
F32, F16, Q8_0, Q4_K, Q5_K, Q6_K = 0, 1, 8, 12, 13, 14
BLOCK = {F32: (1, 4), F16: (1, 2), Q8_0: (32, 34), Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210)}


def row_size(t, k):
    n, b = BLOCK[t]
    assert k % n == 0, (t, k)
    return k // n * b


def tensor_nbytes(t, shape):
    n = row_size(t, shape[0])
    for d in shape[1:]:
        n *= d
    return n


def _pad(n, a=ALIGNMENT):
    return (n + a - 1) // a * a


def _str(s):
    b = s.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def _kv(key, val):
    """val: (gguf_type, python value) or (T_ARR, (elem_type, list))."""
    t, v = val
    out = _str(key) + struct.pack("<I", t)
    if t == T_STR:
        return out + _str(v)
    if t == T_ARR:
        et, items = v
        out += struct.pack("<IQ", et, len(items))
        if et == T_STR:
            return out + b"".join(_str(x) for x in items)
        return out + b"".join(struct.pack(_SCALAR[et], x) for x in items)
    return out + struct.pack(_SCALAR[t], v)


def write_gguf(path, kv, tensors):
    """kv: dict key -> (gguf_type, value); tensors: list of (name, ggml_type, shape (ne0, ne1, ...), data) where data is a
    uint8/any numpy array holding exactly tensor_nbytes bytes, or a callable(file) that writes exactly that many bytes."""
    infos, off = [], 0
    for name, t, shape, data in tensors:
        nb = tensor_nbytes(t, shape)
        infos.append((name, t, shape, off, nb))
        off = _pad(off + nb)
    with open(path, "wb") as f:
        f.write(struct.pack("<IIQQ", GGUF_MAGIC, GGUF_VERSION, len(tensors), len(kv)))
        for k, v in kv.items():
            f.write(_kv(k, v))
        for name, t, shape, o, _ in infos:
            f.write(_str(name) + struct.pack("<I", len(shape)) + b"".join(struct.pack("<Q", d) for d in shape) +
                    struct.pack("<IQ", t, o))
        f.write(b"\0" * (_pad(f.tell()) - f.tell()))
        base = f.tell()
        for (name, t, shape, data), (_, _, _, o, nb) in zip(tensors, infos):
            assert f.tell() == base + o, name
            if callable(data):
                data(f)
            else:
                a = np.ascontiguousarray(data)
                assert a.nbytes == nb, (name, a.nbytes, nb)
                f.write(memoryview(a).cast("B"))
            assert f.tell() == base + o + nb, name
            f.write(b"\0" * (_pad(f.tell() - base) - (f.tell() - base)))
    return path


class GGUFFile:
    """Read-only view: .kv (dict key -> python value), .tensors (dict name -> (type, shape, uint8 memmap view))."""

    def __init__(self, path):
        self.path = path
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        mm, self._p = self._mm, 0
        magic, ver, n_t, n_kv = self._unpack("<IIQQ")
        if magic != GGUF_MAGIC or ver not in (2, 3):
            raise ValueError(f"{path}: not a GGUF v2/v3 file")
        self.kv = {}
        for _ in range(n_kv):
            key = self._rstr()
            (t,) = self._unpack("<I")
            self.kv[key] = self._rval(t)
        infos = []
        for _ in range(n_t):
            name = self._rstr()
            (nd,) = self._unpack("<I")
            shape = self._unpack("<" + "Q" * nd)
            t, off = self._unpack("<IQ")
            infos.append((name, t, tuple(shape), off))
        align = int(self.kv.get("general.alignment", ALIGNMENT))
        base = _pad(self._p, align)
        buf = np.frombuffer(mm, dtype=np.uint8)
        self.tensors = {}
        for name, t, shape, off in infos:
            nb = tensor_nbytes(t, shape) if t in BLOCK else None
            self.tensors[name] = (t, shape, buf[base + off: base + off + nb] if nb is not None else None)

    def _unpack(self, fmt):
        n = struct.calcsize(fmt)
        v = struct.unpack_from(fmt, self._mm, self._p)
        self._p += n
        return v

    def _rstr(self):
        (n,) = self._unpack("<Q")
        s = self._mm[self._p:self._p + n].decode("utf-8", "replace")
        self._p += n
        return s

    def _rval(self, t):
        if t == T_STR:
            return self._rstr()
        if t == T_ARR:
            et, n = self._unpack("<IQ")
            return [self._rval(et) for _ in range(n)]
        return self._unpack(_SCALAR[t])[0]

    def close(self):
        self.tensors = {}
        try:
            self._mm.close()
        except BufferError:
            pass                      # numpy views still alive; the mapping goes away with them
        self._f.close()


# ------------------------------------------------------------------------------------------------------------------
# model-shaped files
# ------------------------------------------------------------------------------------------------------------------
ARCH_NAMES = {0: "llama", 1: "qwen2"}


def model_kv(arch, n_layer, n_embd, n_head, n_head_kv, n_ff, n_vocab, n_ctx_train, rms_eps, rope_freq_base, name="synthetic"):
    """Metadata of llm_load_hparams (src/llama.cpp:5823-5960) for a dense llama / qwen2 model without a vocabulary."""
    a = ARCH_NAMES[arch]
    return {
        "general.architecture": (T_STR, a),
        "general.name": (T_STR, name),
        f"{a}.context_length": (T_U32, n_ctx_train),
        f"{a}.embedding_length": (T_U32, n_embd),
        f"{a}.block_count": (T_U32, n_layer),
        f"{a}.feed_forward_length": (T_U32, n_ff),
        f"{a}.attention.head_count": (T_U32, n_head),
        f"{a}.attention.head_count_kv": (T_U32, n_head_kv),
        f"{a}.attention.layer_norm_rms_epsilon": (T_F32, rms_eps),
        f"{a}.rope.freq_base": (T_F32, rope_freq_base),
        f"{a}.rope.dimension_count": (T_U32, n_embd // n_head),
        f"{a}.vocab_size": (T_U32, n_vocab),
        "tokenizer.ggml.model": (T_STR, "no_vocab"),
    }


# engine tensor kind (include/prima_mi355.h enum pm355_tensor_kind) -> GGUF base name (LLM_TENSOR_NAMES src/llama.cpp:560-600)
LAYER_TENSORS = {"attn_norm": "attn_norm.weight", "wq": "attn_q.weight", "wk": "attn_k.weight", "wv": "attn_v.weight",
                 "wo": "attn_output.weight", "ffn_norm": "ffn_norm.weight", "ffn_gate": "ffn_gate.weight",
                 "ffn_up": "ffn_up.weight", "ffn_down": "ffn_down.weight", "bq": "attn_q.bias", "bk": "attn_k.bias",
                 "bv": "attn_v.bias"}
MODEL_TENSORS = {"tok_embd": "token_embd.weight", "out_norm": "output_norm.weight", "output": "output.weight",
                 "rope_freqs": "rope_freqs.weight"}


def q4_k_m_type(kind, il, n_layer, is_70b=False):
    """The Q4_K_M tensor-type mixture of llama_tensor_get_type (src/llama.cpp:19278-19280, :19382-19389, :19438-19445,
    :19300-19316): "more bits" layers get Q6_K attn_v / ffn_down; MODEL_70B keeps Q5_K attn_v on the rest; output Q6_K."""
    more = il < n_layer // 8 or il >= 7 * n_layer // 8 or (il - n_layer // 8) % 3 == 2
    if kind == "wv":
        return Q6_K if more else (Q5_K if is_70b else Q4_K)
    if kind == "ffn_down":
        return Q6_K if more else Q4_K
    if kind == "output":
        return Q6_K
    if kind in ("attn_norm", "ffn_norm", "out_norm", "bq", "bk", "bv", "rope_freqs"):
        return F32
    return Q4_K


def _cyclic(pool, off, n):
    """n bytes of `pool` read cyclically from offset off (slice copies: memcpy speed; np.tile on a 1-D array is ~50x slower)."""
    out = np.empty(n, dtype=np.uint8)
    done = 0
    while done < n:
        take = min(n - done, pool.size - off)
        out[done:done + take] = pool[off:off + take]
        done += take
        off = 0
    return out


_POOL = None
_POOL_BYTES = (32 << 20) + 8 * 1009          # not a multiple of any block or row size


def random_valid_blocks(t, nrows, k, rng, scale=None):
    """Random VALID quant blocks (every bit pattern of the packed fields, finite fp16 scales) with |w| ~ scale
    (default 1/sqrt(k)); mean ~ 0 for Q4_K/Q5_K because dmin/d is chosen so that E[d*sc*q] == E[dmin*m]."""
    nper, bs = BLOCK[t]
    nb = nrows * (k // nper)
    n = nb * bs
    if n <= _POOL_BYTES:
        raw = rng.integers(0, 2 ** 64 - 1, size=(n + 7) // 8, dtype=np.uint64, endpoint=True).view(np.uint8)[:n].reshape(nb, bs).copy()
    else:
        # multi-GB shapes: drawing every byte from the generator would take minutes of host time. Bytes come from one random pool,
        # read cyclically from a random (odd) offset: no two rows hold the same bytes at the same alignment
        global _POOL
        if _POOL is None:
            _POOL = np.random.default_rng(0x9E3779B9).integers(0, 2 ** 64 - 1, size=_POOL_BYTES // 8, dtype=np.uint64, endpoint=True).view(np.uint8)
        off = int(rng.integers(0, _POOL_BYTES // 2)) | 1
        raw = _cyclic(_POOL, off, n).reshape(nb, bs)
    if scale is None:
        scale = 1.0 / np.sqrt(k)

    def f16(vals):
        return np.asarray(vals, dtype=np.float16).view(np.uint8).reshape(nb, 2)
    u = rng.uniform(0.5, 1.5, nb)
    if t == Q8_0:
        raw[:, 0:2] = f16(u * scale / 64.0)
    elif t in (Q4_K, Q5_K):
        qmax = 15 if t == Q4_K else 31
        d = u * scale / (qmax * 16.0)
        raw[:, 0:2] = f16(d)
        raw[:, 2:4] = f16(d * (qmax / 2.0))
    elif t == Q6_K:
        raw[:, 208:210] = f16(u * scale / (32.0 * 64.0))
    else:
        raise ValueError(t)
    return raw.reshape(-1)


def write_synthetic_model(path, arch, n_layer, n_embd, n_head, n_head_kv, n_ff, n_vocab, n_ctx_train=8192, seed=1234,
                          rope_freqs=None, is_70b=False, all_type=None, distinct_layers=2, branch_scale=1.0):
    """A Llama / Qwen2-shaped GGUF with random valid quant blocks in the Q4_K_M mixture (or every matrix `all_type`).
    To bound generation time for multi-GB shapes only `distinct_layers` different random tensors are drawn per
    (tensor kind, type); further layers reuse them rolled by a layer-dependent number of rows.
    branch_scale multiplies the weights of the two projections that write into the residual stream (attn_output, ffn_down).
    (It does NOT tame the logit-level spread between two float summation orders: that spread comes from the int8 re-quantization
    of the activations in front of every mat-vec, which turns a perturbation eps into flips of size `step` with probability
    eps/step, i.e. amplitude ~sqrt(eps*step) per stage, and saturates at ~step after a handful of stages - DESIGN.md, parity.)"""
    rng = np.random.default_rng(seed)
    a = arch
    E, dh = n_embd, n_embd // n_head
    Eq, Ekv = dh * n_head, dh * n_head_kv
    rms_eps = 1e-5 if a == 0 else 1e-6
    base = 500000.0 if a == 0 else 1000000.0
    kv = model_kv(a, n_layer, E, n_head, n_head_kv, n_ff, n_vocab, n_ctx_train, rms_eps, base)
    shapes = {"wq": (E, Eq), "wk": (E, Ekv), "wv": (E, Ekv), "wo": (Eq, E), "ffn_gate": (E, n_ff), "ffn_up": (E, n_ff),
              "ffn_down": (n_ff, E)}
    pool = {}

    def mat(kind, t, K, N, il, n_out=None):
        """-> (type, writer(file)): rows of the pooled tensor rolled by a layer-dependent amount (tiled up to n_out rows),
        written straight to the file so multi-GB models never sit in memory."""
        if t in (Q4_K, Q5_K, Q6_K) and K % 256:
            t = Q8_0                                     # llama_tensor_get_type fallback for K % 256 != 0 (src/llama.cpp:19547)
        key = (kind, t, il % distinct_layers)
        if key not in pool:
            pool[key] = random_valid_blocks(t, N, K, rng, scale=(branch_scale if kind in ("wo", "ffn_down") else 1.0) / np.sqrt(K))
        d = pool[key].reshape(N, row_size(t, K))
        r = (il // distinct_layers * 37) % N
        n_out = n_out or N

        def write(f):
            left, first = n_out, True
            while left > 0:
                if first and r:
                    f.write(memoryview(d[r:r + left]).cast("B")); left -= min(left, N - r)
                    if left > 0:
                        f.write(memoryview(d[:min(r, left)]).cast("B")); left -= min(r, left)
                else:
                    f.write(memoryview(d[:left]).cast("B")); left -= min(left, N)
                first = False
        return t, write

    tensors = []
    t, d = mat("tok_embd", all_type or Q4_K, E, min(n_vocab, 8192), 0, n_vocab)
    tensors.append(("token_embd.weight", t, (E, n_vocab), d))
    for il in range(n_layer):
        p = f"blk.{il}."
        tensors.append((p + "attn_norm.weight", F32, (E,), (1 + rng.normal(0, 0.02, E)).astype(np.float32)))
        for kind in ("wq", "wk", "wv", "wo"):
            K, N = shapes[kind]
            t, d = mat(kind, all_type or q4_k_m_type(kind, il, n_layer, is_70b), K, N, il)
            tensors.append((p + LAYER_TENSORS[kind], t, (K, N), d))
            if a == 1 and kind in ("wq", "wk", "wv"):
                tensors.append((p + LAYER_TENSORS["b" + kind[1]], F32, (N,), rng.normal(0, 0.1, N).astype(np.float32)))
        tensors.append((p + "ffn_norm.weight", F32, (E,), (1 + rng.normal(0, 0.02, E)).astype(np.float32)))
        for kind in ("ffn_gate", "ffn_up", "ffn_down"):
            K, N = shapes[kind]
            t, d = mat(kind, all_type or q4_k_m_type(kind, il, n_layer, is_70b), K, N, il)
            tensors.append((p + LAYER_TENSORS[kind], t, (K, N), d))
    tensors.append(("output_norm.weight", F32, (E,), (1 + rng.normal(0, 0.02, E)).astype(np.float32)))
    t, d = mat("output", all_type or Q6_K, E, n_vocab, 1)          # every row distinct: no exactly tied logits
    tensors.append(("output.weight", t, (E, n_vocab), d))
    if rope_freqs is None:
        rope_freqs = a == 0
    if rope_freqs:
        tensors.append(("rope_freqs.weight", F32, (dh // 2,), (1.0 + rng.uniform(0, 7, dh // 2)).astype(np.float32)))
    return write_gguf(path, kv, tensors)
