"""Host binding of the engine layer (part (B) of include/prima_mi355.h): a resident decoder for one
piped-ring layer window. Mirrors the reference's vocabulary: a *window* of layers owned by a rank
(this_layer_is_mine, src/llama.cpp:3838), the activation hand-off tensor `sub_gf_out`, KV cache."""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .lib import F32, Q4_K, Q5_K, Q6_K, Q8_0, check, ptr, stream_ptr  # noqa: F401

(T_ATTN_NORM, T_WQ, T_WK, T_WV, T_WO, T_FFN_NORM, T_FFN_GATE, T_FFN_UP, T_FFN_DOWN, T_BQ, T_BK, T_BV,
 T_TOK_EMBD, T_OUT_NORM, T_OUTPUT, T_ROPE_FREQS) = range(16)
HAS_EMBD, HAS_HEAD = 1, 2

_vp, _i32, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t


class HParams(C.Structure):
    _fields_ = [("arch", C.c_int32), ("n_layer", C.c_int32), ("n_embd", C.c_int32), ("n_head", C.c_int32),
                ("n_head_kv", C.c_int32), ("head_dim", C.c_int32), ("n_ff", C.c_int32), ("n_vocab", C.c_int32),
                ("n_ctx", C.c_int32), ("n_ctx_orig", C.c_int32),
                ("rms_eps", C.c_float), ("rope_freq_base", C.c_float), ("rope_freq_scale", C.c_float),
                ("pad_", C.c_int32)]


LLAMA3_8B = dict(arch=0, n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, head_dim=128, n_ff=14336, n_vocab=128256,
                 rms_eps=1e-5, rope_freq_base=500000.0)
LLAMA3_70B = dict(arch=0, n_layer=80, n_embd=8192, n_head=64, n_head_kv=8, head_dim=128, n_ff=28672, n_vocab=128256,
                  rms_eps=1e-5, rope_freq_base=500000.0)
QWEN25_72B = dict(arch=1, n_layer=80, n_embd=8192, n_head=64, n_head_kv=8, head_dim=128, n_ff=29568, n_vocab=152064,
                  rms_eps=1e-6, rope_freq_base=1000000.0)


def q4_k_m_types(hp, il):
    """The Q4_K_M mixture of llama_tensor_get_type (src/llama.cpp:19271-19490) for layer il."""
    n = hp["n_layer"]
    more = il < n // 8 or il >= 7 * n // 8 or (il - n // 8) % 3 == 2       # use_more_bits (:19260)
    is70 = n == 80 and hp["n_head"] // hp["n_head_kv"] == 8                 # MODEL_70B heuristic (:19382)
    v = Q6_K if more else (Q5_K if is70 else Q4_K)
    return {T_WQ: Q4_K, T_WK: Q4_K, T_WV: v, T_WO: Q4_K, T_FFN_GATE: Q4_K, T_FFN_UP: Q4_K,
            T_FFN_DOWN: Q6_K if more else Q4_K}


def q6_k_types(hp, il):
    """'Q6_K' file type: every 2-D weight Q6_K, except rows not divisible by 256 -> Q8_0 fallback (:19447-19470)."""
    def t(k):
        return Q6_K if k % 256 == 0 else Q8_0
    E, F = hp["n_embd"], hp["n_ff"]
    return {T_WQ: t(E), T_WK: t(E), T_WV: t(E), T_WO: t(E), T_FFN_GATE: t(E), T_FFN_UP: t(E), T_FFN_DOWN: t(F)}


_sigs_done = False


def _lib():
    global _sigs_done
    lib = L.load()
    if not _sigs_done:
        S = {
            "pm355_model_new": (_vp, [C.POINTER(HParams), _i32, _i32, _i32]),
            "pm355_model_free": (None, [_vp]),
            "pm355_model_error": (C.c_char_p, [_vp]),
            "pm355_model_set_tensor": (_i32, [_vp, _i32, _i32, _i32, _vp, _sz]),
            "pm355_model_fill_tensor": (_i32, [_vp, _i32, _i32, _i32, C.c_uint64, C.c_float]),
            "pm355_model_finalize": (_i32, [_vp, _i32]),
            "pm355_model_finalize_seqs": (_i32, [_vp, _i32, _i32]),
            "pm355_model_set_seq_pos": (_i32, [_vp, _i32, _i32, _vp]),
            "pm355_model_set_seq": (_i32, [_vp, _i32, _vp]),
            "pm355_model_head": (_i32, [_vp, _vp, _vp, _vp, _vp]),
            "pm355_model_step_ex": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
            "pm355_model_weight_bytes": (_sz, [_vp]),
            "pm355_model_kv_bytes_per_pos": (_sz, [_vp]),
            "pm355_model_kv_clear": (_i32, [_vp, _vp]),
            "pm355_model_kv_ptr": (_vp, [_vp, _i32, _i32]),
            "pm355_model_decode": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
            "pm355_model_generate": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
            "pm355_model_check": (_i32, [_vp]),
            "pm355_model_set_pos": (_i32, [_vp, _i32, _vp]),
            "pm355_model_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
            "pm355_model_set_streaming": (_i32, [_vp, _i32]),
            "pm355_model_streamed_bytes": (C.c_uint64, [_vp]),
        }
        for name, (res, args) in S.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _sigs_done = True
    return lib


class Window:
    """Layers [lo, hi) of a model resident on the current device (+ tok_embd / head when flagged)."""

    def __init__(self, hp, lo=0, hi=None, flags=HAS_EMBD | HAS_HEAD, n_ctx=512):
        self.lib = _lib()
        hp = dict(hp)
        hp.setdefault("n_ctx_orig", 8192)
        hp.setdefault("rope_freq_scale", 1.0)
        hp["n_ctx"] = n_ctx
        self.hp_dict = hp
        self.hp = HParams(**{k: v for k, v in hp.items() if k in dict(HParams._fields_)})
        self.lo, self.hi = lo, hp["n_layer"] if hi is None else hi
        self.flags = flags
        self.h = self.lib.pm355_model_new(C.byref(self.hp), self.lo, self.hi, flags)
        if not self.h:
            raise L.PM355Error("pm355_model_new failed (bad hparams / window)")
        self.dev = torch.device("cuda", torch.cuda.current_device())

    def _chk(self, rc, what):
        if rc != 0:
            raise L.PM355Error(f"{what}: rc={rc}: {self.lib.pm355_model_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.pm355_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tensors -----------------------------------------------------------------------------
    def set_tensor(self, kind, layer, qtype, data):
        data = np.ascontiguousarray(data)
        self._chk(self.lib.pm355_model_set_tensor(self.h, kind, layer, qtype, data.ctypes.data, data.nbytes),
                  f"set_tensor(kind={kind}, layer={layer})")

    def fill_tensor(self, kind, layer, qtype, seed, scale):
        self._chk(self.lib.pm355_model_fill_tensor(self.h, kind, layer, qtype, seed, scale), "fill_tensor")

    def load_desc(self, desc):
        """Load from a tests/_bind.ModelDesc (host pointers, GGUF block order)."""
        from . import lib as LL

        def nbytes(t, k, n):
            return LL.row_size(t, k) * n
        hp = self.hp
        E, Eq, Ekv, F = hp.n_embd, hp.head_dim * hp.n_head, hp.head_dim * hp.n_head_kv, hp.n_ff
        shapes = {T_ATTN_NORM: (E, 1), T_WQ: (E, Eq), T_WK: (E, Ekv), T_WV: (E, Ekv), T_WO: (Eq, E), T_FFN_NORM: (E, 1),
                  T_FFN_GATE: (E, F), T_FFN_UP: (E, F), T_FFN_DOWN: (F, E), T_BQ: (Eq, 1), T_BK: (Ekv, 1), T_BV: (Ekv, 1)}
        names = {T_ATTN_NORM: "attn_norm", T_WQ: "wq", T_WK: "wk", T_WV: "wv", T_WO: "wo", T_FFN_NORM: "ffn_norm",
                 T_FFN_GATE: "ffn_gate", T_FFN_UP: "ffn_up", T_FFN_DOWN: "ffn_down", T_BQ: "bq", T_BK: "bk", T_BV: "bv"}
        for il in range(self.lo, self.hi):
            for kind, (k, n) in shapes.items():
                arr = getattr(desc, names[kind])
                if not arr:
                    continue
                t = arr[il]
                if not t.data:
                    continue
                nb = nbytes(t.type, k, n)
                self._chk(self.lib.pm355_model_set_tensor(self.h, kind, il, t.type, t.data, nb), f"set {names[kind]}[{il}]")
        if self.flags & HAS_EMBD:
            t = desc.tok_embd
            self._chk(self.lib.pm355_model_set_tensor(self.h, T_TOK_EMBD, -1, t.type, t.data, nbytes(t.type, E, hp.n_vocab)), "tok_embd")
        if self.flags & HAS_HEAD:
            t = desc.out_norm
            self._chk(self.lib.pm355_model_set_tensor(self.h, T_OUT_NORM, -1, t.type, t.data, E * 4), "out_norm")
            t = desc.output
            self._chk(self.lib.pm355_model_set_tensor(self.h, T_OUTPUT, -1, t.type, t.data, nbytes(t.type, E, hp.n_vocab)), "output")
        if desc.rope_freqs:
            self._chk(self.lib.pm355_model_set_tensor(self.h, T_ROPE_FREQS, -1, F32, desc.rope_freqs, hp.head_dim // 2 * 4), "rope_freqs")

    def load_gguf(self, path):
        """Load this window's tensors from a GGUF file (names of LLM_TENSOR_NAMES, src/llama.cpp:560-600), straight from the
        mmap'd file through the staged uploader."""
        from . import gguf as G
        g = G.GGUFFile(path)
        layer_kinds = {"attn_norm": T_ATTN_NORM, "wq": T_WQ, "wk": T_WK, "wv": T_WV, "wo": T_WO, "ffn_norm": T_FFN_NORM,
                       "ffn_gate": T_FFN_GATE, "ffn_up": T_FFN_UP, "ffn_down": T_FFN_DOWN, "bq": T_BQ, "bk": T_BK, "bv": T_BV}

        def put(kind, layer, name, required=True):
            if name not in g.tensors:
                if required:
                    raise L.PM355Error(f"{path}: tensor {name} missing")
                return
            t, _, data = g.tensors[name]
            self._chk(self.lib.pm355_model_set_tensor(self.h, kind, layer, t, data.ctypes.data, data.nbytes), f"set_tensor {name}")
        for il in range(self.lo, self.hi):
            for k, kind in layer_kinds.items():
                put(kind, il, f"blk.{il}.{G.LAYER_TENSORS[k]}", required=not k.startswith("b"))
        if self.flags & HAS_EMBD:
            put(T_TOK_EMBD, -1, "token_embd.weight")
        if self.flags & HAS_HEAD:
            put(T_OUT_NORM, -1, "output_norm.weight")
            put(T_OUTPUT, -1, "output.weight" if "output.weight" in g.tensors else "token_embd.weight")   # tied head (src/llama.cpp:7391)
        put(T_ROPE_FREQS, -1, "rope_freqs.weight", required=False)
        torch.cuda.synchronize()
        g.close()

    def fill_synthetic(self, mixture, seed=1234, rope_freqs=True):
        """Random-init weights of the named architecture generated directly in HBM (bench)."""
        hp = self.hp_dict
        for il in range(self.lo, self.hi):
            types = mixture(hp, il)
            s = seed * 1000003 + il * 131
            self.fill_tensor(T_ATTN_NORM, il, F32, s + 1, 0.0)
            self.fill_tensor(T_FFN_NORM, il, F32, s + 2, 0.0)
            for j, kind in enumerate((T_WQ, T_WK, T_WV, T_WO, T_FFN_GATE, T_FFN_UP, T_FFN_DOWN)):
                k = hp["n_ff"] if kind == T_FFN_DOWN else hp["n_embd"]
                self.fill_tensor(kind, il, types[kind], s + 10 + j, 1.0 / np.sqrt(k) * 2.0)
            if hp["arch"] == 1:
                for j, kind in enumerate((T_BQ, T_BK, T_BV)):
                    self.fill_tensor(kind, il, F32, s + 30 + j, 0.1)
        out_t = Q6_K if hp["n_embd"] % 256 == 0 else Q8_0
        if self.flags & HAS_EMBD:
            emb_t = mixture(hp, 0).get(T_TOK_EMBD, Q4_K if mixture is q4_k_m_types else out_t)
            self.fill_tensor(T_TOK_EMBD, -1, emb_t, seed + 7, 1.0)
        if self.flags & HAS_HEAD:
            self.fill_tensor(T_OUT_NORM, -1, F32, seed + 8, 0.0)
            self.fill_tensor(T_OUTPUT, -1, out_t, seed + 9, 1.0 / np.sqrt(hp["n_embd"]) * 2.0)
        if rope_freqs and hp["arch"] == 0:
            self.fill_tensor(T_ROPE_FREQS, -1, F32, seed + 11, 0.0)

    def set_streaming(self, n_slots):
        """Layer tensors in pinned host memory, streamed through n_slots device-side layer slots (before any layer tensor is set)."""
        self._chk(self.lib.pm355_model_set_streaming(self.h, int(n_slots)), "set_streaming")

    def streamed_bytes(self):
        return int(self.lib.pm355_model_streamed_bytes(self.h))

    def finalize(self, max_tokens=1, n_seq=1):
        self.n_seq = n_seq
        self._chk(self.lib.pm355_model_finalize_seqs(self.h, max_tokens, n_seq), "finalize")

    # ---- info ----------------------------------------------------------------------------------
    @property
    def weight_bytes(self):
        return self.lib.pm355_model_weight_bytes(self.h)

    @property
    def kv_bytes_per_pos(self):
        return self.lib.pm355_model_kv_bytes_per_pos(self.h)

    def kv_clear(self):
        self._chk(self.lib.pm355_model_kv_clear(self.h, stream_ptr()), "kv_clear")

    def kv(self, layer, which):
        p = self.lib.pm355_model_kv_ptr(self.h, layer, which)
        n = self.hp.head_dim * self.hp.n_head_kv * self.hp.n_ctx
        t = torch.empty(n, dtype=torch.int16, device=self.dev)
        check(self.lib.pm355_memcpy_d2d(ptr(t), p, n * 2, stream_ptr()), "kv copy")
        return t.cpu().numpy().view(np.uint16)

    # ---- compute -------------------------------------------------------------------------------
    def decode(self, tokens=None, x_in=None, pos0=0, want_hidden=True, want_logits=True, want_argmax=False):
        """One pass over n tokens (llama_decode semantics for this window). tokens: int32 tensor on device,
        or x_in: f32 [n, n_embd] on device. Returns (hidden [n, E] or None, logits [V] or None, argmax or None)."""
        if tokens is not None:
            n = tokens.numel()
        else:
            n = x_in.shape[0]
        hidden = torch.empty((n, self.hp.n_embd), dtype=torch.float32, device=self.dev) if want_hidden else None
        head = bool(self.flags & HAS_HEAD)
        logits = torch.empty(self.hp.n_vocab, dtype=torch.float32, device=self.dev) if (want_logits and head) else None
        am = torch.empty(1, dtype=torch.int32, device=self.dev) if (want_argmax and head) else None
        self._chk(self.lib.pm355_model_decode(self.h, ptr(tokens), ptr(x_in), n, pos0, ptr(hidden), ptr(logits), ptr(am),
                                              stream_ptr()), "decode")
        return hidden, logits, am

    def generate(self, tokens_io, pos0, n_steps, use_graph=True):
        self._chk(self.lib.pm355_model_generate(self.h, ptr(tokens_io), pos0, n_steps, int(use_graph), stream_ptr()), "generate")

    def check(self):
        """0, or non-zero once the barrier watchdog of the opt-in attention + wo kernel fired (synchronizes the device)."""
        return int(self.lib.pm355_model_check(self.h))

    def set_pos(self, pos):
        self._chk(self.lib.pm355_model_set_pos(self.h, pos, stream_ptr()), "set_pos")

    def set_seq_pos(self, seq, pos):
        self._chk(self.lib.pm355_model_set_seq_pos(self.h, seq, pos, stream_ptr()), "set_seq_pos")

    def set_seq(self, seq):
        self._chk(self.lib.pm355_model_set_seq(self.h, seq, stream_ptr()), "set_seq")

    def head(self, x_row, logits=None, argmax=None):
        self._chk(self.lib.pm355_model_head(self.h, ptr(x_row), ptr(logits), ptr(argmax), stream_ptr()), "head")

    def step_ex(self, token=None, x_in=None, x_out=None, logits=None, argmax=None, advance=1, rotate=0, head_first=False,
                use_graph=True):
        self._chk(self.lib.pm355_model_step_ex(self.h, ptr(token), ptr(x_in), ptr(x_out), ptr(logits), ptr(argmax), advance,
                                               rotate, int(head_first), int(use_graph), stream_ptr()), "step_ex")

    def step(self, token=None, x_in=None, x_out=None, logits=None, argmax=None, advance=1, use_graph=True):
        self._chk(self.lib.pm355_model_step(self.h, ptr(token), ptr(x_in), ptr(x_out), ptr(logits), ptr(argmax), advance,
                                            int(use_graph), stream_ptr()), "step")


def smoke(oracle):
    """Tiny Llama-shaped window: 4-token prefill + 2 decode steps vs the CPU oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from _bind import tiny_model
    rng = np.random.default_rng(5)
    d = tiny_model(rng, arch=0, n_layer=2, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64, rope_freqs=True)
    hp = dict(arch=0, n_layer=2, n_embd=256, n_head=4, n_head_kv=2, head_dim=64, n_ff=512, n_vocab=320,
              rms_eps=d.rms_eps, rope_freq_base=d.rope_freq_base)
    w = Window(hp, n_ctx=64)
    w.load_desc(d)
    w.finalize(max_tokens=8)
    ho = oracle.model_new(d)
    toks = rng.integers(0, 320, 6).astype(np.int32)
    for tk, p0 in ((toks[:4], 0), (toks[4:5], 4), (toks[5:6], 5)):
        hid, lg, _ = w.decode(tokens=torch.from_numpy(tk).cuda(), pos0=p0)
        h_ref, l_ref = oracle.model_eval(ho, d, tokens=tk, pos0=p0)
        a, b = lg.cpu().numpy().astype(np.float64), l_ref.astype(np.float64)
        nmse = ((a - b) ** 2).sum() / (b ** 2).sum()
        assert nmse < 1e-3, nmse
    oracle.model_free(ho)
    w.close()
