"""SURVEY §8d model fixtures (TEST INFRASTRUCTURE: uses the reference's quantizer out of oracle/_ref; nothing under
prima_cpp_amd/ imports this module).

Weights: per tensor FP32 i.i.d. N(0, 1/K_in), generator std::mt19937 seeded with 0x9E3779B9 ^ fnv1a(tensor name) (here
numpy's MT19937, one stream per 256-row chunk so that the chunks can be produced by a thread pool), quantized with the
reference's `ggml_quantize_chunk` (ggml/src/ggml.c:21826, no imatrix) to the Q4_K_M mixture of `llama_tensor_get_type`
(src/llama.cpp:19271-19490); norm weights 1 + N(0, 0.02^2); `rope_freqs` = the Llama-3.1 formula (factor 8, low / high
frequency factors 1 / 4, original context 8192). Prompts: token ids uniform in [0, V) from mt19937(1234), first id 1.

Two flavours of the same layers:
  * plain      - token_embd and output.weight are N(0, 1/K) like everything else: logits are flat (top-1 / top-2 margins of
                 the order of the int8 re-quantization noise), the whole-model comparison is statistical;
  * peaked     - the residual stream is dominated by a large token embedding (N(0, s_e^2)) and output.weight row pi(j)
                 = g * D_j / |D_j| + N(0, 1/K) with D_j the DEQUANTIZED embedding row of token j: the logit of pi(current token)
                 stands g * sqrt(E) * cos(theta) ~ 7-8 sigma above the bulk, where theta is the angle between the final
                 hidden state and the token's embedding - every layer still moves the hidden state (|branches| ~ 0.5 |x|),
                 but top-1 / top-2 margins are >> the quantization step, so greedy tokens can be compared for EQUALITY.
"""
import concurrent.futures as cf
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prima_cpp_amd import gguf as G  # noqa: E402  (GGUF container writer + the Q4_K_M type table; no compute)

CHUNK = 256


def fnv1a(s):
    h = 0x811C9DC5
    for b in s.encode():
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h


def _rng(name, chunk):
    return np.random.Generator(np.random.MT19937([0x9E3779B9 ^ fnv1a(name), chunk]))


def chunk_f32(name, chunk, rows, K, sigma):
    return _rng(name, chunk).standard_normal((rows, K), dtype=np.float32) * np.float32(sigma)


def n_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                   # cgroup quota (the GPU box exposes 256 logical CPUs with a 16-CPU quota)
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(q) // int(p)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def llama31_rope_freqs(head_dim, base=500000.0, factor=8.0, low=1.0, high=4.0, orig=8192):
    out = np.empty(head_dim // 2, dtype=np.float32)
    for i in range(head_dim // 2):
        freq = base ** (-2.0 * i / head_dim)
        wavelen = 2 * np.pi / freq
        if wavelen < orig / high:
            out[i] = 1.0
        elif wavelen > orig / low:
            out[i] = factor
        else:
            s = (orig / wavelen - low) / (high - low)
            out[i] = 1.0 / ((1 - s) / factor + s)
    return out


def prompt_tokens(n_vocab, n=16, seed=1234):
    p = np.random.Generator(np.random.MT19937(seed)).integers(0, n_vocab, n)
    p[0] = 1
    return p.astype(np.int64)


def peaked_next(tok, n_vocab):
    """pi(j): the token the peaked fixture's head points to after token j (chunks of 256 ids, rotated by one chunk, ids inside
    a chunk permuted) - n_vocab must be a multiple of 256."""
    nc = n_vocab // CHUNK
    c, r = divmod(int(tok), CHUNK)
    return ((c + 1) % nc) * CHUNK + (r * 37 + 11) % CHUNK


def default_peak_gain(E, n_vocab):
    """Target logit = g sqrt(E) cos(theta) with cos(theta) ~ 0.9, set 4.5 sigma above the expected maximum of the n_vocab bulk
    logits (sqrt(2 ln V) sigma): the smallest top-1 / top-2 margin over 128 steps stays > 1 sigma, the int8 re-quantization noise
    is ~0.05 sigma (measured between the reference's own scalar and AVX2 builds)."""
    return (np.sqrt(2 * np.log(n_vocab)) + 4.5) / (np.sqrt(E) * 0.9)


class Quantizer:
    """Streams a [N, K] matrix through the reference quantizer, 256 rows per task, on a thread pool (ctypes releases the GIL)."""

    def __init__(self, ref, threads=None):
        self.ref = ref
        self.threads = threads or n_threads()
        self.pool = cf.ThreadPoolExecutor(self.threads)

    def writer(self, t, K, N, rows_fn):
        """rows_fn(chunk index, n rows) -> f32 [n, K]. Returns callable(file) for gguf.write_gguf."""
        nch = (N + CHUNK - 1) // CHUNK

        def task(c):
            n = min(CHUNK, N - c * CHUNK)
            return self.ref.quantize_weights(t, np.ascontiguousarray(rows_fn(c, n), dtype=np.float32))

        def write(f):
            ahead = 2 * self.threads
            futs = {}
            nxt = 0
            for c in range(nch):
                while nxt < nch and nxt < c + ahead:
                    futs[nxt] = self.pool.submit(task, nxt)
                    nxt += 1
                f.write(memoryview(futs.pop(c).result()).cast("B"))
        return write

    def dequant_rows(self, t, K, blocks, n):
        rs = G.row_size(t, K)
        out = np.empty((n, K), dtype=np.float32)
        b = np.ascontiguousarray(blocks).reshape(n, rs)
        for i in range(n):
            out[i] = self.ref.dequantize_row(t, b[i], K)
        return out


def write_model(path, ref, arch=0, n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256, is_70b=False,
                peaked=False, embd_sigma=None, peak_gain=None, n_ctx_train=8192, threads=None, tag="m"):
    """Writes a SURVEY-8d GGUF. Layer tensors depend only on (tag, layer, kind): the plain and the peaked file of one shape
    hold identical layers. Returns dict(path, peaked, embd_sigma, peak_gain)."""
    Q = Quantizer(ref, threads)
    E, dh = n_embd, n_embd // n_head
    Eq, Ekv = dh * n_head, dh * n_head_kv
    rms_eps = 1e-5 if arch == 0 else 1e-6
    base = 500000.0 if arch == 0 else 1000000.0
    kv = G.model_kv(arch, n_layer, E, n_head, n_head_kv, n_ff, n_vocab, n_ctx_train, rms_eps, base, name=f"survey-8d-{tag}")
    shapes = {"wq": (E, Eq), "wk": (E, Ekv), "wv": (E, Ekv), "wo": (Eq, E), "ffn_gate": (E, n_ff), "ffn_up": (E, n_ff), "ffn_down": (n_ff, E)}

    def gauss(name, K, sigma):
        return lambda c, n: chunk_f32(name, c, n, K, sigma)

    def f32vec(name, n, mean, std):
        return (mean + _rng(name, 0).standard_normal(n) * std).astype(np.float32)

    def qtype(kind, il, K):
        t = G.q4_k_m_type(kind, il, n_layer, is_70b)
        return G.Q8_0 if (t in (G.Q4_K, G.Q5_K, G.Q6_K) and K % 256) else t       # src/llama.cpp:19547

    # token embedding: N(0, 1/K) (plain) or N(0, s_e^2) (peaked; s_e^2 ~ 4x the variance the 2L residual branches add, so that the
    # embedding keeps cos(theta) ~ 0.9 with the final hidden state whatever the layers do)
    if peaked:
        assert n_vocab % CHUNK == 0
        embd_sigma = embd_sigma or float(np.sqrt(4 * 0.8 * n_layer))
        peak_gain = peak_gain or default_peak_gain(E, n_vocab)
    else:
        embd_sigma = embd_sigma or 1.0 / np.sqrt(E)
    te_name = f"{tag}.token_embd.weight"
    te_t = G.Q4_K

    def out_rows(c, n):
        noise = chunk_f32(f"{tag}.output.weight", c, n, E, 1.0 / np.sqrt(E))
        if not peaked:
            return noise
        # rows [256 c, 256 c + n) = pi(j) for j in chunk c - 1: row 256 c + p(r) <- D_{256 (c - 1) + r}
        nc = n_vocab // CHUNK
        src = (c - 1) % nc
        emb = chunk_f32(te_name, src, CHUNK, E, embd_sigma)
        D = Q.dequant_rows(te_t, E, ref.quantize_weights(te_t, emb), CHUNK)
        D /= np.linalg.norm(D, axis=1, keepdims=True)
        out = noise.copy()
        r = np.arange(CHUNK)
        out[(r * 37 + 11) % CHUNK] += np.float32(peak_gain) * D[r]
        return out

    T = [("token_embd.weight", te_t, (E, n_vocab), Q.writer(te_t, E, n_vocab, gauss(te_name, E, embd_sigma)))]
    for il in range(n_layer):
        p = f"blk.{il}."
        T.append((p + "attn_norm.weight", G.F32, (E,), f32vec(f"{tag}.{p}attn_norm", E, 1.0, 0.02)))
        for kind in ("wq", "wk", "wv", "wo"):
            K, N = shapes[kind]
            t = qtype(kind, il, K)
            nm = p + G.LAYER_TENSORS[kind]
            T.append((nm, t, (K, N), Q.writer(t, K, N, gauss(f"{tag}.{nm}", K, 1.0 / np.sqrt(K)))))
            if arch == 1 and kind in ("wq", "wk", "wv"):
                T.append((p + G.LAYER_TENSORS["b" + kind[1]], G.F32, (N,), f32vec(f"{tag}.{p}b{kind[1]}", N, 0.0, 0.1)))
        T.append((p + "ffn_norm.weight", G.F32, (E,), f32vec(f"{tag}.{p}ffn_norm", E, 1.0, 0.02)))
        for kind in ("ffn_gate", "ffn_up", "ffn_down"):
            K, N = shapes[kind]
            t = qtype(kind, il, K)
            nm = p + G.LAYER_TENSORS[kind]
            T.append((nm, t, (K, N), Q.writer(t, K, N, gauss(f"{tag}.{nm}", K, 1.0 / np.sqrt(K)))))
    T.append(("output_norm.weight", G.F32, (E,), f32vec(f"{tag}.output_norm", E, 1.0, 0.02)))
    T.append(("output.weight", G.Q6_K, (E, n_vocab), Q.writer(G.Q6_K, E, n_vocab, out_rows)))
    if arch == 0:
        T.append(("rope_freqs.weight", G.F32, (dh // 2,), llama31_rope_freqs(dh, base)))
    G.write_gguf(path, kv, T)
    Q.pool.shutdown()
    return dict(path=path, peaked=peaked, embd_sigma=embd_sigma, peak_gain=peak_gain, n_vocab=n_vocab)


def copy_with_new_head(src_path, dst_path, ref, peaked, tag="m", threads=None, embd_sigma=None, peak_gain=None):
    """The same layers under the other token_embd / output.weight: re-quantizes only those two tensors (the layer tensors are
    copied byte for byte out of `src_path`)."""
    g = G.GGUFFile(src_path)
    a = g.kv["general.architecture"]
    E, L, V = int(g.kv[f"{a}.embedding_length"]), int(g.kv[f"{a}.block_count"]), int(g.kv[f"{a}.vocab_size"])
    Q = Quantizer(ref, threads)
    if peaked:
        embd_sigma = embd_sigma or float(np.sqrt(4 * 0.8 * L))
        peak_gain = peak_gain or default_peak_gain(E, V)
    else:
        embd_sigma = embd_sigma or 1.0 / np.sqrt(E)
    te_name, te_t = f"{tag}.token_embd.weight", G.Q4_K

    def out_rows(c, n):
        noise = chunk_f32(f"{tag}.output.weight", c, n, E, 1.0 / np.sqrt(E))
        if not peaked:
            return noise
        nc = V // CHUNK
        emb = chunk_f32(te_name, (c - 1) % nc, CHUNK, E, embd_sigma)
        D = Q.dequant_rows(te_t, E, ref.quantize_weights(te_t, emb), CHUNK)
        D /= np.linalg.norm(D, axis=1, keepdims=True)
        out = noise.copy()
        r = np.arange(CHUNK)
        out[(r * 37 + 11) % CHUNK] += np.float32(peak_gain) * D[r]
        return out

    kv = {}
    # metadata is re-created from the source's values (types as model_kv writes them)
    arch = 0 if a == "llama" else 1
    kv = G.model_kv(arch, L, E, int(g.kv[f"{a}.attention.head_count"]), int(g.kv[f"{a}.attention.head_count_kv"]),
                    int(g.kv[f"{a}.feed_forward_length"]), V, int(g.kv[f"{a}.context_length"]),
                    float(g.kv[f"{a}.attention.layer_norm_rms_epsilon"]), float(g.kv[f"{a}.rope.freq_base"]), name=g.kv["general.name"])
    T = []
    for name, (t, shape, data) in g.tensors.items():
        if name == "token_embd.weight":
            T.append((name, te_t, shape, Q.writer(te_t, E, V, lambda c, n: chunk_f32(te_name, c, n, E, embd_sigma))))
        elif name == "output.weight":
            T.append((name, G.Q6_K, shape, Q.writer(G.Q6_K, E, V, out_rows)))
        else:
            T.append((name, t, shape, data))
    G.write_gguf(dst_path, kv, T)
    Q.pool.shutdown()
    g.close()
    return dict(path=dst_path, peaked=peaked, embd_sigma=embd_sigma, peak_gain=peak_gain, n_vocab=V)


def engine_greedy(path, shape, prompt, n_gen, n_ctx=256):
    """Greedy decode of a fixture file on the resident engine (pm355_model_*): the prompt as one batch, then token by token through the captured graph.
    shape = the write_model keyword dict. Returns (tokens [n_gen], logits [n_gen, n_vocab])."""
    import torch
    import prima_cpp_amd.engine as eng
    s = shape
    hp = dict(arch=0, n_layer=s["n_layer"], n_embd=s["n_embd"], n_head=s["n_head"], n_head_kv=s["n_head_kv"], head_dim=s["n_embd"] // s["n_head"],
              n_ff=s["n_ff"], n_vocab=s["n_vocab"], rms_eps=1e-5, rope_freq_base=500000.0)
    w = eng.Window(hp, n_ctx=n_ctx)
    w.load_gguf(path)
    w.finalize(max_tokens=max(len(prompt), 1))
    toks, logits = [], []
    _, lg, _ = w.decode(tokens=torch.from_numpy(np.asarray(prompt, dtype=np.int32)).cuda(), pos0=0, want_hidden=False)
    pos = len(prompt)
    w.set_pos(pos)
    tok_d = torch.empty(1, dtype=torch.int32, device="cuda")
    lg_d = torch.empty(s["n_vocab"], dtype=torch.float32, device="cuda")
    for i in range(n_gen):
        l = lg.cpu().numpy()
        t = int(np.argmax(l))
        toks.append(t)
        logits.append(l)
        if i == n_gen - 1:
            break
        tok_d.fill_(t)
        w.step(token=tok_d, logits=lg_d)          # single-token step: the replayed hipGraph, position advanced on the device
        lg = lg_d
    w.close()
    return np.array(toks), np.stack(logits)
