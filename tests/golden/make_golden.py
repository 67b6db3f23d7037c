"""Generates tests/golden/*.npz from the UNMODIFIED reference compiled under oracle/_ref (needs /root/reference to
have been built by oracle/Makefile). The reference has no stored golden vectors for this path, so these
are outputs of the reference itself, run here, on deterministic inputs - including the reference's own synthetic data
generator (tests/test-quantize-fns.cpp:30-34: 0.1 + 2*cos(i + offset)).

    python tests/golden/make_golden.py

scalar flavour = -march=x86-64 build (canonical `#else` branches of ggml-quants.c); avx2 flavour = what a real x86 host runs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _bind import desc_to_arrays, F16, Q4_K, Q5_K, Q6_K, Q8_0, QUANT_TYPES, TYPE_NAMES, Ref, rand_blocks, row_size, tiny_model, vec_dot_type  # noqa: E402


def ref_data(n, offset):
    return (0.1 + 2 * np.cos(np.arange(n, dtype=np.float32) + np.float32(offset))).astype(np.float32)


def main():
    ref, avx2 = Ref("scalar"), Ref("avx2")
    rng = np.random.default_rng(20250725)
    out = {}
    K = 1024
    # --- activations -------------------------------------------------------------------------------
    acts = {"cos0": ref_data(K, 0.0), "cos1": ref_data(K, 1.0), "normal": rng.normal(0, 1, K).astype(np.float32),
            "zeros": np.zeros(K, np.float32)}
    ties = rng.normal(0, 0.1, K).astype(np.float32)
    ties[5::256] = -3.0
    ties[9::256] = 3.0
    acts["ties"] = ties
    for name, x in acts.items():
        out[f"act_{name}"] = x
        out[f"q8_K_{name}"] = ref.quantize_row_q8_K(x)
        out[f"q8_0_{name}"] = ref.quantize_row_q8_0(x)
    # --- weights: the reference's own quantizer on its own synthetic data + random valid blocks -----
    nrows = 4
    for t in QUANT_TYPES:
        n = TYPE_NAMES[t]
        w_f32 = np.stack([ref_data(K, 7.0 + r) * 0.25 for r in range(nrows)])
        wq = ref.quantize_weights(t, w_f32)
        wr = rand_blocks(t, nrows, K, rng)
        out[f"w_{n}_quantized"], out[f"w_{n}_random"] = wq, wr
        rs = row_size(t, K)
        for tag, blocks in (("quantized", wq), ("random", wr)):
            out[f"deq_{n}_{tag}"] = np.stack([ref.dequantize_row(t, blocks[r * rs:(r + 1) * rs], K) for r in range(nrows)])
            for aname in ("cos1", "normal"):
                a = ref.quantize_row_q8_0(acts[aname]) if vec_dot_type(t) == Q8_0 else ref.quantize_row_q8_K(acts[aname])
                out[f"dot_{n}_{tag}_{aname}_scalar"] = np.array([ref.vec_dot(t, K, blocks[r * rs:(r + 1) * rs], a) for r in range(nrows)], np.float32)
                out[f"dot_{n}_{tag}_{aname}_avx2"] = np.array([avx2.vec_dot(t, K, blocks[r * rs:(r + 1) * rs], a) for r in range(nrows)], np.float32)
        x2 = np.stack([acts["cos1"], acts["normal"]])
        out[f"mulmat_{n}"] = ref.mul_mat(t, wr, K, nrows, x2)
    # --- layer ops ---------------------------------------------------------------------------------
    x = rng.normal(0, 2, (2, 512)).astype(np.float32)
    w = (1 + rng.normal(0, 0.02, 512)).astype(np.float32)
    out["rms_x"], out["rms_w"] = x, w
    out["rms_y_eps1e-5"] = ref.rms_norm(x, w, 1e-5)
    out["rms_y_eps1e-6_now"] = ref.rms_norm(x, None, 1e-6)
    xr = rng.normal(0, 1, (3, 4, 128)).astype(np.float32)
    pos = np.array([0, 17, 4095], np.int32)
    ff = (1 + rng.uniform(0, 7, 64)).astype(np.float32)
    out["rope_x"], out["rope_pos"], out["rope_ff"] = xr, pos, ff
    out["rope_norm_ff"] = ref.rope(xr, pos, freq_factors=ff, mode=0, freq_base=500000.0)
    out["rope_neox"] = ref.rope(xr, pos, freq_factors=None, mode=2, freq_base=1000000.0)
    sx = rng.normal(0, 3, (4, 2, 64)).astype(np.float32)
    sm = np.zeros((2, 64), np.float32)
    sm[0, 40:] = -np.inf
    sm[1, 41:] = -np.inf
    out["softmax_x"], out["softmax_mask"] = sx, sm
    out["softmax_y"] = ref.soft_max_ext(sx, sm, 0.08838834764831845)
    g, u = rng.normal(0, 3, 512).astype(np.float32), rng.normal(0, 1, 512).astype(np.float32)
    out["silu_g"], out["silu_u"], out["silu_y"] = g, u, ref.silu_mul(g, u)
    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), **out)

    # --- whole tiny models: greedy decode with the reference CPU backend -----------------------------
    for arch, name in ((0, "llama"), (1, "qwen2")):
        mrng = np.random.default_rng(77 + arch)
        kw = dict(arch=arch, n_layer=2, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320, n_ctx=64, rope_freqs=(arch == 0))
        d = tiny_model(mrng, quantize=ref.quantize_weights, **kw)
        h = avx2.model_new(d)
        prompt = mrng.integers(0, d.n_vocab, 6).astype(np.int32)
        hid, logits = avx2.model_eval(h, d, tokens=prompt, pos0=0, n_threads=2)
        toks, all_logits = [int(np.argmax(logits))], [logits.copy()]
        for i in range(10):
            hid, logits = avx2.model_eval(h, d, tokens=np.array([toks[-1]], np.int32), pos0=len(prompt) + i, n_threads=2)
            toks.append(int(np.argmax(logits)))
            all_logits.append(logits.copy())
        avx2.model_free(h)
        np.savez_compressed(os.path.join(HERE, f"tiny_{name}_decode.npz"), prompt=prompt, tokens=np.array(toks, np.int32),
                            logits=np.stack(all_logits), **desc_to_arrays(d))
    print("golden written:", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
