"""Whole-model parity on SURVEY §8d fixtures: N(0, 1/K) weights through the REFERENCE's quantizer
(`ggml_quantize_chunk`), Q4_K_M mixture, 16-token prompt + 128 greedy tokens, at three shapes - a small 8-layer model, the
Llama-3-8B shape and 8 layers of the Llama-3-70B shape - each in two flavours (tests/_fixtures8d.py):

  * peaked : top-1 / top-2 margins >> the int8 re-quantization step. The reference's scalar and AVX2 builds agree 128/128 on it, and
             the plug-in (through the reference's unmodified llama_decode, default and --flash-attn graphs) and the resident engine must
             produce the SAME 128 tokens - equality, no margin gate (north_star: "bit-exact token IDs under greedy decode").
  * plain  : flat logits (the fixture §8d specifies). Two float summation orders of the same arithmetic - the reference's own ISA
             builds included - flip int8 roundings and drift ~1 % apart at the logits (DESIGN.md "parity"), so the un-gated match rate
             and the margin statistics are REPORTED next to "reference AVX2 vs reference scalar", and the logits are held to the
             reference-against-itself yardstick.

PM355_8D_SIZES=small,8b,70b8 selects the shapes (default: all; generation + the scalar CPU runs of the two large shapes take
~6 minutes of host time on the GPU box)."""
import os
import time

import numpy as np
import pytest

import _fixtures8d as F
from _bind import Ref, best_ref_flavour, have_ref, llama_driver_path, run_llama_driver

pytestmark = pytest.mark.gpu
GPU_ARGS = ["--keep-out-in-cuda"]
N_GEN, N_PROMPT = 128, 16

SHAPES = {
    "small": dict(n_layer=8, n_embd=1024, n_head=8, n_head_kv=4, n_ff=2816, n_vocab=8192, is_70b=False),
    "8b": dict(n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256, is_70b=False),
    "70b8": dict(n_layer=8, n_embd=8192, n_head=64, n_head_kv=8, n_ff=28672, n_vocab=128256, is_70b=True),
    # the Llama-3-70B ATTENTION shape (64 query / 8 KV heads of 128: the GQA 8:1 groups the long-context matrix-core kernel tiles by) on two layers with a
    # narrow ffn and vocabulary: what the CPU reference can carry through an 8200-token prompt in minutes (70b8 did not finish it in 50 on 16 threads)
    "70bh": dict(n_layer=2, n_embd=8192, n_head=64, n_head_kv=8, n_ff=8192, n_vocab=32000, is_70b=True),
}
SIZES = [s for s in os.environ.get("PM355_8D_SIZES", "small,8b,70b8").split(",") if s in SHAPES]
N_CTX = 256


def _nmse(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30))


def _threads():
    return F.n_threads()


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    if llama_driver_path() is None or not have_ref("scalar"):
        pytest.skip("oracle/_ref not built")
    return True


class _Files:
    """GGUFs + the reference CPU runs of one shape, built on first use and shared by the tests of the module."""

    def __init__(self, tmp):
        self.tmp, self.files, self.cpu = tmp, {}, {}
        self.ref = Ref(best_ref_flavour())

    def path(self, size, peaked):
        key = (size, peaked)
        if key not in self.files:
            plain = os.path.join(self.tmp, f"s8d_{size}_plain.gguf")
            t0 = time.time()
            if (size, False) not in self.files:
                F.write_model(plain, self.ref, tag=size, **SHAPES[size])
                self.files[(size, False)] = plain
                print(f"\n[8d {size}] plain GGUF {os.path.getsize(plain) / 1e9:.2f} GB quantized by the reference in {time.time() - t0:.0f} s ({_threads()} threads)")
            if peaked:
                t0 = time.time()
                pk = os.path.join(self.tmp, f"s8d_{size}_peaked.gguf")
                info = F.copy_with_new_head(plain, pk, self.ref, True, tag=size)
                self.files[key] = pk
                print(f"[8d {size}] peaked head (embd sigma {info['embd_sigma']:.2f}, gain {info['peak_gain']:.3f}) in {time.time() - t0:.0f} s")
        return self.files[key]

    def cpu_run(self, size, peaked, flavour, force=None, extra=()):
        key = (size, peaked, flavour, None if force is None else tuple(int(t) for t in force), tuple(extra))
        if key not in self.cpu:
            V = SHAPES[size]["n_vocab"]
            t0 = time.time()
            self.cpu[key] = run_llama_driver(self.path(size, peaked), F.prompt_tokens(V, N_PROMPT), N_GEN, ngl=0, n_ctx=N_CTX, threads=_threads(),
                                             flavour=flavour, force=force, extra_args=list(extra), timeout=3000)
            st = self.cpu[key][2]
            print(f"[8d {size} {'peaked' if peaked else 'plain'}] reference CPU {flavour}: {st['decode_tok_s']:.2f} tok/s ({time.time() - t0:.0f} s)")
        return self.cpu[key]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    f = _Files(str(tmp_path_factory.mktemp("s8d")))
    yield f
    for p in f.files.values():
        try:
            os.unlink(p)
        except OSError:
            pass


def _margins(lg):
    srt = np.sort(lg, axis=1)
    return srt[:, -1] - srt[:, -2], lg.std(axis=1)


def _engine_greedy(path, size, prompt, n_gen, n_ctx=N_CTX):
    return F.engine_greedy(path, SHAPES[size], prompt, n_gen, n_ctx)


@pytest.mark.parametrize("mode", ["plugin", "plugin-fa", "engine"])
@pytest.mark.parametrize("size", SIZES)
def test_peaked_fixture_greedy_tokens_identical(gpu, files, size, mode):
    V = SHAPES[size]["n_vocab"]
    prompt = F.prompt_tokens(V, N_PROMPT)
    path = files.path(size, True)
    ts, ls, _ = files.cpu_run(size, True, "scalar")
    ta, la, _ = files.cpu_run(size, True, "avx2")
    expect = [F.peaked_next(prompt[-1], V)]
    for _ in range(N_GEN - 1):
        expect.append(F.peaked_next(expect[-1], V))
    marg, sd = _margins(ls)
    print(f"\n[8d {size} peaked] reference scalar: top1-top2 margin / sigma min {(marg / sd).min():.2f} median {np.median(marg / sd):.2f}; "
          f"AVX2 vs scalar: tokens {(ta == ts).sum()}/{N_GEN}, logits NMSE {_nmse(la, ls):.2e}, max |dlogit| / sigma {(np.abs(la - ls).max(axis=1) / sd).max():.3f}")
    # the fixture itself: the reference's two ISA builds agree on every token, and the tokens are the ones the head was built to emit
    assert ts.tolist() == expect
    assert ta.tolist() == ts.tolist()
    if mode == "engine":
        tg, lg = _engine_greedy(path, size, prompt, N_GEN)
    else:
        tg, lg, st = run_llama_driver(path, prompt, N_GEN, ngl=99, n_ctx=N_CTX, threads=_threads(), timeout=1800,
                                      extra_args=GPU_ARGS + (["-fa"] if mode == "plugin-fa" else []))
        assert "MI355X0" in st["stderr"]
    err = np.abs(lg - ls).max(axis=1)
    print(f"[8d {size} peaked {mode}] tokens {(tg == ts).sum()}/{N_GEN} identical to the reference CPU; logits NMSE {_nmse(lg, ls):.2e}; "
          f"max |dlogit| / sigma {(err / sd).max():.3f}; smallest margin / observed error {(marg / np.maximum(err, 1e-30)).min():.1f}")
    assert tg.tolist() == ts.tolist()                     # 128/128, un-gated
    # logits: within 1e-3 relative, or what the reference's other ISA build differs by on the same file (flash attention keeps P in f32
    # and the reference's CPU flash kernel an F16 accumulator: its own backend tolerance applies there)
    nm_ref = _nmse(la, ls)
    assert _nmse(lg, ls) < max(1e-6, 3 * nm_ref) * (10 if mode == "plugin-fa" else 1), (_nmse(lg, ls), nm_ref)


@pytest.mark.parametrize("mode", ["plugin", "plugin-fa", "engine", "plugin-i8", "engine-i8"])
def test_peaked_fixture_long_context_tokens_identical(gpu, files, mode, monkeypatch):
    """The same equality beyond the long-context threshold (320 cells since round 5): a 700-token prompt, then 24 greedy tokens whose attention runs on
    the matrix-core kernel over cached cells (attn_flash_mfma.hip: rope + KV store in the QKV epilogue, keys split over workgroups,
    spans merged in the launch) - through the plug-in's default graph and on the resident engine."""
    if "small" not in SIZES:
        pytest.skip("PM355_8D_SIZES without 'small'")
    size, n_prompt, n_gen, n_ctx = "small", 700, 24, 1024
    V = SHAPES[size]["n_vocab"]
    prompt = F.prompt_tokens(V, n_prompt)
    path = files.path(size, True)
    ts, ls, _ = run_llama_driver(path, prompt, n_gen, ngl=0, n_ctx=n_ctx, threads=_threads(), flavour="scalar", timeout=3000)
    expect = [F.peaked_next(prompt[-1], V)]
    for _ in range(n_gen - 1):
        expect.append(F.peaked_next(expect[-1], V))
    assert ts.tolist() == expect
    i8 = mode.endswith("-i8")
    if mode.startswith("engine"):
        if i8:
            monkeypatch.setenv("PM355_PROMPT_I8", "1")
        tg, lg = _engine_greedy(path, size, prompt, n_gen, n_ctx=n_ctx)
    else:
        env = {"GGML_MI355_DEBUG_PLAN": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"}
        if i8:
            env["GGML_MI355_PROMPT_I8"] = "1"
        # (plugin-fa: --flash-attn graphs - row-major V cache, F16 mask - on the same matrix-core kernel since round 4)
        tg, lg, st = run_llama_driver(path, prompt, n_gen, ngl=99, n_ctx=n_ctx, threads=_threads(), timeout=1800, extra_args=GPU_ARGS + (["-fa"] if mode == "plugin-fa" else []), env=env)
        assert "cached-split" in st["stderr"], st["stderr"][-2000:]
    print(f"\n[8d small peaked, 700-token prompt, {mode}] tokens {(tg == ts).sum()}/{n_gen} identical to the reference CPU; logits NMSE {_nmse(lg, ls):.2e}")
    assert tg.tolist() == ts.tolist()
    if i8:
        # the prompt batch on the integer matrix cores (mmq_big.hip): Q8_K activations, the CPU's integer block sums - the prompt pass is in the
        # tier of the single-token path (what is left: f32 summation orders and the F16 roundings of the attention, as for any decoded token)
        ta, la, _ = run_llama_driver(path, prompt, n_gen, ngl=0, n_ctx=n_ctx, threads=_threads(), flavour="avx2", timeout=3000)
        print(f"   reference AVX2 vs scalar on the same run: logits NMSE {_nmse(la, ls):.2e}")
        assert _nmse(lg, ls) < max(1e-6, 3 * _nmse(la, ls)), (_nmse(lg, ls), _nmse(la, ls))
    else:
        # (the 700-token batch runs the MFMA prefill path - F16 activations x dequantized F16 weights - and the reference the int8 path:
        # north_star's 1e-3 tier for fp16 accumulation)
        assert _nmse(lg, ls) < 1e-3


@pytest.mark.parametrize("kv", [["-ctk", "q8_0", "-ctv", "q8_0"], ["-ctk", "q8_0"]], ids=["q8_0-kv", "q8_0-k"])
def test_peaked_fixture_long_context_q8_0_cache_tokens_identical(gpu, files, kv):
    """Round 5: `-fa -ctk q8_0 [-ctv q8_0]` beyond the long-context threshold - a 700-token prompt, then 24 greedy tokens whose attention
    runs rope + quantizing KV store + the matrix-core kernel over the cached Q8_0 cells (attn_flash_mfma.hip) - against the reference CPU running the same
    flash-attention graph on the same quantized cache types: the same 24 tokens, logits within the reference's tolerance for the flash-attention op."""
    if "small" not in SIZES:
        pytest.skip("PM355_8D_SIZES without 'small'")
    size, n_prompt, n_gen, n_ctx = "small", 700, 24, 1024
    V = SHAPES[size]["n_vocab"]
    prompt = F.prompt_tokens(V, n_prompt)
    path = files.path(size, True)
    args = ["-fa"] + kv
    ts, ls, _ = run_llama_driver(path, prompt, n_gen, ngl=0, n_ctx=n_ctx, threads=_threads(), flavour=best_ref_flavour(), timeout=3000, extra_args=args)
    expect = [F.peaked_next(prompt[-1], V)]
    for _ in range(n_gen - 1):
        expect.append(F.peaked_next(expect[-1], V))
    assert ts.tolist() == expect
    tg, lg, st = run_llama_driver(path, prompt, n_gen, ngl=99, n_ctx=n_ctx, threads=_threads(), timeout=1800, extra_args=GPU_ARGS + args,
                                  env={"GGML_MI355_DEBUG_PLAN": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"})
    assert "split" in st["stderr"], st["stderr"][-2000:]
    # yardstick: the same run on the one-workgroup-per-head kernel that serves short contexts (GGML_MI355_ATTN_MFMA=0: the reference's block-wise integer
    # K.q). Both see the same 700 prompt cells - stored by the F16 prompt GEMMs and then QUANTIZED, so they differ from the CPU's cells by whole Q8_0
    # steps wherever a value sat near a rounding boundary: that, not the attention kernel, sets the distance to the CPU
    to, lo, so = run_llama_driver(path, prompt, n_gen, ngl=99, n_ctx=n_ctx, threads=_threads(), timeout=1800, extra_args=GPU_ARGS + args,
                                  env={"GGML_MI355_ATTN_MFMA": "0", "GGML_MI355_DEBUG_PLAN": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"})
    print(f"\n[8d small peaked, 700-token prompt, -fa {' '.join(kv)}] tokens {(tg == ts).sum()}/{n_gen} identical to the reference CPU; logits NMSE {_nmse(lg, ls):.2e} "
          f"(one-workgroup-per-head kernel on the same cells: {_nmse(lo, ls):.2e}; the two kernels against each other: {_nmse(lg, lo):.2e})")
    assert tg.tolist() == ts.tolist() and to.tolist() == ts.tolist()
    # (kernel against kernel: the first token's logits are the same to 1e-7; from then on each run attends the cells IT stored - a rotated k that
    #  differs in the last F16-operand bit moves a Q8_0 value by a whole step now and then: ~1e-4 after 24 tokens. The op test holds the kernels to 1e-5.)
    assert _nmse(lg[0], lo[0]) < 1e-6 and _nmse(lg, lo) < 1e-3
    assert _nmse(lg, ls) < max(1e-3, 1.5 * _nmse(lo, ls))


def test_peaked_fixture_8k_prompt_tokens_identical(gpu, files):
    """whole-model equality beyond 8k cached cells. An 8200-token prompt - prompt chunks on the prefill path, then 16 greedy tokens
    whose attention runs the matrix-core kernel over > 8k cached cells - against the reference CPU through the plug-in's default graph and on the
    resident engine: the same 16 tokens, logits within the F16-accumulation tier. Default: the small 8-layer shape (the CPU reference needs ~1 min
    for the prompt); PM355_8D_LONG=8b runs it at the Llama-3-8B shape (32 layers, 32 query / 8 KV heads: ~20 min of host time for the CPU's
    prompt pass; the round-4 attempt was cut off by the box limit after 25 minutes inside that pass - profiles/r04_long_context.txt - and has not been
    repeated), PM355_8D_LONG=70bh at the Llama-3-70B head shape (64 query / 8 KV heads of 128, two layers, n_ff 8192: profiles/r06_parity_long_context.txt;
    PM355_8D_LONG=70b8 - eight full 70B layers - did not get through the CPU reference's prompt pass in 50 minutes of box time in round 6)."""
    size = os.environ.get("PM355_8D_LONG", "small")
    if size not in SHAPES:
        pytest.skip("PM355_8D_LONG names no shape")
    n_prompt, n_gen, n_ctx = 8200, 16, 8448
    V = SHAPES[size]["n_vocab"]
    prompt = F.prompt_tokens(V, n_prompt)
    path = files.path(size, True)
    t0 = time.time()
    ta, la, sa = run_llama_driver(path, prompt, n_gen, ngl=0, n_ctx=n_ctx, threads=_threads(), flavour=best_ref_flavour(), timeout=3000 if size != "small" else 900)
    print(f"\n[8d {size} peaked, 8200-token prompt] reference CPU ({best_ref_flavour()}) in {time.time() - t0:.0f} s")
    expect = [F.peaked_next(prompt[-1], V)]
    for _ in range(n_gen - 1):
        expect.append(F.peaked_next(expect[-1], V))
    assert ta.tolist() == expect
    tg, lg, st = run_llama_driver(path, prompt, n_gen, ngl=99, n_ctx=n_ctx, threads=_threads(), timeout=1800, extra_args=GPU_ARGS,
                                  env={"GGML_MI355_DEBUG_PLAN": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"})
    assert "cached-split" in st["stderr"], st["stderr"][-2000:]
    te, le = _engine_greedy(path, size, prompt, n_gen, n_ctx=n_ctx)
    for mode, t_, l_ in (("plugin", tg, lg), ("engine", te, le)):
        print(f"[8d {size} peaked, 8200-token prompt, {mode}] tokens {(t_ == ta).sum()}/{n_gen} identical to the reference CPU; logits NMSE {_nmse(l_, la):.2e}")
        assert t_.tolist() == ta.tolist()
        assert _nmse(l_, la) < 1e-3


def test_peaked_fixture_at_depth(gpu, tmp_path):
    """VERDICT r4 item 5 - whole-model equality at the depth the bench times. Opt-in (PM355_8D_DEPTH=<layers>, e.g. 32 or 80; recorded under
    profiles/r05_parity_8d.txt): the peaked fixture at the Llama-3-70B shape with that many layers (every layer its own N(0, 1/K) weights through the
    reference's quantizer, Q4_K_M mixture), 16-token prompt + 32 greedy tokens: the reference CPU (its fastest ISA build, -ngl 0) against the plug-in
    through the reference's llama_decode and against the resident engine - the same 32 tokens, logits at the reference-against-itself tier."""
    # Round 6: 16 layers run by DEFAULT under `pytest -m gpu` (8.5 GB GGUF, about a minute), so the driver's GPU tier carries depth beyond the 8-layer fixture;
    # PM355_8D_DEPTH=80 is the full model (profiles/r05_parity_8d.txt), PM355_8D_DEPTH=0 skips.
    depth = int(os.environ.get("PM355_8D_DEPTH", "16") or 0)
    if depth <= 0:
        pytest.skip("PM355_8D_DEPTH=0")
    shape = dict(SHAPES["70b8"], n_layer=depth)
    n_gen = 32
    V = shape["n_vocab"]
    ref = Ref(best_ref_flavour())
    path = str(tmp_path / f"s8d_70b{depth}_peaked.gguf")
    t0 = time.time()
    F.write_model(path, ref, tag=f"70b{depth}", peaked=True, **shape)
    print(f"\n[8d 70b x {depth} layers] peaked GGUF {os.path.getsize(path) / 1e9:.2f} GB quantized by the reference in {time.time() - t0:.0f} s ({_threads()} threads)")
    try:
        prompt = F.prompt_tokens(V, N_PROMPT)
        expect = [F.peaked_next(prompt[-1], V)]
        for _ in range(n_gen - 1):
            expect.append(F.peaked_next(expect[-1], V))
        t0 = time.time()
        tr, lr, sr = run_llama_driver(path, prompt, n_gen, ngl=0, n_ctx=N_CTX, threads=_threads(), flavour=best_ref_flavour(), timeout=3000)
        marg, sd = _margins(lr)
        print(f"[8d 70b x {depth}] reference CPU ({best_ref_flavour()}): {sr['decode_tok_s']:.2f} tok/s ({time.time() - t0:.0f} s); top1-top2 margin / sigma min {(marg / sd).min():.2f}")
        assert tr.tolist() == expect
        # yardstick: the reference against itself - its AVX2 build on the same file (another summation order of the same integer arithmetic)
        nm_ref = None
        if have_ref("avx2") and best_ref_flavour() != "avx2":
            ta, la, _ = run_llama_driver(path, prompt, n_gen, ngl=0, n_ctx=N_CTX, threads=_threads(), flavour="avx2", timeout=3000)
            nm_ref = _nmse(la, lr)
            print(f"[8d 70b x {depth}] reference AVX2 vs {best_ref_flavour()}: tokens {(ta == tr).sum()}/{n_gen}, logits NMSE {nm_ref:.2e}")
        tg, lg, st = run_llama_driver(path, prompt, n_gen, ngl=99, n_ctx=N_CTX, threads=_threads(), timeout=1800, extra_args=GPU_ARGS)
        assert "MI355X0" in st["stderr"]
        te, le = F.engine_greedy(path, shape, prompt, n_gen, N_CTX)
        for mode, t_, l_ in (("plugin", tg, lg), ("engine", te, le)):
            err = np.abs(l_ - lr).max(axis=1)
            print(f"[8d 70b x {depth} peaked {mode}] tokens {(t_ == tr).sum()}/{n_gen} identical to the reference CPU; logits NMSE {_nmse(l_, lr):.2e}; "
                  f"max |dlogit| / sigma {(err / sd).max():.3f}; smallest margin / observed error {(marg / np.maximum(err, 1e-30)).min():.1f}")
            assert t_.tolist() == tr.tolist()
            # (the 16-token prompt runs the MFMA prefill path - F16 activations x dequantized F16 weights - and its cells stay in the cache of every
            #  decoded token: north_star's 1e-3 tier for fp16 accumulation, at any depth)
            assert _nmse(l_, lr) < 1e-3, _nmse(l_, lr)
            # the DISCRIMINATING quantity (token equality on the peaked fixture is weak by construction: the margin is ~50 x the observed error): the logits
            # may sit no further from the reference than 1.5 x the distance of the reference's own AVX2 build from its AVX512 build on this file
            if nm_ref is not None:
                assert _nmse(l_, lr) <= 1.5 * max(nm_ref, 2e-5), (mode, _nmse(l_, lr), nm_ref)
        if st.get("decode_tok_s"):
            print(f"[8d 70b x {depth}] plug-in decode {st['decode_tok_s']:.1f} tok/s")
        # layer by layer, BEFORE a flipped token could cascade: every layer's output row (l_out-<il>, then result_norm) of the first
        # single-token decode, captured through the scheduler's eval callback in the reference's own driver - plug-in against the reference CPU, next to the
        # reference's AVX2 build against the same (its own summation-order distance, growing with depth the same way)
        _, _, s_ref = run_llama_driver(path, prompt, 2, ngl=0, n_ctx=N_CTX, threads=_threads(), flavour=best_ref_flavour(), timeout=3000, lout=True)
        _, _, s_gpu = run_llama_driver(path, prompt, 2, ngl=99, n_ctx=N_CTX, threads=_threads(), timeout=1800, extra_args=GPU_ARGS, lout=True)
        lo_r, lo_g = s_ref["lout"], s_gpu["lout"]
        assert sorted(lo_r) == sorted(lo_g) == [-1] + list(range(depth)), (sorted(lo_r), sorted(lo_g))
        lo_a = None
        if have_ref("avx2") and best_ref_flavour() != "avx2":
            _, _, s_a = run_llama_driver(path, prompt, 2, ngl=0, n_ctx=N_CTX, threads=_threads(), flavour="avx2", timeout=3000, lout=True)
            lo_a = s_a["lout"]
        order = list(range(depth)) + [-1]
        nm_g = np.array([_nmse(lo_g[l], lo_r[l]) for l in order])
        nm_a = np.array([_nmse(lo_a[l], lo_r[l]) for l in order]) if lo_a else None
        print(f"[8d 70b x {depth}] per-layer l_out NMSE of the first decoded token vs the reference CPU ({best_ref_flavour()}); last entry = result_norm")
        print("   plug-in : " + " ".join(f"{v:.1e}" for v in nm_g))
        if nm_a is not None:
            print("   ref AVX2: " + " ".join(f"{v:.1e}" for v in nm_a))
        assert np.isfinite(nm_g).all() and nm_g.max() < 1e-3, nm_g.max()
        if nm_a is not None:
            # no layer further from the reference than a few times the reference's own two builds are from each other at that depth (floor: the first layers,
            # where the AVX2 build's distance is a handful of rounding decisions)
            assert (nm_g <= 5.0 * np.maximum(nm_a, 1e-6)).all(), (nm_g, nm_a)
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


@pytest.mark.parametrize("size", SIZES)
def test_plain_fixture_statistics(gpu, files, size):
    V = SHAPES[size]["n_vocab"]
    prompt = F.prompt_tokens(V, N_PROMPT)
    path = files.path(size, False)
    tg, lg, st = run_llama_driver(path, prompt, N_GEN, ngl=99, n_ctx=N_CTX, threads=_threads(), extra_args=GPU_ARGS, timeout=1800)
    ts, ls, _ = files.cpu_run(size, False, "scalar", force=tg[:-1])          # teacher-forced with the GPU's tokens
    ta, la, _ = files.cpu_run(size, False, "avx2", force=tg[:-1])
    marg, sd = _margins(ls)
    err, spread = np.abs(lg - ls).max(axis=1), np.abs(la - ls).max(axis=1)
    bad = [i for i in range(N_GEN) if tg[i] != ts[i]]
    bad_a = [i for i in range(N_GEN) if ta[i] != ts[i]]
    print(f"\n[8d {size} plain] plug-in {st['decode_tok_s']:.1f} tok/s; margin / sigma min {(marg / sd).min():.4f} median {np.median(marg / sd):.3f}\n"
          f"   plug-in vs scalar: tokens {N_GEN - len(bad)}/{N_GEN} (un-gated), logits NMSE {_nmse(lg, ls):.2e}, max |dlogit| / sigma {(err / sd).max():.3f}\n"
          f"   AVX2    vs scalar: tokens {N_GEN - len(bad_a)}/{N_GEN} (un-gated), logits NMSE {_nmse(la, ls):.2e}, max |dlogit| / sigma {(spread / sd).max():.3f}\n"
          f"   mismatching steps (step, margin / sigma): plug-in {[(i, round(float(marg[i] / sd[i]), 4)) for i in bad]} AVX2 {[(i, round(float(marg[i] / sd[i]), 4)) for i in bad_a]}")
    nm_gpu, nm_ref = _nmse(lg, ls), _nmse(la, ls)
    assert nm_gpu < max(1e-6, 3 * nm_ref), (nm_gpu, nm_ref)
    assert err.max() <= max(1e-3 * float(np.abs(ls).max()), 3 * spread.max()), (err.max(), spread.max())
    for i in bad:                                       # a flipped argmax only inside the observed logit noise
        assert marg[i] <= 2 * err[i], (i, marg[i], err[i])
    # the plug-in must not flip more often than the reference's other ISA build does (+ binomial slack)
    assert len(bad) <= 2 * len(bad_a) + 6, (len(bad), len(bad_a))
