"""The reference's UNMODIFIED driver (gpt_params_parse -> llama_init_from_gpt_params -> llama_decode, built from
/root/reference by oracle/Makefile with the committed gate patch) on GGUF files written by prima_cpp_amd/gguf.py.

CPU part (this file): `-ngl 0` reproduces the committed goldens bit for bit. The goldens were produced by a graph the
builder wrote after build_llama / build_qwen2 (oracle/ref_ops.c); this pins them - and the GGUF writer - against the real
llama_decode. The GPU part (-ngl 99 through the MI355 plug-in) is tests/test_gpu_llama_decode.py."""
import os

import numpy as np
import pytest

from _bind import llama_driver_path, run_llama_driver, write_gguf_from_arrays

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(llama_driver_path() is None, reason="oracle/_ref/llama-ref-driver-* not built")


@pytest.mark.parametrize("name", ["llama", "qwen2"])
def test_llama_decode_cpu_reproduces_goldens(name, tmp_path):
    z = np.load(os.path.join(HERE, "golden", f"tiny_{name}_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / f"tiny_{name}.gguf"), z)
    # the goldens come from the avx2 flavour (tests/golden/make_golden.py): same flavour -> same bits
    toks, logits, stats = run_llama_driver(path, z["prompt"], len(z["tokens"]), ngl=0, n_ctx=int(z["hp_n_ctx"]), flavour="avx2")
    assert toks.tolist() == z["tokens"].tolist()
    assert np.array_equal(logits, z["logits"]), np.abs(logits - z["logits"]).max()
    assert stats["ngl"] == 0
    if llama_driver_path("avx512") and "avx512" in open("/proc/cpuinfo").read():
        # another ISA path of the same reference differs by summation order only (cf. tests/test_oracle_vs_ref.py)
        toks2, logits2, _ = run_llama_driver(path, z["prompt"], len(z["tokens"]), ngl=0, n_ctx=int(z["hp_n_ctx"]), flavour="avx512",
                                             force=z["tokens"][:-1])
        assert np.abs(logits2 - z["logits"]).max() < 1e-4


def test_gguf_roundtrip(tmp_path):
    from prima_cpp_amd import gguf as G
    p = G.write_synthetic_model(str(tmp_path / "m.gguf"), arch=1, n_layer=2, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=320)
    f = G.GGUFFile(p)
    assert f.kv["general.architecture"] == "qwen2" and f.kv["qwen2.block_count"] == 2
    assert f.kv["tokenizer.ggml.model"] == "no_vocab"
    t, shape, data = f.tensors["blk.1.ffn_down.weight"]
    assert shape == (512, 256) and data.nbytes == G.tensor_nbytes(t, shape)
    assert "blk.0.attn_q.bias" in f.tensors and f.tensors["output.weight"][0] == G.Q6_K
    f.close()


@pytest.mark.parametrize("name", ["llama", "qwen2"])
@pytest.mark.parametrize("extra", [[], ["-fa"], ["-ctk", "q8_0"]], ids=["default", "flash-attn", "q8_0-k"])
def test_graph_reuse_patch_leaves_the_reference_cpu_decode_unchanged(name, extra, tmp_path, monkeypatch):
    """oracle/ref_patches/graph_reuse.patch, judged on the reference's OWN CPU backend: with LLAMA_MI355_GRAPH_REUSE=1 libllama keeps
    the previous single-token graphs and scheduler allocations and only moves the KV-store views to the new cache head - 40 greedy tokens (the n_kv
    bucket of 32 is crossed: a rebuild in the middle of the run) must give the same tokens and the same logits, bit for bit, as the unpatched path;
    also with cells removed and a defragmentation queued mid-run (the reuse state must be dropped when the cache graph runs)."""
    z = np.load(os.path.join(HERE, "golden", f"tiny_{name}_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / f"tiny_{name}.gguf"), z)
    n_ctx = int(z["hp_n_ctx"])
    n_gen = min(40, n_ctx - len(z["prompt"]) - 1)
    monkeypatch.delenv("LLAMA_MI355_GRAPH_REUSE", raising=False)
    t0, l0, _ = run_llama_driver(path, z["prompt"], n_gen, ngl=0, n_ctx=n_ctx, flavour="avx2", extra_args=extra)
    monkeypatch.setenv("LLAMA_MI355_GRAPH_REUSE", "1")
    t1, l1, st = run_llama_driver(path, z["prompt"], n_gen, ngl=0, n_ctx=n_ctx, flavour="avx2", extra_args=extra)
    import re
    m = re.search(r"graph-reuse patch\): (\d+) of (\d+) single-token decodes reused", st["stderr"])
    assert m and int(m.group(1)) >= int(m.group(2)) - 3 and int(m.group(2)) >= n_gen - 1, st["stderr"][-500:]     # (rebuilt at the start and at each 32-cell bucket)
    assert t1.tolist() == t0.tolist()
    assert np.array_equal(l1, l0), np.abs(l1 - l0).max()
    if extra:
        return
    monkeypatch.setenv("REFDRV_RM", "3,2,5")
    monkeypatch.setenv("REFDRV_DEFRAG", "5")
    t3, l3, _ = run_llama_driver(path, z["prompt"], 12, ngl=0, n_ctx=n_ctx, flavour="avx2")
    monkeypatch.delenv("LLAMA_MI355_GRAPH_REUSE")
    t2, l2, _ = run_llama_driver(path, z["prompt"], 12, ngl=0, n_ctx=n_ctx, flavour="avx2")
    assert t3.tolist() == t2.tolist() and np.array_equal(l3, l2)


def test_graph_reuse_patch_drops_the_kept_graphs_on_a_context_shift(tmp_path, monkeypatch):
    """A context shift in mid-generation (llama_kv_cache_seq_rm + seq_add -> build_k_shift runs on the scheduler inside the next llama_decode): the
    kept single-token graphs must be dropped and rebuilt - same tokens and logits as the unpatched path, on the reference's own CPU backend."""
    z = np.load(os.path.join(HERE, "golden", "tiny_llama_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / "tiny_llama.gguf"), z)
    monkeypatch.setenv("REFDRV_SHIFT", "4,2,3")
    monkeypatch.delenv("LLAMA_MI355_GRAPH_REUSE", raising=False)
    t0, l0, _ = run_llama_driver(path, z["prompt"], 14, ngl=0, n_ctx=64, flavour="avx2")
    monkeypatch.setenv("LLAMA_MI355_GRAPH_REUSE", "1")
    t1, l1, st = run_llama_driver(path, z["prompt"], 14, ngl=0, n_ctx=64, flavour="avx2")
    assert "graph-reuse patch" in st["stderr"]
    assert t1.tolist() == t0.tolist() and np.array_equal(l1, l0)
