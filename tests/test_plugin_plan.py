"""The plug-in's graph lowering (prima_cpp_amd/csrc/ggml_graph_plan.h) exercised under the reference's REAL llama_decode on a
machine without a GPU: GGML_MI355_PLAN_ONLY=1 makes the plug-in register one pretend device backed by host memory whose
graph_compute plans every split, prints the plan and launches nothing (outputs are garbage by design - this is not a compute
path). What is checked: the single-token layer graph of build_llama / build_qwen2 lowers to 5 launches per layer
(QKV, attention, wo+residual, gate/up pair, down+residual; +3 node-equivalent launches in the last layer whose ADD output
aliases the mat-vec input) and the head to one launch; multi-token batches stay node-equivalent."""
import os
import re

import numpy as np
import pytest

from _bind import llama_driver_path, run_llama_driver, write_gguf_from_arrays

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(llama_driver_path() is None, reason="oracle/_ref/llama-ref-driver-* not built")
PLAN = re.compile(r"ggml-mi355 plan: (\d+) nodes -> (\d+) launches \((\d+) fused mat-vec, (\d+) attention, (\d+) node-equivalent; (\d+) nodes fused\) "
                  r"single_token=(\d) cell=(\d+) n_kv=(\d+) graphable=(\d)")


@pytest.mark.parametrize("name,nodes", [("llama", 66), ("qwen2", 72)])
def test_decode_graph_lowers_to_five_launches_per_layer(name, nodes, tmp_path):
    z = np.load(os.path.join(HERE, "golden", f"tiny_{name}_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / f"tiny_{name}.gguf"), z)
    _, _, st = run_llama_driver(path, z["prompt"][:3], 4, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda"],
                                env={"GGML_MI355_PLAN_ONLY": "1"}, flavour="avx2", timeout=120)
    plans = [tuple(int(x) for x in m.groups()) for m in PLAN.finditer(st["stderr"])]
    assert "plan-only" in st["stderr"] or plans, st["stderr"][-2000:]
    layer = [p for p in plans if p[0] == nodes]
    head = [p for p in plans if p[0] == 3]
    decode = [p for p in layer if p[6] == 1]
    assert len(decode) >= 3 and len(head) >= 3
    n_layer = int(z["hp_n_layer"])
    for p in decode:
        _, launches, gemv, attn, node_eq, fused, single, cell, n_kv, graphable = p
        assert attn == n_layer and gemv == 4 * n_layer and node_eq == 3 and launches == 5 * n_layer + 3, p
        assert fused == nodes - 3 and graphable == 1 and n_kv == 32
    # the KV cell advances by one per decoded token (the value that is written to the device before the captured graph is replayed)
    cells = [p[7] for p in decode]
    assert cells[0] == 0                                       # the warm-up token of llama_init_from_gpt_params (common/common.cpp:1959)
    assert cells[1:] == list(range(3, 3 + len(cells) - 1)), cells
    for p in head:
        assert p[1] == 1 and p[2] == 1, p
    # the 3-token prompt batch: nothing fused, not a hipGraph candidate
    prefill = [p for p in layer if p[6] == 0]
    assert prefill and all(p[2] == 0 and p[3] == 0 and p[9] == 0 for p in prefill), prefill
    # ... except the attention chain of every layer, which becomes one launch of the masked MFMA attention
    assert f"{n_layer} multi-token attention chain(s) -> MFMA masked attention" in st["stderr"]


@pytest.mark.parametrize("name", ["llama", "qwen2"])
def test_rms_norm_sum_of_squares_is_wired_from_the_producing_launch(name, tmp_path):
    """Round 5: wo / ffn_down (+ residual) leave per-workgroup partials of the next rms_norm's sum of squares; the planner wires every fused
    norm + mat-vec whose input row is the output of the launch right before it (ffn_gate | ffn_up after wo in every layer; wq | wk | wv after
    the previous layer's ffn_down). Opt-in: GGML_MI355_SS=1 switches the wiring on (measured -1.0 %: the default is off, every rms_norm prologue reduces its own row)."""
    z = np.load(os.path.join(HERE, "golden", f"tiny_{name}_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / f"tiny_{name}.gguf"), z)
    n_layer = int(z["hp_n_layer"])
    # (the LAST layer's wo output passes through GET_ROWS(inp_out_ids) + ADD nodes before its ffn_norm - src/llama.cpp:11141-11147 - so its
    #  ffn_gate | ffn_up launch reduces the row itself, and the head is a graph of its own)
    for ss, want in (("1", 2 * n_layer - 2), ("0", 0)):
        _, _, st = run_llama_driver(path, z["prompt"][:3], 3, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda"],
                                    env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1", "GGML_MI355_SS": ss}, flavour="avx2", timeout=120)
        blocks = st["stderr"].split("ggml-mi355 plan: ")
        decode = [b for b in blocks if "single_token=1" in b.split("\n")[0] and ") attention H=" in b]
        assert decode, st["stderr"][-2000:]
        for b in decode:
            assert b.count("sumsq<-producer") == want and b.count("sumsq->consumer") == want, b[:3000]


def test_flash_attn_decode_graph_lowers_to_five_launches_per_layer(tmp_path):
    """--flash-attn: CPY(V -> row of the row-major V cache), FLASH_ATTN_EXT(q, k, v, F16 mask) (llm_build_kv, src/llama.cpp:9705,
    :10075-10095) lower to the same five launches per layer; the once-per-graph F32 -> F16 cast of the KQ mask stays a node."""
    z = np.load(os.path.join(HERE, "golden", "tiny_llama_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / "tiny_llama.gguf"), z)
    _, _, st = run_llama_driver(path, z["prompt"][:3], 4, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda", "-fa"],
                                env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"}, flavour="avx2", timeout=120)
    assert "flash_attn   = 1" in st["stderr"]
    plans = [tuple(int(x) for x in m.groups()) for m in PLAN.finditer(st["stderr"])]
    n_layer = int(z["hp_n_layer"])
    decode = [p for p in plans if p[0] > 3 and p[6] == 1]
    assert len(decode) >= 3, st["stderr"][-2000:]
    for p in decode:
        nodes, launches, gemv, attn, node_eq, fused, single, cell, n_kv, graphable = p
        assert attn == n_layer and gemv == 4 * n_layer and node_eq == 1 and launches == 5 * n_layer + 1, p
        assert graphable == 1 and n_kv == 256                     # (flash attention pads the cells attended to 256, src/llama.cpp:18425)
    assert "KQ_mask (copy)" in st["stderr"] and "FLASH_ATTN_EXT" not in st["stderr"].split("single_token=1")[1].split("ggml-mi355 plan")[0]
    cells = [p[7] for p in decode]
    assert cells[0] == 0 and cells[1:] == list(range(3, 3 + len(cells) - 1)), cells


@pytest.mark.parametrize("kv", [["-ctk", "q8_0", "-ctv", "q8_0"], ["-ctk", "q8_0"], ["-ctv", "q8_0"]])
def test_quantized_kv_graphs_stay_on_the_plugin_and_lower(kv, tmp_path):
    """-ctk / -ctv q8_0 (with --flash-attn): the KV buffer lives in the MI355 buffer type, every node of the split is accepted by
    supports_op (one graph per layer window, nothing bounced to the CPU backend) and single tokens lower to five launches per layer."""
    z = np.load(os.path.join(HERE, "golden", "tiny_llama_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / "tiny_llama.gguf"), z)
    _, _, st = run_llama_driver(path, z["prompt"][:3], 4, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda", "-fa"] + kv,
                                env={"GGML_MI355_PLAN_ONLY": "1"}, flavour="avx2", timeout=120)
    assert "MI355X0 KV buffer size" in st["stderr"] and "q8_0" in st["stderr"]
    plans = [tuple(int(x) for x in m.groups()) for m in PLAN.finditer(st["stderr"])]
    n_layer = int(z["hp_n_layer"])
    decode = [p for p in plans if p[0] > 3 and p[6] == 1]
    assert len(decode) >= 3, st["stderr"][-2000:]
    for p in decode:
        nodes, launches, gemv, attn, node_eq, fused, single, cell, n_kv, graphable = p
        assert nodes == 59 and attn == n_layer and gemv == 4 * n_layer and launches == 5 * n_layer + 1 and graphable == 1, p
    # the 3-token prompt batch: one graph of all 59 nodes as well (CPY f32 -> Q8_0 and FLASH_ATTN_EXT on Q8_0 blocks are served)
    assert any(p[0] == 59 and p[6] == 0 for p in plans)


def test_flash_attn_prompt_batches_use_the_mfma_attention(tmp_path):
    z = np.load(os.path.join(HERE, "golden", "tiny_llama_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / "tiny_llama.gguf"), z)
    _, _, st = run_llama_driver(path, z["prompt"][:3], 2, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda", "-fa"],
                                env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"}, flavour="avx2", timeout=120)
    n_layer = int(z["hp_n_layer"])
    assert f"{n_layer} multi-token attention chain(s) -> MFMA masked attention" in st["stderr"]
    assert "batch attention T=3" in st["stderr"]


@pytest.mark.parametrize("fa", [False, True])
def test_round3_attention_block_rope_in_the_qkv_epilogue(fa, tmp_path):
    """Where every workgroup's wq | wk | wv row slices hold whole rotation pairs (N % 512 == 0) and the rope is NORM-mode, the attention
    block lowers to: ONE cos / sin table launch per graph (the per-layer rope_freqs copies hold the same numbers), the mat-vec launch
    with RoPE + KV store in its epilogue, and the attention over cached cells - still five launches per layer, plus the table."""
    from _bind import Ref, best_ref_flavour
    import _fixtures8d as F
    ref = Ref(best_ref_flavour())
    path = str(tmp_path / "m.gguf")
    F.write_model(path, ref, n_layer=3, n_embd=1024, n_head=8, n_head_kv=4, n_ff=1024, n_vocab=512, tag="plan")
    _, _, st = run_llama_driver(path, [1, 5, 9], 4, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda"] + (["-fa"] if fa else []),
                                env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"}, flavour="avx2", timeout=120)
    plans = [tuple(int(x) for x in m.groups()) for m in PLAN.finditer(st["stderr"])]
    decode = [p for p in plans if p[0] > 3 and p[6] == 1]
    assert len(decode) >= 3, st["stderr"][-3000:]
    for p in decode:
        nodes, launches, gemv, attn, node_eq, fused, single, cell, n_kv, graphable = p
        assert attn == 3 and gemv == 12 and graphable == 1, p
        assert launches == 5 * 3 + 1 + node_eq, p                  # + the rope table
    segs = [g for g in st["stderr"].split("ggml-mi355 plan:") if "single_token=1" in g and "rope table" in g]
    assert segs, st["stderr"][-3000:]
    body = segs[-1]
    assert body.count("rope table") == 1 and body.count("matvec + rope + KV store") == 3 and body.count("cached mask=") == 3, body[:3000]
    # and the switch back to the round-2 form
    _, _, st0 = run_llama_driver(path, [1, 5, 9], 2, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda"] + (["-fa"] if fa else []),
                                 env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1", "GGML_MI355_QKV_EPI": "0"}, flavour="avx2", timeout=120)
    assert "rope table" not in st0["stderr"] and "fused mask=" in st0["stderr"]


def test_round3_long_context_attention_lowers_to_the_matrix_core_kernel(tmp_path):
    """Beyond the split threshold (320 cells attended since round 5) the round-3 form keeps its QKV epilogue and the attention step becomes the
    matrix-core kernel over cached cells ("cached-split") - since round 4 also with --flash-attn (row-major V cache: the kernel's transposing LDS
    read); GGML_MI355_ATTN_MFMA=0 keeps the round-2 flash-decoding form for both."""
    from _bind import Ref, best_ref_flavour
    import _fixtures8d as F
    ref = Ref(best_ref_flavour())
    path = str(tmp_path / "m.gguf")
    F.write_model(path, ref, n_layer=2, n_embd=1024, n_head=8, n_head_kv=4, n_ff=1024, n_vocab=512, tag="planlc")
    prompt = [int(t) for t in F.prompt_tokens(512, 700)]
    for fa, mfma, want, never in ((False, "1", "cached-split", " split mask"), (True, "1", "cached-split", " split mask"), (True, "0", " split mask", "cached-split")):
        _, _, st = run_llama_driver(path, prompt, 3, ngl=99, n_ctx=1024, threads=1, extra_args=["--keep-out-in-cuda"] + (["-fa"] if fa else []),
                                    env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1", "GGML_MI355_ATTN_MFMA": mfma}, flavour="avx2", timeout=300)
        segs = [g for g in st["stderr"].split("ggml-mi355 plan:") if "single_token=1" in g and "attention H=8" in g]
        assert segs, st["stderr"][-3000:]
        body = segs[-1]
        assert body.count(want) == 2 and never not in body, body[:3000]


@pytest.mark.parametrize("kv", [["-ctk", "q8_0", "-ctv", "q8_0"], ["-ctk", "q8_0"]], ids=["q8_0-kv", "q8_0-k"])
def test_round5_long_context_attention_over_q8_0_caches_is_split_over_workgroups(kv, tmp_path):
    """Round 5: `-fa -ctk q8_0 [-ctv q8_0]` beyond the split threshold - the attention step of the fused plan is the split form (rope + quantizing KV
    store launch, then the matrix-core kernel over the cached Q8_0 cells: pm355_attn_token with split = 1); GGML_MI355_ATTN_MFMA=0 keeps the
    one-workgroup-per-head kernel of attn_q8.hip (its scores then have to fit LDS)."""
    from _bind import Ref, best_ref_flavour
    import _fixtures8d as F
    ref = Ref(best_ref_flavour())
    path = str(tmp_path / "m.gguf")
    F.write_model(path, ref, n_layer=2, n_embd=1024, n_head=8, n_head_kv=4, n_ff=1024, n_vocab=512, tag="planlq")
    prompt = [int(t) for t in F.prompt_tokens(512, 700)]
    for mfma, split in (("1", True), ("0", False)):
        _, _, st = run_llama_driver(path, prompt, 3, ngl=99, n_ctx=1024, threads=1, extra_args=["--keep-out-in-cuda", "-fa"] + kv,
                                    env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1", "GGML_MI355_ATTN_MFMA": mfma}, flavour="avx2", timeout=300)
        segs = [g for g in st["stderr"].split("ggml-mi355 plan:") if "single_token=1" in g and "attention H=8" in g]
        assert segs, st["stderr"][-3000:]
        body = segs[-1]
        assert (body.count(" split") == 2) == split, body[:3000]


@pytest.mark.parametrize("kv", [[], ["-ctk", "q8_0"]], ids=["f16", "k_q8_0"])
def test_k_shift_graph_is_accepted_by_the_plugin(kv, tmp_path, monkeypatch):
    """build_k_shift (src/llama.cpp:10665) after a context shift: every node lands on the plug-in's pre-allocated KV buffer, so supports_op
    must accept ROPE on the F16 cache view (and the CPY pair around an F32 ROPE for a Q8_0 cache) - the scheduler aborts otherwise
    ("pre-allocated tensor in a backend that cannot run the operation")."""
    z = np.load(os.path.join(HERE, "golden", "tiny_llama_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / "tiny_llama.gguf"), z)
    monkeypatch.setenv("REFDRV_SHIFT", "2,2,3")
    _, _, st = run_llama_driver(path, z["prompt"], 6, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda"] + kv,
                                env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"}, flavour="avx2", timeout=120)
    assert "cannot run the operation" not in st["stderr"]
    assert "ROPE 'cache_k_l0 (view)" in st["stderr"] and "ROPE 'cache_k_l1 (view)" in st["stderr"], st["stderr"][-3000:]
    if kv:
        assert st["stderr"].count("CPY 'cache_k_l0 (view) (copy") == 2


@pytest.mark.parametrize("kv", [[], ["-ctk", "q8_0"], ["-fa", "-ctk", "q8_0", "-ctv", "q8_0"]], ids=["f16", "k_q8_0", "fa_kv_q8_0"])
def test_defrag_graph_is_accepted_by_the_plugin(kv, tmp_path, monkeypatch):
    """build_defrag (src/llama.cpp:10721-10790) after holes were cut into the cache (llama_kv_cache_seq_rm) and llama_kv_cache_defrag +
    llama_kv_cache_update were called: one CPY between two views of the K cache and one of the V cache per layer and run of moved cells -
    F16 strided views, or views of native Q8_0 blocks (block-wise copy) - all on the plug-in's pre-allocated KV buffer."""
    z = np.load(os.path.join(HERE, "golden", "tiny_llama_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / "tiny_llama.gguf"), z)
    monkeypatch.setenv("REFDRV_RM", "3,2,5")
    monkeypatch.setenv("REFDRV_DEFRAG", "5")
    _, _, st = run_llama_driver(path, z["prompt"], 8, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda"] + kv,
                                env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"}, flavour="avx2", timeout=120)
    assert "cannot run the operation" not in st["stderr"]
    segs = [g for g in st["stderr"].split("ggml-mi355 plan:") if "CPY 'cache_k_l0 (view) (copy of cache_k_l0 (view))'" in g]
    assert len(segs) == 1 and segs[0].count("CPY 'cache_") == 4, st["stderr"][-3000:]


def test_reference_defrag_leaves_an_f16_cache_decode_unchanged(tmp_path, monkeypatch):
    """What the driver's REFDRV_RM / REFDRV_DEFRAG options do, on the reference's CPU backend: defragmentation moves cells, the logits of the
    following tokens do not change (F16 cache). With a Q8_0 cache the reference's own byte copy mis-sizes block rows (ggml_compute_forward_
    dup_same_cont multiplies ELEMENT counts with the 34-byte block size, ggml.c:7865-7893) and the logits move by half their range: there is no
    parity target for defragmentation of quantized caches - the plug-in moves whole blocks (cpy_q80_q80_kernel)."""
    z = np.load(os.path.join(HERE, "golden", "tiny_llama_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / "tiny_llama.gguf"), z)
    monkeypatch.setenv("REFDRV_RM", "3,2,5")
    t0, l0, _ = run_llama_driver(path, z["prompt"], 10, ngl=0, n_ctx=64, flavour="scalar")
    monkeypatch.setenv("REFDRV_DEFRAG", "5")
    t1, l1, _ = run_llama_driver(path, z["prompt"], 10, ngl=0, n_ctx=64, flavour="scalar", force=t0[:-1])
    assert np.abs(l1 - l0).max() <= 1e-4 * np.abs(l0).max()
    t2, l2, _ = run_llama_driver(path, z["prompt"], 10, ngl=0, n_ctx=64, flavour="scalar", force=t0[:-1], extra_args=["-ctk", "q8_0"])
    monkeypatch.delenv("REFDRV_DEFRAG")
    t3, l3, _ = run_llama_driver(path, z["prompt"], 10, ngl=0, n_ctx=64, flavour="scalar", force=t0[:-1], extra_args=["-ctk", "q8_0"])
    assert np.abs(l2[:6] - l3[:6]).max() == 0.0 and np.abs(l2[6:] - l3[6:]).max() > 0.05 * np.abs(l3).max()      # the reference's bug, pinned


def test_neox_rope_graphs_take_the_qkv_epilogue_where_slices_hold_both_halves_of_a_pair(tmp_path):
    """build_qwen2 rotates with NEOX pairs (i, i + n_dims / 2). Round 4: the wq | wk | wv launch gives every workgroup a slice made of two runs of
    rows n_dims / 2 apart (GemvJob::nx_s), so the epilogue rotates there too - rope table + 'matvec + rope + KV store' + cached attention, as for
    build_llama - whenever head_dim and the slice sizes are powers of two and n_rot == head_dim; a shape whose slices do not divide that way keeps
    the round-2 form (RoPE + KV store inside the attention kernel)."""
    from _bind import Ref, best_ref_flavour
    import _fixtures8d as F
    ref = Ref(best_ref_flavour())
    for n_head, n_head_kv, epi in ((8, 4, True), (6, 3, False)):          # N_q = 1024 / 768 rows over 256 workgroups: slices of 4 / 3 rows
        path = str(tmp_path / f"q{n_head}.gguf")
        F.write_model(path, ref, arch=1, n_layer=2, n_embd=128 * n_head, n_head=n_head, n_head_kv=n_head_kv, n_ff=1024, n_vocab=512, tag=f"planq{n_head}")
        _, _, st = run_llama_driver(path, [1, 5, 9], 3, ngl=99, n_ctx=64, threads=1, extra_args=["--keep-out-in-cuda"],
                                    env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1"}, flavour="avx2", timeout=120)
        plans = [tuple(int(x) for x in m.groups()) for m in PLAN.finditer(st["stderr"])]
        decode = [p for p in plans if p[0] > 3 and p[6] == 1]
        assert decode and all(p[3] == 2 and p[2] == 8 for p in decode), plans           # 2 attention + 8 fused mat-vec launches for 2 layers
        if epi:
            assert "rope table n_dims=128 mode=2" in st["stderr"] and "matvec + rope + KV store" in st["stderr"] and " cached mask=" in st["stderr"]
        else:
            assert "rope table" not in st["stderr"] and "matvec + rope + KV store" not in st["stderr"] and "fused mask=" in st["stderr"]


def test_mha_models_keep_the_round2_long_context_kernel(tmp_path):
    """More than 16 KV heads (Llama-2-7B / 13B style multi-head attention): the matrix-core long-context kernel does not serve the shape
    (pm355_attn_cached_long_check), so beyond the split threshold the planner must NOT emit the QKV-epilogue + cached-split form - the launch
    would be refused at run time - but the round-2 one (plain QKV mat-vec, rope + KV store + split attention in one launch); a GQA
    model with the same threshold does take the matrix-core form."""
    from _bind import Ref, best_ref_flavour
    import _fixtures8d as F
    ref = Ref(best_ref_flavour())
    for n_head_kv, want in ((32, "split"), (8, "cached-split")):
        path = str(tmp_path / f"mha{n_head_kv}.gguf")
        F.write_model(path, ref, arch=0, n_layer=1, n_embd=2048, n_head=32, n_head_kv=n_head_kv, n_ff=1024, n_vocab=512, tag=f"planmha{n_head_kv}")
        _, _, st = run_llama_driver(path, list(range(1, 41)), 3, ngl=99, n_ctx=128, threads=1, extra_args=["--keep-out-in-cuda"],
                                    env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_DEBUG_PLAN_STEPS": "1", "GGML_MI355_ATTN_SPLIT_MIN": "32"},
                                    flavour="avx2", timeout=180)
        plans = [tuple(int(x) for x in m.groups()) for m in PLAN.finditer(st["stderr"])]
        decode = [p for p in plans if p[0] > 3 and p[6] == 1 and p[8] >= 32]
        assert decode and all(p[3] == 1 and p[2] == 4 for p in decode), (n_head_kv, plans)
        segs = [g for g in st["stderr"].split("ggml-mi355 plan: ")[1:] if "single_token=1" in g.split("\n")[0] and " attention H=" in g]
        segs = [g for g in segs if int(re.search(r"n_kv=(\d+)", g).group(1)) >= 32]
        assert segs, st["stderr"][-2000:]
        for g in segs:
            assert f"Hkv={n_head_kv} dh=64 n_ctx=128 {want} " in g, (n_head_kv, g[-1500:])
            assert ("matvec + rope + KV store" in g) == (want == "cached-split"), g[-1500:]


def test_graph_reuse_patch_with_two_devices_never_reuses_under_a_pipeline_parallel_scheduler(tmp_path):
    """with more than one device libllama MAY build its scheduler with n_copies > 1 (src/llama.cpp:21328-21355) - split_graph then binds the
    graph inputs to copy `cur_copy`, which advances after every compute, so a KEPT graph would read the copy its inputs were not uploaded into. The
    patch's guard asks `ggml_backend_sched_get_n_copies == 1` of every scheduler. Two pretend devices (GGML_MI355_PLAN_DEVICES=2), LLAMA_MI355_GRAPH_REUSE=1:
    whenever the run reports pipeline parallelism, no single-token decode may have reused a graph. (prima.cpp's loader clamps n_gpu_layers to its layer
    window, :21330 `n_gpu_layers > n_layer` - on this reference the second device stays idle and the scheduler single-copy; the guard is what keeps the patch
    safe on an upstream libllama.) With one copy the reuse must still work with both devices registered."""
    z = np.load(os.path.join(HERE, "golden", "tiny_llama_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / "tiny_llama.gguf"), z)
    pat = re.compile(r"graph-reuse patch\): (\d+) of (\d+) single-token decodes reused")
    for ndev in (1, 2):
        _, _, st = run_llama_driver(path, z["prompt"][:3], 6, ngl=99, n_ctx=64, threads=1, flavour="avx2", timeout=120, extra_args=["--keep-out-in-cuda"],
                                    env={"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_PLAN_DEVICES": str(ndev), "LLAMA_MI355_GRAPH_REUSE": "1"})
        m = pat.search(st["stderr"])
        reused = int(m.group(1)) if m else 0
        if ndev == 2:
            assert "MI355X1" in st["stderr"], st["stderr"][-1500:]
        if "pipeline parallelism enabled" in st["stderr"]:
            assert reused == 0, st["stderr"][-800:]
        else:
            assert m and reused >= 3, st["stderr"][-800:]


def test_graph_reuse_patch_tells_the_plugin_and_the_plan_is_found_by_graph_pointer(tmp_path):
    """with the graph-reuse patch on, libllama resolves `ggml_backend_mi355_set_graph_reused` through the registry's get_proc_address and
    tells the plug-in before every compute whether the ggml_cgraph objects are last token's; the plug-in then finds its plan by graph pointer + node count and
    skips the fingerprint walk over the nodes (45-84 us per token on an 80-layer graph). Without the patch every graph is fingerprinted."""
    z = np.load(os.path.join(HERE, "golden", "tiny_llama_decode.npz"))
    path = write_gguf_from_arrays(str(tmp_path / "tiny_llama.gguf"), z)
    pat = re.compile(r"fingerprint hits (\d+), graph-pointer hits (\d+)")
    hits = {}
    for reuse in ("0", "1"):
        env = {"GGML_MI355_PLAN_ONLY": "1", "GGML_MI355_STATS": "1"}
        if reuse == "1":
            env["LLAMA_MI355_GRAPH_REUSE"] = "1"
        _, _, st = run_llama_driver(path, z["prompt"][:3], 8, ngl=99, n_ctx=64, threads=1, flavour="avx2", timeout=120, extra_args=["--keep-out-in-cuda"], env=env)
        m = pat.search(st["stderr"])
        assert m, st["stderr"][-1500:]
        hits[reuse] = (int(m.group(1)), int(m.group(2)))
    assert hits["0"][1] == 0 and hits["0"][0] >= 6, hits
    # two graphs per token (layers, head): at least the reused tokens' graphs are found by pointer
    assert hits["1"][1] >= 8 and hits["1"][0] < hits["0"][0], hits
