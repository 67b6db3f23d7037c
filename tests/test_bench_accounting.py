"""The byte accounting behind bench.py's roofline numbers must be the one SURVEY.md 8(d) states: weights read once per
token with the Q4_K_M type mixture of llama_tensor_get_type (src/llama.cpp:19271-19490). No GPU, no library needed."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_per_token_match_survey():
    b = _bench()
    from prima_cpp_amd.lib import Q6_K, row_size
    want = {"llama3-70b": (41.922e9, 41.060e9), "llama3-8b": (4.617e9, 4.186e9)}
    for name, (total, layers) in want.items():
        hp, mixture, _ = b.model_cfg(name)
        lb = b.layer_bytes(hp, mixture)
        head = row_size(Q6_K, hp["n_embd"]) * hp["n_vocab"] + hp["n_embd"] * 4
        assert abs(sum(lb) - layers) / layers < 5e-4, (name, sum(lb))
        assert abs(sum(lb) + head - total) / total < 5e-4, (name, sum(lb) + head)
    # the dominant launch: Q4_K gate + up of one 70B layer
    hp, _, _ = b.model_cfg("llama3-70b")
    from prima_cpp_amd.lib import Q4_K
    assert 2 * row_size(Q4_K, hp["n_embd"]) * hp["n_ff"] == 264241152


def test_q4_k_m_mixture_follows_llama_tensor_get_type():
    """attn_v / ffn_down are 'more bits' on layers i < L/8, i >= 7L/8 and (i - L/8) % 3 == 2 (use_more_bits, src/llama.cpp:19300);
    the 70B shape uses Q5_K for the remaining attn_v (n_gqa >= 4 rule, :19382-19389)."""
    b = _bench()
    import prima_cpp_amd.engine as E
    from prima_cpp_amd.lib import Q4_K, Q5_K, Q6_K
    hp, mixture, _ = b.model_cfg("llama3-70b")
    L = hp["n_layer"]
    more = [i for i in range(L) if i < L // 8 or i >= 7 * L // 8 or (i - L // 8) % 3 == 2]
    assert len(more) == 40
    for i in range(L):
        t = mixture(hp, i)
        assert t[E.T_FFN_DOWN] == (Q6_K if i in more else Q4_K)
        assert t[E.T_WV] == (Q6_K if i in more else Q5_K)
        assert t[E.T_WQ] == t[E.T_WK] == t[E.T_WO] == t[E.T_FFN_GATE] == t[E.T_FFN_UP] == Q4_K
