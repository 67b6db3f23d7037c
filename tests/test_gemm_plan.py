"""Host logic of the prompt GEMM's launch geometry (prima_cpp_amd/csrc/mmq_pf.hip pm_gemm_pf_plan, through the C ABI - no GPU, no kernel launch): tile size by
batch, the uniform K split of launches with few tiles, and the tail-only split of launches that are full rounds of workgroups + a short tail. The shapes are
the layer shapes of BASELINE.json's models on a 256-CU device in 8 XCDs; every workgroup id of the planned grid must map to exactly one (tile, K slice)."""
import ctypes as C

import pytest

from prima_cpp_amd import lib as L

CUS = 256


def plan(tiles, K, T, mixed=1, cus=CUS):
    lib = L.load()
    lib.pm355_gemm_plan.restype = C.c_int
    lib.pm355_gemm_plan.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
    out = (C.c_int32 * 5)()
    assert lib.pm355_gemm_plan(tiles, K, T, cus, mixed, out) == 0
    return dict(tile_tokens=out[0], nt_t=out[1], S=out[2], full=out[3], grid=out[4])


def coverage(tiles, p):
    """The kernel's block id -> (row tile, token tile, K slice) mapping (gemm_pf_kernel), restated: returns {(tile_n, tile_t): sorted slices}."""
    seen = {}
    for bid in range(p["grid"]):
        x, q = bid & 7, bid >> 3
        if q < p["full"]:
            S, ks, slot = 1, 0, q
        else:
            S, ks, slot = p["S"], (q - p["full"]) % p["S"], p["full"] + (q - p["full"]) // p["S"]
        per_x = (tiles + 7 - x) >> 3
        if slot >= per_x * p["nt_t"]:
            continue
        key = (x + 8 * (slot // p["nt_t"]), slot % p["nt_t"])
        seen.setdefault(key, []).append((ks, S))
    return seen


@pytest.mark.parametrize("tiles,K,T", [(40, 8192, 2048), (231, 8192, 2048), (112, 8192, 2048), (224, 8192, 2048), (32, 8192, 2048), (32, 28672, 2048),
                                       (40, 8192, 512), (32, 8192, 512), (224, 8192, 512), (32, 28672, 512), (112, 8192, 64), (32, 8192, 40), (5, 1024, 300),
                                       (33, 4096, 2048), (17, 2048, 4096), (1, 512, 1)])
def test_every_tile_and_slice_is_covered_exactly_once(tiles, K, T):
    for mixed in (1, 0):
        p = plan(tiles, K, T, mixed)
        assert p["tile_tokens"] == (64 if T <= 64 else 128 if T <= 128 else 256)
        assert p["nt_t"] == -(-T // p["tile_tokens"])
        cov = coverage(tiles, p)
        assert len(cov) == tiles * p["nt_t"]
        for key, sl in cov.items():
            S = sl[0][1]
            assert sorted(sl) == [(k, S) for k in range(S)], (key, sl)
        if not mixed:
            assert p["full"] == 0
        if p["S"] > 1:
            assert (K + 255) // 256 // p["S"] >= 4                  # at least four super-blocks per slice


def test_layer_shapes_of_the_baseline_models():
    # Llama-3-70B, 2048-token prompt: wq | wk | wv = 40 row tiles x 8 token tiles = 320 workgroups on 256 CUs: one full round of whole tiles (32 slots per
    # XCD), the 64 tiles behind them in four K slices each (8 slots x 4 = the XCD's 32 CUs)
    assert plan(40, 8192, 2048) == dict(tile_tokens=256, nt_t=8, S=4, full=32, grid=8 * (32 + 8 * 4))
    # ... the uniform split it replaces: three slices of every tile
    assert plan(40, 8192, 2048, mixed=0)["S"] == 3 and plan(40, 8192, 2048, mixed=0)["full"] == 0
    # ffn_gate | ffn_up as pair tiles: 224 x 8 = exactly 7 rounds, nothing to split; wo / ffn_down: exactly one round
    assert plan(224, 8192, 2048) == dict(tile_tokens=256, nt_t=8, S=1, full=0, grid=8 * 224)
    assert plan(32, 8192, 2048)["S"] == 1 and plan(32, 28672, 2048)["S"] == 1
    # Qwen2.5-72B's pair launch: 231 tiles x 8 = 7 rounds + 56 tiles: XCDs 0..6 own 29 row tiles (232 slots), XCD 7 owns 28 - 224 whole-tile slots, the rest x 4
    assert plan(231, 8192, 2048) == dict(tile_tokens=256, nt_t=8, S=4, full=224, grid=8 * (224 + 8 * 4))
    # the reference's default micro-batch (n_ubatch 512, common/common.h:178): 64-80 tiles for 256 CUs - every tile split
    assert plan(32, 8192, 512)["S"] == 4 and plan(32, 8192, 512)["full"] == 0
    assert plan(32, 28672, 512)["S"] == 4
    assert plan(40, 8192, 512)["S"] == 3
    # 33..64-token batches: 64-token tiles, one token tile
    p = plan(32, 8192, 64)
    assert p["tile_tokens"] == 64 and p["nt_t"] == 1 and p["S"] == 8


def test_bad_arguments_are_refused():
    lib = L.load()
    lib.pm355_gemm_plan.restype = C.c_int
    lib.pm355_gemm_plan.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
    out = (C.c_int32 * 5)()
    assert lib.pm355_gemm_plan(0, 8192, 512, CUS, 1, out) != 0
    assert lib.pm355_gemm_plan(4, 8192, 512, CUS, 1, None) != 0
    assert lib.pm355_gemm_plan(4, 128, 512, CUS, 1, out) != 0
