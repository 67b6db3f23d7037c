"""GPU parity tests (run with -m gpu on an MI355X): every op goes through the C ABI of
libprima_mi355.so and is compared with the CPU oracle on the same seeded inputs.
Integer / byte results: bit-exact. Float results: tolerance stated per test."""
import numpy as np
import pytest

from _bind import Q4_K, Q5_K, Q6_K, Q8_0, Q8_K, QUANT_TYPES, TYPE_NAMES, rand_blocks, row_size, vec_dot_type

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("GPU tests need a HIP device")
    import prima_cpp_amd.ops as ops
    ops.torch = torch
    return ops


def _dev(P, a):
    return P.torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def _acts(rng, k, kind):
    if kind == "normal":
        return rng.normal(0, 1, k).astype(np.float32)
    if kind == "cos":
        return (0.1 + 2 * np.cos(np.arange(k, dtype=np.float32))).astype(np.float32)
    if kind == "zeros":
        return np.zeros(k, dtype=np.float32)
    if kind == "ties":
        x = rng.normal(0, 0.1, k).astype(np.float32)
        x[5::256] = -3.0
        x[9::256] = 3.0
        if k > 300:
            x[300] = 3.0      # second block: positive one comes first
            x[261] = 0.0
        return x
    if kind == "halfway":
        return ((np.arange(k) % 255 - 127).astype(np.float32) * 0.5).astype(np.float32)
    if kind == "sparse":
        x = np.zeros(k, dtype=np.float32)
        x[::97] = rng.normal(0, 5, x[::97].size)
        return x
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["normal", "cos", "zeros", "ties", "halfway", "sparse"])
@pytest.mark.parametrize("K", [256, 2048, 8192])
def test_quantize_q8_K_bitexact(P, oracle, kind, K):
    rng = np.random.default_rng(21)
    x = np.stack([_acts(rng, K, kind), _acts(rng, K, "normal"), _acts(rng, K, kind)])
    yq = P.quantize_act(_dev(P, x), Q8_K)
    got = P.act_to_ggml_blocks(yq, Q8_K, K, 3).reshape(3, -1)
    for r in range(3):
        want = oracle.quantize_row_q8_K(x[r])
        assert np.array_equal(got[r], want), (kind, K, r)


@pytest.mark.parametrize("kind", ["normal", "cos", "zeros", "halfway", "sparse"])
def test_quantize_q8_0_bitexact(P, oracle, kind):
    rng = np.random.default_rng(22)
    K = 4096
    x = np.stack([_acts(rng, K, kind), _acts(rng, K, "normal")])
    yq = P.quantize_act(_dev(P, x), Q8_0)
    got = P.act_to_ggml_blocks(yq, Q8_0, K, 2).reshape(2, -1)
    for r in range(2):
        assert np.array_equal(got[r], oracle.quantize_row_q8_0(x[r]))


@pytest.mark.parametrize("t", QUANT_TYPES)
@pytest.mark.parametrize("K", [2048, 768])
def test_repack_roundtrip_bitexact(P, t, K):
    rng = np.random.default_rng(23)
    N = 19
    blocks = rand_blocks(t, N, K, rng)
    w = P.upload_weight(t, blocks, K, N)
    assert np.array_equal(P.download_weight(w), blocks)
    if t in (Q4_K, Q6_K, Q8_0):  # and the HBM image is the documented row-SoA permutation (rows 16-B aligned):
        # one stream per 16-byte piece a lane loads (repack.hip)
        stride = P.L.load().pm355_row_stride(t, K)
        assert stride % 16 == 0 and 0 <= stride - row_size(t, K) < 16
        img = w.data.cpu().numpy().reshape(N, stride)[:, :row_size(t, K)]
        src = blocks.reshape(N, -1)
        if t == Q6_K:
            nb = K // 256
            b = src.reshape(N, nb, 210)
            ql = b[:, :, :128].reshape(N, nb, 2, 2, 2, 16)          # [block][half hh][second][v][16]
            la = ql[:, :, :, 0].reshape(N, -1)                       # unit order (b, hh, v)
            lb = ql[:, :, :, 1].reshape(N, -1)
            if P.L.load().pm355_q6k_tail_grouped():                  # round 5: per group of <= 8 blocks scales[r][16] | d[r]
                tail = [np.concatenate([b[:, g0:g0 + 8, 192:208].reshape(N, -1), b[:, g0:g0 + 8, 208:].reshape(N, -1)], axis=1) for g0 in range(0, nb, 8)]
            else:
                tail = [b[:, :, 192:208].reshape(N, -1), b[:, :, 208:].reshape(N, -1)]
            want = np.concatenate([la, lb, b[:, :, 128:192].reshape(N, -1)] + tail, axis=1)
        elif t == Q4_K:
            nb = K // 256
            b = src.reshape(N, nb, 144)
            qs = b[:, :, 16:].reshape(N, nb, 4, 2, 16)               # [block][j][second][16]
            want = np.concatenate([qs[:, :, :, 0].reshape(N, -1), qs[:, :, :, 1].reshape(N, -1),
                                   b[:, :, :16].reshape(N, -1)], axis=1)
        else:
            nb = K // 32
            b = src.reshape(N, nb, 34)
            qs = b[:, :, 2:].reshape(N, nb, 2, 16)
            want = np.concatenate([qs[:, :, 0].reshape(N, -1), qs[:, :, 1].reshape(N, -1), b[:, :, :2].reshape(N, -1)], axis=1)
        assert np.array_equal(img, want)


@pytest.mark.parametrize("threads", [0, 3])
def test_async_uploader_matches_synchronous_set_tensor(P, threads):
    """upload.hip (pinned ring + copier threads + private stream + repack per chunk) leaves exactly the HBM image of the synchronous
    H2D + repack path, for chunks smaller than a tensor (rows split over many chunks), ragged tails, and plain byte tensors."""
    import ctypes as C
    import torch
    lib = P.L.load()
    rng = np.random.default_rng(231)
    up = lib.pm355_uploader_new(1 << 20, threads)                      # 1-MiB chunks: every tensor below spans several
    assert up
    try:
        total = 0
        for t, K, N in ((Q4_K, 4096, 1500), (Q6_K, 2304, 1111), (Q8_0, 1056, 4099), (Q5_K, 2048, 999)):
            blocks = rand_blocks(t, N, K, rng)
            want = P.upload_weight(t, blocks, K, N)
            host = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1)
            dst = torch.zeros_like(want.data)
            assert lib.pm355_upload(up, t, K, host.ctypes.data, dst.data_ptr(), host.size, 1) == 0
            host[:] = 0                                               # the source may be reused as soon as the call returns
            assert lib.pm355_uploader_sync(up) == 0
            assert torch.equal(dst, want.data)
            total += host.size
        raw = rng.integers(0, 256, (3 << 20) + 12345, dtype=np.uint8)
        dst = torch.zeros(raw.size, dtype=torch.uint8, device="cuda")
        assert lib.pm355_upload(up, -1, 0, raw.ctypes.data, dst.data_ptr(), raw.size, 0) == 0
        assert lib.pm355_uploader_sync(up) == 0
        assert np.array_equal(dst.cpu().numpy(), raw)
        assert lib.pm355_uploader_bytes(up) == total + raw.size
        # a row that does not fit a chunk is refused, never truncated
        big = rand_blocks(Q4_K, 1, 2 << 20, rng).reshape(-1)
        d2 = torch.zeros(big.size, dtype=torch.uint8, device="cuda")
        assert lib.pm355_upload(up, Q4_K, 2 << 20, big.ctypes.data, d2.data_ptr(), big.size, 1) != 0
    finally:
        lib.pm355_uploader_free(up)


@pytest.mark.parametrize("t", QUANT_TYPES)
@pytest.mark.parametrize("K,N", [(256, 3), (768, 5), (2048, 5), (4096, 37), (5120, 4), (8192, 64), (14336, 9), (28672, 6)])
def test_gemv_integer_partials_bitexact_and_float_close(P, oracle, t, K, N):
    rng = np.random.default_rng(24)
    blocks = rand_blocks(t, N, K, rng)
    x = rng.normal(0, 1, K).astype(np.float32)
    w = P.upload_weight(t, blocks, K, N)
    xq = P.quantize_act(_dev(P, x[None]), vec_dot_type(t))
    y, ip = P.mul_mat_vec_dbg(w, xq)
    y, ip = y.cpu().numpy(), ip.cpu().numpy().astype(np.int64)
    a = oracle.quantize_act(t, x)
    rs = row_size(t, K)
    upb = ip.shape[1] // (K // (32 if t == Q8_0 else 256))      # units per activation block
    for r in range(N):
        wr = blocks[r * rs:(r + 1) * rs]
        isum, msum = oracle.int_partials(t, K, wr, a)
        got_i = ip[r, :, 0].reshape(-1, upb).sum(1)
        got_m = ip[r, :, 1].reshape(-1, upb).sum(1)
        assert np.array_equal(got_i, isum.astype(np.int64)), (TYPE_NAMES[t], K, r)
        assert np.array_equal(got_m, msum.astype(np.int64)), (TYPE_NAMES[t], K, r)
        want = oracle.vec_dot(t, K, wr, a)
        mag = float(np.abs(oracle.dequantize_row(t, wr, K)) @ np.abs(x))
        # identical integer terms; only the f32 summation order differs (K/256 <= 112 terms)
        assert abs(y[r] - want) <= 4e-6 * mag + 1e-30, (TYPE_NAMES[t], K, r, y[r], want)


@pytest.mark.parametrize("C", [2, 3, 4, 5, 7, 8])       # column slots per launch group (mmvq_cols.hip): 2, 4 (one idle), 4, 4+1, 4+4 (one idle), 8; pairs: 2 per launch
@pytest.mark.parametrize("t", QUANT_TYPES)
def test_gemv_epilogues_and_columns(P, oracle, t, C):
    rng = np.random.default_rng(25)
    K, N = 4096, 70
    b1, b2 = rand_blocks(t, N, K, rng), rand_blocks(t, N, K, rng)
    x = rng.normal(0, 1, (C, K)).astype(np.float32)
    bias = rng.normal(0, 1, N).astype(np.float32)
    resid = rng.normal(0, 1, (C, N)).astype(np.float32)
    w1, w2 = P.upload_weight(t, b1, K, N), P.upload_weight(t, b2, K, N)
    want1 = oracle.mul_mat(t, b1, K, N, x)
    want2 = oracle.mul_mat(t, b2, K, N, x)
    tol = dict(rtol=2e-5, atol=2e-5)
    y = P.mul_mat_vec(w1, x=_dev(P, x)).cpu().numpy()
    assert np.allclose(y, want1, **tol)
    y = P.mul_mat_vec(w1, x=_dev(P, x), bias=_dev(P, bias), resid=_dev(P, resid)).cpu().numpy()
    assert np.allclose(y, want1 + bias[None] + resid, **tol)
    y = P.mul_mat_vec(w1, x=_dev(P, x), w2=w2).cpu().numpy()
    assert np.allclose(y, oracle.silu_mul(want1, want2), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("T", [1, 3, 8, 9, 16, 17, 33, 64])
def test_small_batch_matmul_q8_0_blocks(P, oracle, T):
    """mmq_i8.hip, Q8_0 weights x Q8_0 activations (ggml_vec_dot_q8_0_q8_0: integer block sums times d_w * d_a, f32): one 32-k matrix instruction per
    block. Shapes: K not a multiple of 128 or 256 (Qwen2.5-72B's ffn_down fallback, K = 29568 = 924 blocks: the last step holds 4 of 8 blocks), ragged rows,
    several row groups; passes of 16 tokens over 32-slot tables (T = 17, 33: second half / second table)."""
    rng = np.random.default_rng(4000 + T)
    for K, N in ((4096, 70), (608, 600), (1184, 9000), (29568, 40)):
        if (N == 9000 or K == 29568) and T not in (3, 17, 64): continue
        b = rand_blocks(Q8_0, N, K, rng)
        x = rng.normal(0, 1, (T, K)).astype(np.float32)
        bias = rng.normal(0, 1, N).astype(np.float32)
        resid = rng.normal(0, 1, (T, N)).astype(np.float32)
        w = P.upload_weight(Q8_0, b, K, N)
        want = oracle.mul_mat(Q8_0, b, K, N, x)
        tol = dict(rtol=2e-5, atol=2e-5 * np.sqrt(K / 4096))
        y = P.mul_mat_small(w, x=_dev(P, x)).cpu().numpy()
        assert np.isfinite(y).all()
        assert np.allclose(y, want, **tol), (K, N, np.abs(y - want).max())
        xq = P.quantize_act(_dev(P, x), P.vec_dot_act_type(Q8_0))
        y2 = P.mul_mat_small(w, xq=xq, n_tokens=T, bias=_dev(P, bias), resid=_dev(P, resid)).cpu().numpy()
        assert np.allclose(y2, want + bias[None] + resid, **tol)
        if T <= 8:
            y1 = P.mul_mat_vec(w, xq=xq, ncols=T).cpu().numpy()
            assert np.allclose(y, y1, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("T", [1, 2, 5, 8, 16, 17, 32, 33, 64])
@pytest.mark.parametrize("t", [Q4_K, Q5_K, Q6_K])
def test_small_batch_matmul_on_integer_matrix_cores(P, oracle, t, T):
    """mmq_i8.hip: same integer block sums as the mat-vec (vec_dot_q4_K_q8_K / vec_dot_q6_K_q8_K), f32 super-block terms added in a
    different order -> oracle.mul_mat within the mat-vec's tolerance. Shapes: ragged row slices (N = 70: clamped rows), several row
    groups per workgroup (N = 600 on <= 256 workgroups is 1 group; N = 20000 gives 3), odd super-block count (K = 768)."""
    rng = np.random.default_rng(100 * t + T)
    for K, N in ((4096, 70), (768, 600), (1024, 20000)):
        if N == 20000 and T not in (5, 32, 64): continue
        b = rand_blocks(t, N, K, rng)
        x = rng.normal(0, 1, (T, K)).astype(np.float32)
        bias = rng.normal(0, 1, N).astype(np.float32)
        resid = rng.normal(0, 1, (T, N)).astype(np.float32)
        w = P.upload_weight(t, b, K, N)
        want = oracle.mul_mat(t, b, K, N, x)
        tol = dict(rtol=2e-5, atol=2e-5 * np.sqrt(K / 4096))
        y = P.mul_mat_small(w, x=_dev(P, x)).cpu().numpy()
        assert np.allclose(y, want, **tol), (K, N, np.abs(y - want).max())
        xq = P.quantize_act(_dev(P, x), P.vec_dot_act_type(t))
        y2 = P.mul_mat_small(w, xq=xq, n_tokens=T, bias=_dev(P, bias), resid=_dev(P, resid)).cpu().numpy()
        assert np.allclose(y2, want + bias[None] + resid, **tol)
        # against the mat-vec itself: identical integers, f32 sums of <= K/256 terms in another order
        if T <= 8:
            y1 = P.mul_mat_vec(w, xq=xq, ncols=T).cpu().numpy()
            assert np.allclose(y, y1, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("t", [Q4_K, Q6_K])
def test_prompt_matmul_on_integer_matrix_cores(P, oracle, t):
    """mmq_big.hip: prompt-sized batches with the CPU reference's integer arithmetic (Q8_K activations, exact int32 sub-block sums, one f32
    multiply-add per super-block, super-blocks in k order) -> oracle.mul_mat within f32 summation order, where the F16 GEMM sits at 1e-3.
    Shapes: ragged token and row tiles (T = 130 / 200, N = 70 / 300: clamped rows and tokens, tables of a partial 32-token pass), several
    tiles per XCD (N = 4100), many super-blocks (K = 8192), bias + residual epilogue; and the mat-vec's own result for the same rows."""
    rng = np.random.default_rng(900 + t)
    for K, N, T in ((1024, 70, 130), (2048, 300, 200), (8192, 520, 128), (1024, 4100, 257)):
        b = rand_blocks(t, N, K, rng)
        x = rng.normal(0, 1, (T, K)).astype(np.float32)
        bias = rng.normal(0, 1, N).astype(np.float32)
        resid = rng.normal(0, 1, (T, N)).astype(np.float32)
        w = P.upload_weight(t, b, K, N)
        want = oracle.mul_mat(t, b, K, N, x)
        tol = dict(rtol=2e-5, atol=2e-5 * np.sqrt(K / 4096))
        y = P.mul_mat_i8(w, _dev(P, x)).cpu().numpy()
        assert np.isfinite(y).all()
        assert np.allclose(y, want, **tol), (K, N, T, np.abs(y - want).max(), np.argwhere(~np.isclose(y, want, **tol))[:8])
        y2 = P.mul_mat_i8(w, _dev(P, x), bias=_dev(P, bias), resid=_dev(P, resid)).cpu().numpy()
        assert np.allclose(y2, want + bias[None] + resid, **tol)
        xq = P.quantize_act(_dev(P, x[:4]), P.vec_dot_act_type(t))
        y1 = P.mul_mat_vec(w, xq=xq, ncols=4).cpu().numpy()
        assert np.allclose(y[:4], y1, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("t", [Q4_K, Q6_K])
def test_small_batch_matmul_multi_job_launch(P, oracle, t):
    """wq | wk | wv and ffn_gate | ffn_up as ONE launch over a virtual row space: every job equals its own single-job launch (same integers; the
    K split over waves may differ) and the oracle; row groups straddle the job boundaries (N not a multiple of 32)."""
    rng = np.random.default_rng(300 + t)
    for K, Ns, T in ((1024, (300, 70, 75), 5), (768, (2100, 2100), 16), (4096, (515, 40), 9), (1024, (300, 70, 75), 20), (2048, (2100, 2100), 32), (1024, (300, 70), 45), (768, (600, 600, 90), 64)):
        blocks = [rand_blocks(t, N, K, rng) for N in Ns]
        ws = [P.upload_weight(t, b, K, N) for b, N in zip(blocks, Ns)]
        x = rng.normal(0, 1, (T, K)).astype(np.float32)
        biases = [rng.normal(0, 1, N).astype(np.float32) for N in Ns]
        xq = P.quantize_act(_dev(P, x), P.vec_dot_act_type(t))
        ys = P.mul_mat_small_multi(ws, xq, T, biases=[_dev(P, b) for b in biases])
        for w, b, bias, y in zip(ws, blocks, biases, ys):
            want = oracle.mul_mat(t, b, K, w.N, x) + bias[None]
            got = y.cpu().numpy()
            assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * np.sqrt(K / 4096)), (K, w.N, np.abs(got - want).max())
            one = P.mul_mat_small(w, xq=xq, n_tokens=T, bias=_dev(P, bias)).cpu().numpy()
            assert np.allclose(got, one, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tv", [Q6_K, Q5_K])
def test_small_batch_matmul_wv_of_another_type_in_the_same_grid(P, oracle, tv):
    """wq | wk (Q4_K) and wv (Q6_K / Q5_K: the Q4_K_M files' use_more_bits layers, src/llama.cpp:19447) as ONE grid - the device's workgroups divided by
    weight bytes, every part with its own row split: each job equals its own single launch (same integers; the K split over waves may differ) and the oracle."""
    rng = np.random.default_rng(330 + tv)
    for K, Ns, T in ((1024, (300, 70, 75), 5), (8192, (1024, 128, 128), 8), (2048, (2100, 260, 300), 16), (4096, (515, 40, 33), 9)) + (((1024, (600, 64, 90), 32), (8192, (520, 130, 100), 24)) if tv == Q6_K else ()):
        types = (Q4_K, Q4_K, tv)
        blocks = [rand_blocks(t, N, K, rng) for t, N in zip(types, Ns)]
        ws = [P.upload_weight(t, b, K, N) for t, b, N in zip(types, blocks, Ns)]
        x = rng.normal(0, 1, (T, K)).astype(np.float32)
        biases = [rng.normal(0, 1, N).astype(np.float32) for N in Ns]
        xq = P.quantize_act(_dev(P, x), P.vec_dot_act_type(Q4_K))
        ys = P.mul_mat_small_mixed(ws, xq, T, biases=[_dev(P, b) for b in biases])
        for w, t, b, bias, y in zip(ws, types, blocks, biases, ys):
            want = oracle.mul_mat(t, b, K, w.N, x) + bias[None]
            got = y.cpu().numpy()
            assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * np.sqrt(K / 4096)), (K, w.N, np.abs(got - want).max())
            one = P.mul_mat_small(w, xq=xq, n_tokens=T, bias=_dev(P, bias)).cpu().numpy()
            assert np.allclose(got, one, rtol=1e-5, atol=1e-5)
    # what the single grid does not serve is refused, not approximated
    ws = [P.upload_weight(t, rand_blocks(t, 64, 1024, rng), 1024, 64) for t in (Q4_K, Q4_K, tv)]
    xq = P.quantize_act(_dev(P, rng.normal(0, 1, (40, 1024)).astype(np.float32)), P.vec_dot_act_type(Q4_K))
    with pytest.raises(Exception):
        P.mul_mat_small_mixed(ws, xq, 40)


def test_rms_norm_matches_oracle(P, oracle):
    rng = np.random.default_rng(26)
    for K, eps in ((4096, 1e-5), (8192, 1e-6)):
        x = rng.normal(0, 2, (3, K)).astype(np.float32)
        w = (1 + rng.normal(0, 0.02, K)).astype(np.float32)
        y, yq = P.rms_norm(_dev(P, x), _dev(P, w), eps, want_f32=True, want_q8=True)
        want = oracle.rms_norm(x, w, eps)
        got = y.cpu().numpy()
        # f64 sum of f32 squares in a different order: the f32 mean is the same except on rare rounding
        # boundaries -> allow 1 ulp
        ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)).max()
        assert ulp <= 1, ulp
        blocks = P.act_to_ggml_blocks(yq, Q8_K, K, 3).reshape(3, -1)
        for r in range(3):
            assert np.array_equal(blocks[r], oracle.quantize_row_q8_K(got[r]))
        y2 = P.rms_norm(_dev(P, x), None, eps).cpu().numpy()
        assert np.abs(y2.view(np.int32).astype(np.int64) - oracle.rms_norm(x, None, eps).view(np.int32).astype(np.int64)).max() <= 1


def test_gemv_full_size_linearity(P):
    """BASELINE-size property test (no oracle at this size): W.(a) + W.(b) == W.(a+b) is NOT exact after
    activation quantization, so instead check y(W, x) against dequantize-free structure: scaling x by 2
    (exact in fp) must scale y by exactly 2, and permuting rows of W permutes y."""
    torch = P.torch
    rng = np.random.default_rng(27)
    K, N = 8192, 8192
    blocks = rand_blocks(Q4_K, N, K, rng)
    w = P.upload_weight(Q4_K, blocks, K, N)
    x = torch.from_numpy(rng.normal(0, 1, (1, K)).astype(np.float32)).cuda()
    y1 = P.mul_mat_vec(w, x=x)
    y2 = P.mul_mat_vec(w, x=2 * x)
    assert torch.equal(y2, 2 * y1)
    perm = torch.from_numpy(rng.permutation(N)).cuda()
    wp = P.QWeight(Q4_K, K, N, w.data.view(N, -1)[perm].contiguous().view(-1))
    assert torch.equal(P.mul_mat_vec(wp, x=x), y1[:, perm])
    assert torch.isfinite(y1).all()


def _f16bits(P, a):
    return P.torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).cuda()


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("with_ff", [False, True])
def test_rope_kv_store_vs_oracle(P, oracle, mode, with_ff):
    torch = P.torch
    rng = np.random.default_rng(41)
    T, H, Hkv, dh, n_ctx, pos0 = 3, 8, 2, 128, 64, 37
    q = rng.normal(0, 1, (T, H * dh)).astype(np.float32)
    k = rng.normal(0, 1, (T, Hkv * dh)).astype(np.float32)
    v = rng.normal(0, 1, (T, Hkv * dh)).astype(np.float32)
    ff = (1 + rng.uniform(0, 7, dh // 2)).astype(np.float32) if with_ff else None
    kc = torch.zeros(n_ctx * Hkv * dh, dtype=torch.int16, device="cuda")
    vc = torch.zeros(n_ctx * Hkv * dh, dtype=torch.int16, device="cuda")
    qr, kr = P.rope_kv_store(_dev(P, q), _dev(P, k), _dev(P, v), kc, vc, pos0, H, Hkv, dh, n_ctx,
                             freq_factors=None if ff is None else _dev(P, ff), mode=mode, freq_base=500000.0)
    pos = np.arange(pos0, pos0 + T, dtype=np.int32)
    q_ref = oracle.rope(q.reshape(T, H, dh), pos, freq_factors=ff, mode=mode, freq_base=500000.0).reshape(T, -1)
    k_ref = oracle.rope(k.reshape(T, Hkv, dh), pos, freq_factors=ff, mode=mode, freq_base=500000.0).reshape(T, -1)
    # identical f32 op sequence except device cosf/sinf (<= 2 ulp): |err| <= 4 ulp of the operand magnitude
    for got, want, src in ((qr.cpu().numpy(), q_ref, q), (kr.cpu().numpy(), k_ref, k)):
        assert np.abs(got - want).max() <= 1e-6 * np.abs(src).max() * 4
    # KV cache contents: K rows = f16(rope(k)) at positions pos0.., V transposed = f16(v); everything else zero
    kc_h = kc.cpu().numpy().view(np.uint16).reshape(n_ctx, Hkv * dh)
    vc_h = vc.cpu().numpy().view(np.uint16).reshape(Hkv * dh, n_ctx)
    assert np.array_equal(kc_h[pos0:pos0 + T], kr.cpu().numpy().astype(np.float16).view(np.uint16))
    assert np.array_equal(vc_h[:, pos0:pos0 + T], v.T.astype(np.float16).view(np.uint16))
    assert not kc_h[:pos0].any() and not kc_h[pos0 + T:].any() and not vc_h[:, :pos0].any() and not vc_h[:, pos0 + T:].any()


@pytest.mark.parametrize("n_past,T", [(0, 1), (5, 1), (63, 1), (200, 3), (0, 4)])
def test_attn_decode_vs_oracle(P, oracle, n_past, T):
    """Same rounding points as the reference (q and p rounded to F16, F16 K/V): compare with an exact numpy
    restatement of llm_build_kqv in float64 on the SAME f16-rounded inputs."""
    torch = P.torch
    rng = np.random.default_rng(42)
    H, Hkv, dh, n_ctx = 8, 2, 128, 256
    n_kv = n_past + T
    q = rng.normal(0, 1, (T, H, dh)).astype(np.float32)
    K = rng.normal(0, 1, (n_ctx, Hkv, dh)).astype(np.float16)
    V = rng.normal(0, 1, (Hkv, dh, n_ctx)).astype(np.float16)
    K[n_kv:] = 0; V[:, :, n_kv:] = 0
    out = P.attn_decode(_dev(P, q.reshape(T, -1)), _f16bits(P, K), _f16bits(P, V), n_past, H, Hkv, dh, n_ctx,
                        1.0 / np.sqrt(dh)).cpu().numpy().reshape(T, H, dh)
    qh = q.astype(np.float16).astype(np.float64)
    for t in range(T):
        nk = n_past + t + 1
        for h in range(H):
            hk = h // (H // Hkv)
            s = (K[:nk, hk].astype(np.float64) @ qh[t, h]).astype(np.float32) * np.float32(1.0 / np.sqrt(dh))
            e = np.exp((s - s.max()).astype(np.float32)).astype(np.float32)
            p = (e * np.float32(1.0 / e.astype(np.float64).sum())).astype(np.float16).astype(np.float64)
            want = V[hk, :, :nk].astype(np.float64) @ p
            assert np.abs(out[t, h] - want).max() <= 2e-3 * np.abs(want).max() + 1e-4, (t, h)


@pytest.mark.parametrize("dh", [64, 128])
@pytest.mark.parametrize("n_past,T", [(0, 1), (0, 16), (0, 37), (5, 128), (40, 200), (96, 160)])
def test_attn_prefill_mfma_vs_exact(P, oracle, n_past, T, dh):
    """MFMA prefill attention: same rounding points as the reference (q, p -> F16; F16 K/V; f32 accumulate), compared with
    the float64 restatement of llm_build_kqv on the same f16-rounded inputs AND with the per-token kernel."""
    torch = P.torch
    rng = np.random.default_rng(43)
    H, Hkv, n_ctx = 8, 2, 256
    n_kv = n_past + T
    q = rng.normal(0, 1, (T, H, dh)).astype(np.float32)
    K = rng.normal(0, 1, (n_ctx, Hkv, dh)).astype(np.float16)
    V = rng.normal(0, 1, (Hkv, dh, n_ctx)).astype(np.float16)
    K[n_kv:] = 0; V[:, :, n_kv:] = 0
    args = (_dev(P, q.reshape(T, -1)), _f16bits(P, K), _f16bits(P, V), n_past, H, Hkv, dh, n_ctx, 1.0 / np.sqrt(dh))
    out = P.attn_prefill(*args).cpu().numpy().reshape(T, H, dh)
    ref = P.attn_decode(*args).cpu().numpy().reshape(T, H, dh)
    assert np.abs(out - ref).max() <= 2e-3 * np.abs(ref).max()
    qh = q.astype(np.float16).astype(np.float64)
    for t in sorted({0, T // 2, T - 1}):
        nk = n_past + t + 1
        for h in range(H):
            hk = h // (H // Hkv)
            s = (K[:nk, hk].astype(np.float64) @ qh[t, h]).astype(np.float32) * np.float32(1.0 / np.sqrt(dh))
            e = np.exp((s - s.max()).astype(np.float32)).astype(np.float32)
            p = (e * np.float32(1.0 / e.astype(np.float64).sum())).astype(np.float16).astype(np.float64)
            want = V[hk, :, :nk].astype(np.float64) @ p
            assert np.abs(out[t, h] - want).max() <= 2e-3 * np.abs(want).max() + 1e-4, (t, h)


@pytest.mark.parametrize("dh", [64, 128])
@pytest.mark.parametrize("n_past", [0, 5, 255, 256, 700, 1023])
def test_attn_decode_split_equals_single_workgroup_kernel(P, n_past, dh):
    """Long-context path (keys split over workgroups, attn_split.hip) against the per-head kernel: same rounding points,
    only the summation order differs."""
    rng = np.random.default_rng(44)
    H, Hkv, n_ctx = 16, 2, 1024
    n_kv = n_past + 1
    q = rng.normal(0, 1, (1, H, dh)).astype(np.float32)
    K = rng.normal(0, 1, (n_ctx, Hkv, dh)).astype(np.float16)
    V = rng.normal(0, 1, (Hkv, dh, n_ctx)).astype(np.float16)
    K[n_kv:] = 0; V[:, :, n_kv:] = 0
    args = (_dev(P, q.reshape(1, -1)), _f16bits(P, K), _f16bits(P, V), n_past, H, Hkv, dh, n_ctx, 1.0 / np.sqrt(dh))
    ref = P.attn_decode(*args).cpu().numpy()
    out = P.attn_decode_split(*args).cpu().numpy()
    # p is rounded to F16 in both kernels; a sum of exponentials assembled per chunk can move 1/sum by an ulp and with it a
    # few p values by one F16 ulp (2^-11 relative): bound = a handful of such flips
    assert np.abs(out - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-6


def test_argmax_first_maximum(P):
    torch = P.torch
    x = torch.randn(128256, device="cuda")
    x[777] = 50.0; x[90000] = 50.0
    assert int(P.argmax(x).item()) == 777
    x[3] = 60.0
    assert int(P.argmax(x).item()) == 3


@pytest.mark.parametrize("t", QUANT_TYPES)
def test_get_rows_bitexact(P, oracle, t):
    torch = P.torch
    rng = np.random.default_rng(43)
    K, N = 768, 50
    blocks = rand_blocks(t, N, K, rng, scale=1.0)
    w = P.upload_weight(t, blocks, K, N)
    toks = np.array([0, 49, 7, 7, 23], dtype=np.int32)
    got = P.get_rows(w, _dev(P, toks)).cpu().numpy()
    rs = row_size(t, K)
    for i, tk in enumerate(toks):
        want = oracle.dequantize_row(t, blocks[tk * rs:(tk + 1) * rs], K)
        assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), (t, i)


@pytest.mark.parametrize("mode,dh", [(0, 128), (2, 128), (0, 64)])
@pytest.mark.parametrize("n_past", [0, 1, 7, 8, 9, 63, 300])
def test_attn_rope_fused_equals_two_kernel_path(P, mode, dh, n_past):
    """The fused single-token kernel must agree with rope_kv_store + attn_decode (same rounding points) and must
    leave the caches in exactly the same state."""
    torch = P.torch
    rng = np.random.default_rng(44)
    H, Hkv, n_ctx = 8, 2, 512
    K0 = rng.normal(0, 1, (n_ctx, Hkv * dh)).astype(np.float16)
    V0 = rng.normal(0, 1, (Hkv * dh, n_ctx)).astype(np.float16)
    K0[n_past:] = 0; V0[:, n_past:] = 0
    q = rng.normal(0, 1, (1, H * dh)).astype(np.float32)
    k = rng.normal(0, 1, (1, Hkv * dh)).astype(np.float32)
    v = rng.normal(0, 1, (1, Hkv * dh)).astype(np.float32)
    ff = (1 + rng.uniform(0, 7, dh // 2)).astype(np.float32)
    kw = dict(freq_factors=_dev(P, ff), mode=mode, freq_base=500000.0)
    kc1, vc1 = _f16bits(P, K0), _f16bits(P, V0)
    qr, _ = P.rope_kv_store(_dev(P, q), _dev(P, k), _dev(P, v), kc1, vc1, n_past, H, Hkv, dh, n_ctx, **kw)
    want = P.attn_decode(qr, kc1, vc1, n_past, H, Hkv, dh, n_ctx, 1.0 / np.sqrt(dh))
    kc2, vc2 = _f16bits(P, K0), _f16bits(P, V0)
    got = P.attn_rope_fused(_dev(P, q), _dev(P, k), _dev(P, v), kc2, vc2, n_past, H, Hkv, dh, n_ctx, 1.0 / np.sqrt(dh), **kw)
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    w, g = want.cpu().numpy(), got.cpu().numpy()
    assert np.abs(w - g).max() <= 2e-6 * max(1.0, np.abs(w).max())      # summation order only


@pytest.mark.parametrize("t", QUANT_TYPES)
@pytest.mark.parametrize("K", [256, 4096, 8192, 14336, 28672])
@pytest.mark.parametrize("norm", [False, True])
def test_fused_prologue_equals_separate_kernels_bitexact(P, t, K, norm):
    """f32 -> (rms_norm) -> quantize inside the GEMV prologue must give EXACTLY what the stand-alone rms_norm /
    quantize kernels + pre-quantized GEMV give (same integers, same float order)."""
    torch = P.torch
    rng = np.random.default_rng(51)
    N = 33
    w = P.upload_weight(t, rand_blocks(t, N, K, rng), K, N)
    x = torch.from_numpy(rng.normal(0, 1.5, (1, K)).astype(np.float32)).cuda()
    x[0, 3] = 7.5; x[0, 5] = -7.5                    # tie on |max| inside the first block
    nw = torch.from_numpy((1 + rng.normal(0, 0.05, K)).astype(np.float32)).cuda() if norm else None
    resid = torch.from_numpy(rng.normal(0, 1, N).astype(np.float32)).cuda()
    xn = P.rms_norm(x, nw, 1e-5) if norm else x
    want = P.mul_mat_vec(w, x=xn, resid=resid.view(1, -1))[0]
    got = P.mul_mat_vec_fused([w], x, norm_w=nw, eps=1e-5, resids=[resid])[0]
    assert torch.equal(got, want), (got - want).abs().max()


@pytest.mark.experiments
@pytest.mark.parametrize("t", [Q4_K, Q6_K, Q8_0])
@pytest.mark.parametrize("K_in,E", [(4096, 4096), (28672, 8192), (1024, 768)])
def test_producer_side_sum_of_squares_gives_the_same_q8_K_blocks(P, oracle, t, K_in, E):
    """Round 5: the wo / ffn_down launch leaves per-workgroup f64 partials of sum (f32 x^2) of the row it writes (+ residual); the next launch's
    rms_norm prologue adds them instead of reducing the row (ggml_compute_forward_rms_norm_f32, ggml.c:11975-11980). Acceptance: the consumer's
    outputs - i.e. its Q8_K / Q8_0 activation blocks and everything after them - are bit-identical to the plain form, and the partials add up to
    the f64 sum of the f32-rounded squares."""
    torch = P.torch
    rng = np.random.default_rng(505 + K_in + E)
    wp = P.upload_weight(Q4_K, rand_blocks(Q4_K, E, K_in, rng), K_in, E)                 # producer: E rows (wo / ffn_down), + residual
    a = torch.from_numpy(rng.normal(0, 1.0, (1, K_in)).astype(np.float32)).cuda()
    resid = torch.from_numpy(rng.normal(0, 2.0, E).astype(np.float32)).cuda()
    (y_ss,), ss = P.mul_mat_vec_fused_ss([wp], a, resids=[resid], want_sumsq=True)
    y_plain = P.mul_mat_vec_fused([wp], a, resids=[resid])[0]
    assert torch.equal(y_ss, y_plain)                                                       # the producer's output itself is untouched
    y64 = (y_ss.cpu().numpy().astype(np.float32) ** 2).astype(np.float32).astype(np.float64)   # f32-rounded squares, then widened
    assert abs(float(ss.sum().cpu()) - y64.sum()) <= 1e-12 * y64.sum()
    # per-workgroup slices: workgroup b owns rows [E b / G, E (b + 1) / G)
    G = ss.numel()
    for b in (0, G // 2, G - 1):
        r0, r1 = E * b // G, E * (b + 1) // G
        assert abs(float(ss[b].cpu()) - y64[r0:r1].sum()) <= 1e-12 * max(1.0, y64[r0:r1].sum())
    # consumer: rms_norm(y) * w -> quantize -> mat-vec, with and without the partials
    Nc = 70
    wc = P.upload_weight(t, rand_blocks(t, Nc, E, rng), E, Nc)
    nw = torch.from_numpy((1 + rng.normal(0, 0.05, E)).astype(np.float32)).cuda()
    want = P.mul_mat_vec_fused([wc], y_ss.view(1, -1), norm_w=nw, eps=1e-5)[0]
    got = P.mul_mat_vec_fused_ss([wc], y_ss.view(1, -1), norm_w=nw, eps=1e-5, sumsq_in=ss)[0]
    assert torch.equal(got, want), (got - want).abs().max()
    # and a pair launch (ffn_gate | ffn_up) as consumer is covered by the engine test (PM355_SS=0 vs default: identical logits)


def test_fused_qkv_mixed_types_one_launch(P, oracle):
    """wq/wk (Q4_K) + wv (Q6_K or Q5_K) in one launch == three separate mat-vecs, and == the oracle's mul_mat."""
    torch = P.torch
    rng = np.random.default_rng(52)
    K = 4096
    for tv in (Q6_K, Q5_K, Q4_K):
        blocks = [rand_blocks(Q4_K, 96, K, rng), rand_blocks(Q4_K, 40, K, rng), rand_blocks(tv, 40, K, rng)]
        ws = [P.upload_weight(Q4_K, blocks[0], K, 96), P.upload_weight(Q4_K, blocks[1], K, 40), P.upload_weight(tv, blocks[2], K, 40)]
        x = rng.normal(0, 1, (1, K)).astype(np.float32)
        nw = (1 + rng.normal(0, 0.05, K)).astype(np.float32)
        bias = [torch.from_numpy(rng.normal(0, 1, w.N).astype(np.float32)).cuda() for w in ws]
        ys = P.mul_mat_vec_fused(ws, _dev(P, x), norm_w=_dev(P, nw), eps=1e-6, biases=bias)
        xn = oracle.rms_norm(x, nw, 1e-6)
        for w, b, y, bl in zip(ws, bias, ys, blocks):
            sep = P.mul_mat_vec_fused([w], _dev(P, x), norm_w=_dev(P, nw), eps=1e-6, biases=[b])[0]
            assert torch.equal(y, sep)
            want = oracle.mul_mat(w.type, bl, K, w.N, xn)[0] + b.cpu().numpy()
            assert np.allclose(y.cpu().numpy(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("t", QUANT_TYPES)
@pytest.mark.parametrize("K,N,T", [(256, 132, 17), (1024, 256, 128), (768, 64, 200), (2048, 516, 300), (1536, 260, 129), (6144, 64, 96), (8192, 260, 33), (4096, 516, 64)])
def test_mfma_prefill_gemm_vs_oracle(P, oracle, t, K, N, T):
    """MFMA batched GEMM (F16 tiles, f32 accumulate, activations not re-quantized) against the reference arithmetic
    (activations quantized to Q8_K/Q8_0): the reference's own backend tolerance is NMSE <= 5e-4
    (tests/test-backend-ops.cpp:1660); also against an exact f64 product of the dequantized weights."""
    rng = np.random.default_rng(71)
    if t == Q8_0 and K == 1536:
        K = 1152                                           # 4.5 super-blocks of 256: Q8_0 rows may end on half of one (Qwen2.5-72B's ffn_down, K = 29568)
    blocks = rand_blocks(t, N, K, rng, scale=1.0)
    w = P.upload_weight(t, blocks, K, N)
    x = rng.normal(0, 1, (T, K)).astype(np.float32)
    bias = rng.normal(0, 1, N).astype(np.float32)
    resid = rng.normal(0, 1, (T, N)).astype(np.float32)
    got = P.mul_mat_mfma(w, _dev(P, x), bias=_dev(P, bias), resid=_dev(P, resid)).cpu().numpy()
    rs = row_size(t, K)
    Wf = np.stack([oracle.dequantize_row(t, blocks[r * rs:(r + 1) * rs], K) for r in range(N)]).astype(np.float64)
    exact = x.astype(np.float64) @ Wf.T + bias + resid
    nm_exact = ((got - exact) ** 2).sum() / (exact ** 2).sum()
    assert nm_exact < 1e-6, nm_exact                       # f16 rounding of operands only
    ref = oracle.mul_mat(t, blocks, K, N, x) + bias + resid
    nm_ref = ((got - ref) ** 2).sum() / (ref ** 2).sum()
    assert nm_ref < 5e-4, nm_ref


@pytest.mark.parametrize("t", [Q4_K, Q6_K])
def test_prompt_gemm_tail_tiles_split_along_k(P, oracle, t):
    """A launch of full rounds of workgroups + a short tail (17 row tiles x 16 token tiles: 32 whole-tile slots per XCD, then 16 slots on one XCD): only the
    tail's tiles are split along K (two slices each, slabs summed in slice order by the last arriver) - against the exact f64 product of the dequantized
    weights, whole tiles and split tiles separately, and bit-identical between two launches."""
    torch = P.torch
    rng = np.random.default_rng(91 + t)
    K, N, T = 2048, 4352, 4096
    blocks = rand_blocks(t, N, K, rng, scale=1.0)
    w = P.upload_weight(t, blocks, K, N)
    x = rng.normal(0, 1, (T, K)).astype(np.float32)
    bias = rng.normal(0, 1, N).astype(np.float32)
    xd, bd = _dev(P, x), _dev(P, bias)
    got = P.mul_mat_mfma(w, xd, bias=bd)
    again = P.mul_mat_mfma(w, xd, bias=bd)
    assert torch.equal(got, again)
    rs = row_size(t, K)
    Wf = np.stack([oracle.dequantize_row(t, blocks[r * rs:(r + 1) * rs], K) for r in range(N)]).astype(np.float64)
    exact = x.astype(np.float64) @ Wf.T + bias
    g = got.cpu().numpy()
    for lo, hi in ((0, 4096), (4096, N)):             # whole tiles | the split tail (row tile 16 = XCD 0's third)
        nm = ((g[:, lo:hi] - exact[:, lo:hi]) ** 2).sum() / (exact[:, lo:hi] ** 2).sum()
        assert nm < 1e-6, (lo, nm)


@pytest.mark.parametrize("types", [(Q4_K, Q4_K, Q6_K), (Q4_K, Q4_K, Q5_K), (Q6_K, Q6_K, Q6_K), (Q5_K, Q6_K), (Q4_K,)])
@pytest.mark.parametrize("T", [77, 300])
def test_prompt_gemm_jobs_one_launch(P, oracle, types, T):
    """wq | wk | wv of a prompt batch as jobs of ONE launch (mmq_pf.hip), each with its own quant type, bias and row count (incl. a ragged last tile):
    every job's output is bit-identical to its own single launch, and equals the f64 product of the dequantized weights to F16 operand rounding."""
    torch = P.torch
    rng = np.random.default_rng(81)
    K = 1024
    Ns = [520, 132, 256][:len(types)]
    blocks = [rand_blocks(t, n, K, rng, scale=1.0) for t, n in zip(types, Ns)]
    ws = [P.upload_weight(t, b, K, n) for t, b, n in zip(types, blocks, Ns)]
    x = rng.normal(0, 1, (T, K)).astype(np.float32)
    biases = [torch.from_numpy(rng.normal(0, 1, n).astype(np.float32)).cuda() for n in Ns]
    xd = _dev(P, x)
    ys = P.mul_mat_mfma_multi(ws, xd, biases=biases)
    for w, t, b, bl, y in zip(ws, types, biases, blocks, ys):
        single = P.mul_mat_mfma(w, xd, bias=b)
        assert torch.equal(y, single)
        rs = row_size(t, K)
        Wf = np.stack([oracle.dequantize_row(t, bl[r * rs:(r + 1) * rs], K) for r in range(w.N)]).astype(np.float64)
        exact = x.astype(np.float64) @ Wf.T + b.cpu().numpy()
        nm = ((y.cpu().numpy() - exact) ** 2).sum() / (exact ** 2).sum()
        assert nm < 1e-6, (t, nm)


@pytest.mark.parametrize("t", [Q4_K, Q5_K, Q6_K])
@pytest.mark.parametrize("K,N,T", [(1024, 384, 200), (2048, 132, 64), (512, 256, 257)])
def test_prompt_gemm_pair_tiles_silu_gate_times_up(P, oracle, t, K, N, T):
    """ffn_gate | ffn_up as one launch of pair tiles: silu(gate) * up formed inside the workgroup == the two separate launches followed by the product
    (the reference's MUL_MAT, MUL_MAT, SILU, MUL: llm_build_ffn, src/llama.cpp:9804)."""
    torch = P.torch
    rng = np.random.default_rng(83)
    wg = P.upload_weight(t, rand_blocks(t, N, K, rng, scale=1.0), K, N)
    wu = P.upload_weight(t, rand_blocks(t, N, K, rng, scale=1.0), K, N)
    x = _dev(P, rng.normal(0, 1, (T, K)).astype(np.float32))
    got = P.mul_mat_mfma_pair(wg, wu, x)
    g, u = P.mul_mat_mfma(wg, x), P.mul_mat_mfma(wu, x)
    want = (g.double() / (1.0 + torch.exp(-g.double()))) * u.double()
    err = (got.double() - want).abs().max().item()
    assert err <= 2e-6 * want.abs().max().item(), err


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("tv", [Q6_K, Q5_K, Q4_K])
@pytest.mark.parametrize("dh,H,Hkv", [(128, 8, 4), (64, 16, 8)])
def test_qkv_epilogue_and_cached_attention_equal_the_fused_attention_path(P, oracle, tv, dh, H, Hkv, mode):
    """Round-3 decode form - RoPE + F16 KV store in the EPILOGUE of the wq | wk | wv launch (per-token cos / sin table, wk / wv dealt out
    step by step over all waves), attention over cached cells only (one-barrier kernel up to 64 cells, per-head body beyond) - against
    the round-2 form (plain mat-vecs, rope + store inside the fused attention kernel) on a run of tokens that crosses the 64-cell
    boundary; the mat-vec outputs against the oracle's mul_mat. mode 2 = NEOX rope (build_qwen2: pairs (i, i + n_rot / 2); every workgroup's
    slice of wq / wk is two runs of rows n_rot / 2 apart, GemvJob::nx_s)."""
    torch = P.torch
    rng = np.random.default_rng(61)
    if tv != Q4_K and mode == 2 and dh == 64:
        pytest.skip("one attn_v type per NEOX shape is enough")
    K, n_ctx = 1024, 128
    Nq, Nkv = H * dh, Hkv * dh
    assert Nq % 512 == 0 and Nkv % 512 == 0
    blocks = [rand_blocks(Q4_K, Nq, K, rng), rand_blocks(Q4_K, Nkv, K, rng), rand_blocks(tv, Nkv, K, rng)]
    ws = [P.upload_weight(Q4_K, blocks[0], K, Nq), P.upload_weight(Q4_K, blocks[1], K, Nkv), P.upload_weight(tv, blocks[2], K, Nkv)]
    nw = _dev(P, (1 + rng.normal(0, 0.05, K)).astype(np.float32))
    bias = [_dev(P, rng.normal(0, 0.3, n).astype(np.float32)) for n in (Nq, Nkv, Nkv)]
    ff = _dev(P, (1 + rng.uniform(0, 7, dh // 2)).astype(np.float32))
    kcA = torch.zeros(n_ctx * Nkv, dtype=torch.int16, device="cuda"); vcA = torch.zeros_like(kcA)
    kcB = torch.zeros_like(kcA); vcB = torch.zeros_like(kcA)
    kcC = torch.zeros_like(kcA); vcC = torch.zeros_like(kcA)          # ggml-graph mode of the same launches (cell / cells attended + mask)
    mask0 = torch.zeros(n_ctx, dtype=torch.float32, device="cuda")
    scale = 1.0 / np.sqrt(dh)
    worst = 0.0
    for pos in list(range(0, 6)) + [31, 32, 62, 63, 64, 65, 90]:
        x = rng.normal(0, 1, (1, K)).astype(np.float32)
        xd = _dev(P, x)
        qA, kA, vA = P.mul_mat_vec_fused(ws, xd, norm_w=nw, eps=1e-5, biases=bias)
        outA = P.attn_rope_fused(qA, kA, vA, kcA, vcA, pos, H, Hkv, dh, n_ctx, scale, freq_factors=ff, freq_base=500000.0, mode=mode)
        pd = torch.tensor([pos], dtype=torch.int32, device="cuda")
        tab = P.rope_table(pd, dh, freq_factors=ff, freq_base=500000.0, mode=mode)
        qB = P.mul_mat_vec_qkv(ws, xd, tab, pd, kcB, vcB, Hkv, dh, n_ctx, norm_w=nw, eps=1e-5, biases=bias, neox=mode == 2)
        outB = P.attn_cached(qB, kcB, vcB, pd, H, Hkv, dh, n_ctx, scale)
        dyn = torch.tensor([pos, pos + 1], dtype=torch.int32, device="cuda")
        qC = P.mul_mat_vec_qkv(ws, xd, tab, None, kcC, vcC, Hkv, dh, n_ctx, norm_w=nw, eps=1e-5, biases=bias, cell_nkv=dyn, neox=mode == 2)
        outC = P.attn_cached(qC, kcC, vcC, None, H, Hkv, dh, n_ctx, scale, cell_nkv=dyn, mask=mask0, max_keys=n_ctx)
        torch.cuda.synchronize()
        assert torch.equal(qC, qB) and torch.equal(outC, outB) and torch.equal(kcC, kcB) and torch.equal(vcC, vcB), pos
        # the K row / V column of this cell: the same F16 values up to the summation order of the split wk / wv rows (<= 1 F16 ulp, rarely)
        a = kcA.cpu().numpy().view(np.float16).astype(np.float32).reshape(n_ctx, Nkv)[pos]
        b = kcB.cpu().numpy().view(np.float16).astype(np.float32).reshape(n_ctx, Nkv)[pos]
        assert np.abs(a - b).max() <= 2e-3 * max(1.0, np.abs(a).max()) and (a != b).mean() < 0.02, (pos, np.abs(a - b).max(), (a != b).mean())
        va = vcA.cpu().numpy().view(np.float16).astype(np.float32).reshape(Nkv, n_ctx)[:, pos]
        vb = vcB.cpu().numpy().view(np.float16).astype(np.float32).reshape(Nkv, n_ctx)[:, pos]
        assert np.abs(va - vb).max() <= 2e-3 * max(1.0, np.abs(va).max()) and (va != vb).mean() < 0.02
        # q: rotated + F16-rounded == F16 rounding of the oracle's rope of the round-2 q
        q_ref = oracle.rope(qA.cpu().numpy().reshape(1, H, dh), np.array([pos], dtype=np.int32), freq_factors=ff.cpu().numpy(), freq_base=500000.0, mode=mode)
        q_ref = q_ref.astype(np.float16).astype(np.float32).reshape(-1)
        qb = qB.cpu().numpy()
        assert np.abs(qb - q_ref).max() <= 2e-3 * max(1.0, np.abs(q_ref).max()) and (qb != q_ref).mean() < 0.02
        # from here on both caches must hold the SAME bytes, or later cells would compare different histories
        kcB.copy_(kcA); vcB.copy_(vcA); kcC.copy_(kcA); vcC.copy_(vcA)
        wa, wb = outA.cpu().numpy(), outB.cpu().numpy()
        worst = max(worst, float(np.abs(wa - wb).max() / max(1.0, np.abs(wa).max())))
        assert np.abs(wa - wb).max() <= 3e-3 * max(1.0, np.abs(wa).max()), (pos, np.abs(wa - wb).max())
        # cells in between: identical random history for both paths
        if pos >= 5:
            nxt = {5: 31, 31: 32, 32: 62, 62: 63, 63: 64, 64: 65, 65: 90}.get(pos)
            if nxt:
                fill_k = rng.normal(0, 1, (nxt - pos - 1, Nkv)).astype(np.float16).view(np.int16)
                fill_v = rng.normal(0, 1, (Nkv, nxt - pos - 1)).astype(np.float16).view(np.int16)
                if fill_k.size:
                    kcA.view(n_ctx, Nkv)[pos + 1:nxt] = torch.from_numpy(fill_k).cuda()
                    vcA.view(Nkv, n_ctx)[:, pos + 1:nxt] = torch.from_numpy(fill_v).cuda()
                    kcB.copy_(kcA); vcB.copy_(vcA); kcC.copy_(kcA); vcC.copy_(vcA)
    print(f"\n[qkv epilogue + cached attention, dh {dh}, attn_v type {tv}] max |d out| / max |out| = {worst:.2e}")
    # and the epilogue's mat-vecs against the oracle (q before rope is not observable: check k / v through the cache at one more cell)
    x = rng.normal(0, 1, (1, K)).astype(np.float32)
    pd = torch.tensor([100], dtype=torch.int32, device="cuda")
    tab = P.rope_table(pd, dh, freq_factors=ff, freq_base=500000.0, mode=mode)
    P.mul_mat_vec_qkv(ws, _dev(P, x), tab, pd, kcB, vcB, Hkv, dh, n_ctx, norm_w=nw, eps=1e-5, biases=bias, neox=mode == 2)
    xn = oracle.rms_norm(x, nw.cpu().numpy(), 1e-5)
    v_ref = oracle.mul_mat(tv, blocks[2], K, Nkv, xn)[0] + bias[2].cpu().numpy()
    v_got = vcB.cpu().numpy().view(np.float16).astype(np.float32).reshape(Nkv, n_ctx)[:, 100]
    assert np.allclose(v_got, v_ref.astype(np.float16).astype(np.float32), rtol=2e-3, atol=2e-3)
    k_ref = oracle.mul_mat(Q4_K, blocks[1], K, Nkv, xn)[0] + bias[1].cpu().numpy()
    k_ref = oracle.rope(k_ref.reshape(1, Hkv, dh), np.array([100], dtype=np.int32), freq_factors=ff.cpu().numpy(), freq_base=500000.0, mode=mode).reshape(-1)
    k_got = kcB.cpu().numpy().view(np.float16).astype(np.float32).reshape(n_ctx, Nkv)[100]
    assert np.allclose(k_got, k_ref.astype(np.float16).astype(np.float32), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("dh,H,Hkv", [(128, 8, 1), (128, 8, 2), (64, 8, 4), (64, 4, 4), (64, 16, 1)])
def test_long_context_matrix_core_attention_over_cached_cells(P, dh, H, Hkv):
    """attn_flash_mfma.hip: scores and P.V of a GQA group on v_mfma_f32_16x16x32_f16 with permuted key rows, per-wave online softmax, spans
    merged in the launch - against softmax(K q * scale + mask) V in float64 on the same F16 cache values, over cells-attended that
    exercise: one partial tile, one span, several spans with a ragged tail, the doubled span (> 8k cells), with and without a mask that
    hides cells (-inf), position given as d_pos and as {cell, cells attended}."""
    torch = P.torch
    rng = np.random.default_rng(77 + dh + H)
    n_ctx = 16640
    Nkv = Hkv * dh
    R = H // Hkv
    kc = rng.normal(0, 1, (n_ctx, Nkv)).astype(np.float16)
    vt = rng.normal(0, 1, (Nkv, n_ctx)).astype(np.float16)
    kcd = torch.from_numpy(kc.view(np.int16)).cuda().reshape(-1)
    vcd = torch.from_numpy(vt.view(np.int16)).cuda().reshape(-1)
    scratch = P.attn_split_scratch(H, dh, n_ctx)
    scale = 1.0 / np.sqrt(dh)
    worst = 0.0
    for n_kv, use_mask, dyn_mode in [(1, False, False), (33, False, True), (700, True, True), (2000, False, False), (2049, True, True),
                                     (5000, False, True), (8192, False, False), (16300, True, True)]:
        q = (rng.normal(0, 1, (H, dh)) * 1.5).astype(np.float16).astype(np.float32)      # (the epilogue hands over F16-rounded values)
        qd = torch.from_numpy(q.reshape(1, -1)).cuda()
        mask = None
        mnp = np.zeros(n_kv, dtype=np.float64)
        if use_mask:
            m32 = np.zeros(n_ctx, dtype=np.float32)
            hide = rng.random(n_kv) < 0.3
            hide[n_kv - 1] = False
            m32[:n_kv][hide] = -np.inf
            m32[n_kv:] = -np.inf
            mask = torch.from_numpy(m32).cuda()
            mnp = m32[:n_kv].astype(np.float64)
        cells = 1024
        while cells < n_kv: cells *= 2
        cells = min(cells, n_ctx)
        if dyn_mode:
            dyn = torch.tensor([n_kv - 1, n_kv], dtype=torch.int32, device="cuda")
            out = P.attn_cached(qd, kcd, vcd, None, H, Hkv, dh, n_ctx, scale, cell_nkv=dyn, mask=mask, max_keys=cells, scratch=scratch)
        else:
            pd = torch.tensor([n_kv - 1], dtype=torch.int32, device="cuda")
            out = P.attn_cached(qd, kcd, vcd, pd, H, Hkv, dh, n_ctx, scale, mask=mask, max_keys=0, scratch=scratch)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(H, dh)
        for h in range(H):
            g = h // R
            Kh = kc[:n_kv, g * dh:(g + 1) * dh].astype(np.float64)
            Vh = vt[g * dh:(g + 1) * dh, :n_kv].astype(np.float64)
            s = Kh @ q[h].astype(np.float64) * scale + mnp
            pr = np.exp(s - s.max()); pr /= pr.sum()
            ref = Vh @ pr
            err = np.abs(got[h] - ref).max() / max(1e-3, np.abs(ref).max())
            worst = max(worst, float(err))
            assert err < 2e-3, (n_kv, h, err)
    print(f"\n[matrix-core long-context attention dh {dh} H {H} Hkv {Hkv}] worst max|d| / max|ref| = {worst:.2e}")


def _attn_refs(kc, vt, q, n_kv, H, Hkv, dh, scale):
    """float64 attention over F16 cache values, and the reference graph's rounding points (MUL_MAT(k, q) -> SOFT_MAX_EXT -> MUL_MAT(v, kq), ggml.c:12445-12473:
    the NORMALISED probabilities are rounded to F16 as src1 of the V^T.p product, accumulation in f32): [H][dh] each."""
    R = H // Hkv
    ref64 = np.empty((H, dh)); ref16 = np.empty((H, dh))
    for g in range(Hkv):
        Kh = kc[:n_kv, g * dh:(g + 1) * dh].astype(np.float32)
        Vh = vt[g * dh:(g + 1) * dh, :n_kv].astype(np.float32)
        for h in range(g * R, (g + 1) * R):
            s = (Kh.astype(np.float64) @ q[h].astype(np.float64)) * scale
            pr = np.exp(s - s.max()); pr /= pr.sum()
            ref64[h] = Vh.astype(np.float64) @ pr
            s32 = (Kh @ q[h]).astype(np.float32) * np.float32(scale)
            e32 = np.exp(s32 - s32.max()).astype(np.float32)
            p16 = (e32 / np.float32(e32.astype(np.float64).sum())).astype(np.float16).astype(np.float32)
            ref16[h] = (Vh.astype(np.float64) @ p16.astype(np.float64))
    return ref64, ref16


@pytest.mark.parametrize("dh,H,Hkv", [(128, 64, 8), (64, 12, 4)])
@pytest.mark.parametrize("n_kv", [700, 8192])
def test_long_context_attention_on_a_row_major_v_cache(P, n_kv, dh, H, Hkv):
    """--flash-attn layout (V cache [cell][n_head_kv * head_dim], F16 mask; llm_build_kv, src/llama.cpp:9705, :10075-10095) on the matrix-core kernel: the wave's
    V rows go through its LDS image and come back as operands through the transposing LDS read. Same cache VALUES as the transposed layout -> the two layouts'
    results must be bitwise equal; and both against float64. A sparse F16 mask (-inf on every 7th cell) checks the masked path on the way."""
    torch = P.torch
    ATTN_V_ROWMAJOR, ATTN_MASK_F16 = 1, 2                                    # include/prima_mi355.h: PM355_ATTN_V_ROWMAJOR, PM355_ATTN_MASK_F16
    rng = np.random.default_rng(7000 + n_kv + dh)
    n_ctx = 8192 + 256
    Nkv = Hkv * dh
    kc = rng.normal(0, 1, (n_ctx, Nkv)).astype(np.float16)
    vt = rng.normal(0, 1, (Nkv, n_ctx)).astype(np.float16)
    vrow = np.ascontiguousarray(vt.T)                                        # [cell][Hkv * dh]
    kcd = torch.from_numpy(kc.view(np.int16)).cuda().reshape(-1)
    vtd = torch.from_numpy(vt.view(np.int16)).cuda().reshape(-1)
    vrd = torch.from_numpy(vrow.view(np.int16)).cuda().reshape(-1)
    scale = 1.0 / np.sqrt(dh)
    q = (rng.normal(0, 1, (H, dh)) * 1.5).astype(np.float16).astype(np.float32)
    qd = torch.from_numpy(q.reshape(1, -1)).cuda()
    pd = torch.tensor([n_kv - 1], dtype=torch.int32, device="cuda")
    mnp = np.zeros(n_ctx, dtype=np.float16); mnp[::7] = -np.inf
    md = torch.from_numpy(mnp.view(np.int16)).cuda()
    cells = 1024
    while cells < n_kv: cells *= 2
    cells = min(cells, n_ctx)
    for mask, flags in ((None, 0), (md, ATTN_MASK_F16)):
        s_t, s_r = P.attn_split_scratch(H, dh, n_ctx), P.attn_split_scratch(H, dh, n_ctx)
        out_t = P.attn_cached(qd, kcd, vtd, pd, H, Hkv, dh, n_ctx, scale, mask=mask, max_keys=cells, flags=flags, scratch=s_t).cpu().numpy().reshape(H, dh)
        out_r = P.attn_cached(qd, kcd, vrd, pd, H, Hkv, dh, n_ctx, scale, mask=mask, max_keys=cells, flags=flags | ATTN_V_ROWMAJOR, scratch=s_r).cpu().numpy().reshape(H, dh)
        assert np.isfinite(out_r).all()
        assert np.array_equal(out_t, out_r), np.abs(out_t - out_r).max()
        R = H // Hkv
        for h in range(H):
            g = h // R
            Kh = kc[:n_kv, g * dh:(g + 1) * dh].astype(np.float64)
            Vh = vt[g * dh:(g + 1) * dh, :n_kv].astype(np.float64)
            sc = Kh @ q[h].astype(np.float64) * scale + (mnp[:n_kv].astype(np.float64) if mask is not None else 0.0)
            pr = np.exp(sc - sc.max()); pr /= pr.sum()
            ref = Vh @ pr
            assert np.abs(out_r[h] - ref).max() / max(1e-3, np.abs(ref).max()) < 2e-3, (n_kv, h)


@pytest.mark.parametrize("n_kv", [8192, 32700])
def test_long_context_attention_at_the_70b_head_shape(P, n_kv):
    """The shape the long-context numbers are quoted on - 64 query heads, 8 KV heads, head_dim 128, 8k and 32.7k cells (grid = 8 x 32 spans: every
    resident slot of the chip) - against float64 AND against the reference graph's own rounding points (p rounded to F16 AFTER the normalisation;
    the kernel rounds exp(s - m) before it, attn_flash_mfma.hip): both differences are measured and bounded."""
    torch = P.torch
    H, Hkv, dh = 64, 8, 128
    rng = np.random.default_rng(1000 + n_kv)
    n_ctx = 32768 + 256
    Nkv = Hkv * dh
    kc = rng.normal(0, 1, (n_ctx, Nkv)).astype(np.float16)
    vt = rng.normal(0, 1, (Nkv, n_ctx)).astype(np.float16)
    kcd = torch.from_numpy(kc.view(np.int16)).cuda().reshape(-1)
    vcd = torch.from_numpy(vt.view(np.int16)).cuda().reshape(-1)
    scratch = P.attn_split_scratch(H, dh, n_ctx)
    scale = 1.0 / np.sqrt(dh)
    q = (rng.normal(0, 1, (H, dh)) * 1.5).astype(np.float16).astype(np.float32)
    qd = torch.from_numpy(q.reshape(1, -1)).cuda()
    pd = torch.tensor([n_kv - 1], dtype=torch.int32, device="cuda")
    cells = 1024
    while cells < n_kv: cells *= 2
    cells = min(cells, n_ctx)
    outs = [P.attn_cached(qd, kcd, vcd, pd, H, Hkv, dh, n_ctx, scale, max_keys=cells, scratch=scratch).cpu().numpy().reshape(H, dh) for _ in range(3)]
    assert all(np.array_equal(outs[0], o) for o in outs[1:])              # launch to launch: bitwise (fixed merge order)
    got = outs[0]
    ref64, ref16 = _attn_refs(kc, vt, q, n_kv, H, Hkv, dh, scale)
    den = np.maximum(1e-3, np.abs(ref64).max(axis=1))
    e64 = (np.abs(got - ref64).max(axis=1) / den).max()
    e16 = (np.abs(got - ref16).max(axis=1) / den).max()
    r1664 = (np.abs(ref16 - ref64).max(axis=1) / den).max()
    print(f"\n[matrix-core attention, H 64 / Hkv 8 / dh 128, {n_kv} cells] max|d| / max|ref|: kernel vs float64 {e64:.2e}, kernel vs F16-rounded-p reference {e16:.2e}, "
          f"that reference vs float64 {r1664:.2e}")
    assert np.isfinite(got).all()
    assert e64 < 2e-3 and e16 < 2e-3 + r1664


def test_long_context_attention_merge_survives_a_starved_chip(P):
    """The in-launch merge of attn_flash_mfma.hip lets the last arrivals of a KV head spin (bounded) for workgroups still behind them. Provoked here:
    a co-running stream-read kernel holds every CU's wave slots while the 8 x 32 attention grid is launched on another stream, so its workgroups
    trickle in as slots free up. Required: the launch ENDS (no hang) and the result is either correct or the fail-loud NaN - never a wrong number."""
    import ctypes as C
    torch = P.torch
    plib = P.L.load_probe()
    plib.pm355_probe_stream_read.restype = C.c_int
    plib.pm355_probe_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    H, Hkv, dh, n_kv = 64, 8, 128, 16384
    rng = np.random.default_rng(4242)
    n_ctx = 16384 + 256
    Nkv = Hkv * dh
    kc = rng.normal(0, 1, (n_ctx, Nkv)).astype(np.float16)
    vt = rng.normal(0, 1, (Nkv, n_ctx)).astype(np.float16)
    kcd = torch.from_numpy(kc.view(np.int16)).cuda().reshape(-1)
    vcd = torch.from_numpy(vt.view(np.int16)).cuda().reshape(-1)
    scratch = P.attn_split_scratch(H, dh, n_ctx)
    scale = 1.0 / np.sqrt(dh)
    q = (rng.normal(0, 1, (H, dh)) * 1.5).astype(np.float16).astype(np.float32)
    qd = torch.from_numpy(q.reshape(1, -1)).cuda()
    pd = torch.tensor([n_kv - 1], dtype=torch.int32, device="cuda")
    ref64, _ = _attn_refs(kc, vt, q, n_kv, H, Hkv, dh, scale)
    hog_src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    n_nan = n_ok = 0
    for trial in range(6):
        # two 1024-thread workgroups per CU (every wave slot of the chip), several passes queued back to back on the side stream
        for _ in range(3):
            assert plib.pm355_probe_stream_read(hog_src.data_ptr(), hog_src.numel(), 2, 8, sink.data_ptr(), side.cuda_stream) == 0
        out = P.attn_cached(qd, kcd, vcd, pd, H, Hkv, dh, n_ctx, scale, max_keys=16384, scratch=scratch)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(H, dh)
        if np.isnan(got).any():
            n_nan += 1                                    # fail-loud path: a whole KV head's outputs are NaN
            bad = np.isnan(got).any(axis=1)
            assert all(bad[g * 8:(g + 1) * 8].all() or not bad[g * 8:(g + 1) * 8].any() for g in range(Hkv))
            good = ~bad
        else:
            n_ok += 1
            good = np.ones(H, dtype=bool)
        err = np.abs(got[good] - ref64[good]).max(axis=1) / np.maximum(1e-3, np.abs(ref64[good]).max(axis=1)) if good.any() else np.zeros(1)
        assert err.max() < 2e-3, (trial, err.max())
    print(f"\n[starved merge] {n_ok} launches correct, {n_nan} launches took the fail-loud NaN path; none hung, none wrong")
    # and an undisturbed launch afterwards is clean (the arrival counters are never reset: a timed-out launch must not poison the next)
    out = P.attn_cached(qd, kcd, vcd, pd, H, Hkv, dh, n_ctx, scale, max_keys=16384, scratch=scratch).cpu().numpy().reshape(H, dh)
    assert np.isfinite(out).all() and (np.abs(out - ref64).max(axis=1) / np.maximum(1e-3, np.abs(ref64).max(axis=1))).max() < 2e-3


def _pm_tensor(C, t, type_, ne, nb):
    class PMT(C.Structure):
        _fields_ = [("data", C.c_void_p), ("type", C.c_int32), ("pad_", C.c_int32), ("ne", C.c_int64 * 4), ("nb", C.c_size_t * 4)]
    x = PMT()
    x.data, x.type, x.pad_ = t.data_ptr(), type_, 0
    for i in range(4):
        x.ne[i] = ne[i] if i < len(ne) else 1
        x.nb[i] = nb[i] if i < len(nb) else nb[len(nb) - 1] * (ne[len(nb) - 1] if len(nb) == len(ne) else 1)
    return x


def test_native_q8_0_cache_view_matmul_and_dequantizing_copy(P, oracle):
    """The two node-equivalent ops a `-ctk q8_0` cache needs outside flash attention, on a strided view of native 34-byte blocks shaped like
    llm_build_kqv's k view [head_dim, n_kv, n_head_kv] (row stride = a whole cache row): MUL_MAT(k, q) = quantize_row_q8_0(q) +
    ggml_vec_dot_q8_0_q8_0, against the oracle's mul_mat per head; and CPY -> F32 (K-shift graphs) = dequantize_row_q8_0."""
    import ctypes as C
    torch = P.torch
    lib = P.L.load()
    rng = np.random.default_rng(77)
    dh, Hkv, H, n_kv, n_ctx, T = 128, 2, 4, 40, 64, 3
    Ekv = dh * Hkv
    kf = rng.normal(0, 1, (n_ctx, Ekv)).astype(np.float32)
    blocks = np.concatenate([oracle.quantize_row_q8_0(kf[i]) for i in range(n_ctx)])          # cache rows in native block order
    kc = torch.from_numpy(blocks).cuda()
    row_b, head_b = Ekv // 32 * 34, dh // 32 * 34
    q = rng.normal(0, 1, (H, T, dh)).astype(np.float32)
    qd = torch.from_numpy(q).cuda()
    out = torch.zeros((H, T, n_kv), dtype=torch.float32, device="cuda")
    a = _pm_tensor(C, kc, Q8_0, [dh, n_kv, Hkv, 1], [34, row_b, head_b, n_ctx * row_b])
    b = _pm_tensor(C, qd, 0, [dh, T, H, 1], [4, dh * 4, T * dh * 4, H * T * dh * 4])
    d = _pm_tensor(C, out, 0, [n_kv, T, H, 1], [4, n_kv * 4, T * n_kv * 4, H * T * n_kv * 4])
    lib.pm355_op_mul_mat_f.restype = C.c_int
    lib.pm355_op_mul_mat_f.argtypes = [C.c_void_p] * 4
    assert lib.pm355_op_mul_mat_f(C.addressof(a), C.addressof(b), C.addressof(d), P.stream_ptr()) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    kb = blocks.reshape(n_ctx, Hkv, head_b)
    for h in range(H):
        hk = h // (H // Hkv)
        w = np.ascontiguousarray(kb[:n_kv, hk]).reshape(-1)
        want = oracle.mul_mat(Q8_0, w, dh, n_kv, q[h])                                           # [T, n_kv]
        assert np.allclose(got[h], want, rtol=1e-5, atol=1e-5), (h, np.abs(got[h] - want).max())
    # dequantizing copy of the view [dh, Hkv, n_ctx] -> contiguous f32
    deq = torch.zeros((n_ctx, Hkv, dh), dtype=torch.float32, device="cuda")
    s = _pm_tensor(C, kc, Q8_0, [dh, Hkv, n_ctx, 1], [34, head_b, row_b, n_ctx * row_b])
    t = _pm_tensor(C, deq, 0, [dh, Hkv, n_ctx, 1], [4, dh * 4, Hkv * dh * 4, n_ctx * Hkv * dh * 4])
    lib.pm355_op_cpy.restype = C.c_int
    lib.pm355_op_cpy.argtypes = [C.c_void_p] * 3
    assert lib.pm355_op_cpy(C.addressof(s), C.addressof(t), P.stream_ptr()) == 0
    torch.cuda.synchronize()
    want = np.stack([oracle.dequantize_row(Q8_0, blocks[i * row_b:(i + 1) * row_b], Ekv) for i in range(n_ctx)]).reshape(n_ctx, Hkv, dh)
    assert np.array_equal(deq.cpu().numpy(), want)


@pytest.mark.experiments
@pytest.mark.parametrize("t", [Q4_K, Q6_K])
@pytest.mark.parametrize("K,N", [(1024, 512), (8192, 8192), (4096, 1024), (5120, 2560)])
def test_engine_matvec_phase_is_bit_identical_to_the_launch(P, t, K, N):
    """Persistent decode engine, one mat-vec phase (+ residual, + producer-side partials) against pm355_mul_mat_vec_fused_ss on the same operands: the same
    row loops (the leading steps of a wave's first rows come out of its LDS prefetch slot instead of registers: same consume() calls in the same order)
    -> the same bits; then a two-phase list (wo-like phase -> rms_norm + ffn_gate | ffn_up pair phase fed through the in-launch seam; the pair is Q4_K,
    the only pair type compiled into the engine)."""
    torch = P.torch
    rng = np.random.default_rng(900 + K + N)
    w = P.upload_weight(t, rand_blocks(t, N, K, rng), K, N)
    x = torch.from_numpy(rng.normal(0, 1.0, (1, K)).astype(np.float32)).cuda()
    resid = torch.from_numpy(rng.normal(0, 2.0, N).astype(np.float32)).cuda()
    (want,), ss_want = P.mul_mat_vec_fused_ss([w], x, resids=[resid], want_sumsq=True)
    e = P.EngineRun()
    (got,), ss = e.matvec([w], x, resids=[resid], want_sumsq=True)
    e.run()
    assert torch.equal(got, want), (got - want).abs().max()
    assert torch.equal(ss[:ss_want.numel()], ss_want)
    if N % 256:
        return
    # two phases: y = W x + resid, then h = silu(G n(y)) * (U n(y)) with the sum of squares handed over inside the launch
    F = 768
    g = P.upload_weight(Q4_K, rand_blocks(Q4_K, F, N, rng), N, F)
    u = P.upload_weight(Q4_K, rand_blocks(Q4_K, F, N, rng), N, F)
    nw = torch.from_numpy((1 + rng.normal(0, 0.05, N)).astype(np.float32)).cuda()
    want_h = P.mul_mat_vec_fused([g], want.view(1, -1), norm_w=nw, eps=1e-5, w2s=[u])[0]
    e = P.EngineRun()
    (y1,), ss1 = e.matvec([w], x, resids=[resid], want_sumsq=True)
    (h,) = e.matvec([g], y1, norm_w=nw, eps=1e-5, w2s=[u], sumsq_in=ss1[:ss_want.numel()])
    e.run()
    assert torch.equal(y1, want)
    assert torch.equal(h, want_h), (h - want_h).abs().max()


@pytest.mark.experiments
@pytest.mark.parametrize("t", [Q4_K, Q6_K])
def test_engine_long_rows_continue_from_the_prefetched_steps(P, oracle, t):
    """ffn_down-sized rows (K = 28672: seven steps per row) do not fit a wave's prefetch slot: the first two or three steps of a wave's first row come out
    of LDS, the per-lane partial sums are handed to the register row loop (Item::run_job's c_start / acc0) - same accumulation chain, same bits as the launch."""
    torch = P.torch
    rng = np.random.default_rng(77)
    K, N = 28672, 8192
    blocks = rand_blocks(t, N, K, rng)
    w = P.upload_weight(t, blocks, K, N)
    x = torch.from_numpy(rng.normal(0, 1.0, (1, K)).astype(np.float32)).cuda()
    resid = torch.from_numpy(rng.normal(0, 2.0, N).astype(np.float32)).cuda()
    want = P.mul_mat_vec_fused([w], x, resids=[resid])[0]
    e = P.EngineRun()
    (got,) = e.matvec([w], x, resids=[resid])
    e.run()
    assert torch.equal(got, want), (got - want).abs().max().item()


@pytest.mark.experiments
@pytest.mark.parametrize("tv", [Q6_K, Q5_K, Q4_K])
@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("n_past", [0, 5, 63, 64, 200])
def test_engine_qkv_and_attention_phases_equal_the_launches(P, mode, n_past, tv):
    """wq | wk | wv + RoPE + KV store, then attention over the cached cells, as two phases of one engine launch (q and the token's cell cross the
    in-launch seam) against pm355_mul_mat_vec_qkv + pm355_attn_cached: same bits in q, in the caches and in the attention output, below and above the
    64-cell short path."""
    torch = P.torch
    rng = np.random.default_rng(31 + n_past + mode)
    E_, H, Hkv, dh, n_ctx = 2048, 16, 4, 128, 256
    ws = [P.upload_weight(Q4_K, rand_blocks(Q4_K, H * dh, E_, rng), E_, H * dh), P.upload_weight(Q4_K, rand_blocks(Q4_K, Hkv * dh, E_, rng), E_, Hkv * dh),
          P.upload_weight(tv, rand_blocks(tv, Hkv * dh, E_, rng), E_, Hkv * dh)]
    x = torch.from_numpy(rng.normal(0, 1.0, (1, E_)).astype(np.float32)).cuda()
    nw = torch.from_numpy((1 + rng.normal(0, 0.05, E_)).astype(np.float32)).cuda()
    pos = torch.tensor([n_past], dtype=torch.int32, device="cuda")
    tab = P.rope_table(pos, dh, mode=mode, freq_base=500000.0)
    kc0 = torch.from_numpy(rng.normal(0, 1, (n_ctx, Hkv * dh)).astype(np.float16)).cuda()
    vc0 = torch.from_numpy(rng.normal(0, 1, (Hkv * dh, n_ctx)).astype(np.float16)).cuda()
    scale = 1.0 / np.sqrt(dh)
    ssx = (x.double() ** 2).float().double().sum().view(1)          # one-partial sum of squares of the input row (f32-rounded squares)
    # launches
    kc1, vc1 = kc0.clone(), vc0.clone()
    q1 = P.mul_mat_vec_qkv(ws, x, tab, pos, kc1, vc1, Hkv, dh, n_ctx, norm_w=nw, eps=1e-5, neox=bool(mode & 2))
    a1 = P.attn_cached(q1, kc1, vc1, pos, H, Hkv, dh, n_ctx, scale)
    # engine
    kc2, vc2 = kc0.clone(), vc0.clone()
    e = P.EngineRun()
    ys = e.matvec(ws, x, norm_w=nw, eps=1e-5, sumsq_in=ssx,
                  qkv=dict(tab=tab, pos=pos, k_cache=kc2, v_cache=vc2, n_head_kv=Hkv, head_dim=dh, n_ctx=n_ctx, neox=bool(mode & 2)))
    a2 = e.attention(ys[0], kc2, vc2, pos, H, Hkv, dh, n_ctx, scale)
    e.run()
    assert torch.equal(ys[0], q1), (ys[0] - q1).abs().max()
    assert torch.equal(kc2, kc1) and torch.equal(vc2, vc1)
    assert torch.equal(a2, a1), (a2 - a1).abs().max()


@pytest.mark.experiments
@pytest.mark.parametrize("shape", [(8192, 64, 8, 128), (4096, 32, 8, 128), (2048, 16, 4, 128), (1024, 16, 8, 64)])
@pytest.mark.parametrize("tv", [Q6_K, Q4_K])
@pytest.mark.parametrize("mode", [0, 2])
def test_attention_in_the_qkv_launch_tail_equals_the_two_launches(P, shape, tv, mode):
    """Round 5: pm355_mul_mat_vec_qkv_attn - the attention over the cached cells computed in the TAIL of the wq | wk | wv launch by
    the last workgroups of each KV-head group (per-group ticket; q and the token's cell are read back through write-through stores / cache-bypassing
    loads) - against pm355_mul_mat_vec_qkv followed by pm355_attn_cached: the same bits in q, in both caches and in the attention output, at the
    Llama-3-70B / 8B head shapes and two small ones, on both sides of the 64-cell short path, the launch repeated on one set of ticket counters
    (they are monotonic across launches), NORM and NEOX rope."""
    torch = P.torch
    E_, H, Hkv, dh = shape
    n_ctx = 1024
    rng = np.random.default_rng(77 + E_ + mode)
    ws = [P.upload_weight(Q4_K, rand_blocks(Q4_K, H * dh, E_, rng), E_, H * dh), P.upload_weight(Q4_K, rand_blocks(Q4_K, Hkv * dh, E_, rng), E_, Hkv * dh),
          P.upload_weight(tv, rand_blocks(tv, Hkv * dh, E_, rng), E_, Hkv * dh)]
    nw = torch.from_numpy((1 + rng.normal(0, 0.05, E_)).astype(np.float32)).cuda()
    kc0 = torch.from_numpy(rng.normal(0, 1, (n_ctx, Hkv * dh)).astype(np.float16)).cuda()
    vc0 = torch.from_numpy(rng.normal(0, 1, (Hkv * dh, n_ctx)).astype(np.float16)).cuda()
    scale = 1.0 / np.sqrt(dh)
    ticket = torch.zeros(Hkv, dtype=torch.int32, device="cuda")
    dog = torch.zeros(1, dtype=torch.int32, device="cuda")
    kc1, vc1, kc2, vc2 = kc0.clone(), vc0.clone(), kc0.clone(), vc0.clone()
    for n_past in (0, 1, 17, 63, 64, 65, 200, 639):
        x = torch.from_numpy(rng.normal(0, 1.0, (1, E_)).astype(np.float32)).cuda()
        pos = torch.tensor([n_past], dtype=torch.int32, device="cuda")
        tab = P.rope_table(pos, dh, mode=mode, freq_base=500000.0)
        q1 = P.mul_mat_vec_qkv(ws, x, tab, pos, kc1, vc1, Hkv, dh, n_ctx, norm_w=nw, eps=1e-5, neox=bool(mode & 2))
        a1 = P.attn_cached(q1, kc1, vc1, pos, H, Hkv, dh, n_ctx, scale, max_keys=648)
        q2, a2 = P.mul_mat_vec_qkv_attn(ws, x, tab, pos, kc2, vc2, H, Hkv, dh, n_ctx, scale, ticket, norm_w=nw, eps=1e-5, neox=bool(mode & 2), max_keys=648, watchdog=dog)
        torch.cuda.synchronize()
        assert int(dog.item()) == 0
        assert torch.equal(q2, q1), (n_past, (q2 - q1).abs().max())
        assert torch.equal(kc2, kc1) and torch.equal(vc2, vc1), n_past
        assert torch.equal(a2, a1), (n_past, (a2 - a1).abs().max())


def _q8_cache(oracle, rng, n_ctx, Ekv):
    f = rng.normal(0, 1, (n_ctx, Ekv)).astype(np.float32)
    blocks = np.concatenate([oracle.quantize_row_q8_0(f[i]) for i in range(n_ctx)])
    deq = np.stack([oracle.dequantize_row(Q8_0, blocks[i * (Ekv // 32 * 34):(i + 1) * (Ekv // 32 * 34)], Ekv) for i in range(n_ctx)])
    return blocks, deq


@pytest.mark.parametrize("kq8,vq8", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("shape", [(64, 8, 128), (32, 8, 128), (16, 4, 64)])
@pytest.mark.parametrize("n_kv", [700, 2500, 8200])
def test_long_context_attention_over_q8_0_caches_on_the_matrix_cores(P, oracle, kq8, vq8, shape, n_kv):
    """Round 5: attn_flash_mfma.hip over Q8_0 K and / or V caches (native 34-byte blocks, row-major V: `-fa -ctk q8_0 -ctv q8_0`,
    src/llama.cpp:3531-3560, :10075-10095) - the dequantizing operand fetch (q_i * d rounded to F16 per value) against float64 attention over the
    dequantized cache values (dequantize_row_q8_0 / F16), 2e-3 of max |out| as for the F16 caches (probabilities are rounded to F16 as MFMA
    operands), at the Llama-3-70B / 8B head shapes and a head_dim-64 one, 700 .. 8200 cells."""
    torch = P.torch
    H, Hkv, dh = shape
    Ekv = Hkv * dh
    n_ctx = ((n_kv + 300) // 256) * 256
    rng = np.random.default_rng(n_kv + H + 2 * kq8 + vq8)
    if kq8:
        kb, kd = _q8_cache(oracle, rng, n_ctx, Ekv)
        kc = torch.from_numpy(kb).cuda()
    else:
        kd = rng.normal(0, 1, (n_ctx, Ekv)).astype(np.float16).astype(np.float32)
        kc = torch.from_numpy(kd.astype(np.float16)).cuda()
    if vq8:
        vb, vd = _q8_cache(oracle, rng, n_ctx, Ekv)
        vc = torch.from_numpy(vb).cuda()
    else:
        vd = rng.normal(0, 1, (n_ctx, Ekv)).astype(np.float16).astype(np.float32)
        vc = torch.from_numpy(vd.astype(np.float16)).cuda()
    q = rng.normal(0, 1, (H, dh)).astype(np.float16).astype(np.float32)
    scale = 1.0 / np.sqrt(dh)
    dyn = torch.tensor([n_kv - 1, n_kv], dtype=torch.int32, device="cuda")
    scratch = P.attn_split_scratch(H, dh, n_ctx)
    grid = 1024
    while grid < n_kv:
        grid *= 2
    flags = P.ATTN_V_ROWMAJOR | (P.ATTN_K_Q8_0 if kq8 else 0) | (P.ATTN_V_Q8_0 if vq8 else 0)
    out = P.attn_cached(torch.from_numpy(q.reshape(1, -1)).cuda(), kc, vc, None, H, Hkv, dh, n_ctx, scale, cell_nkv=dyn, max_keys=min(grid, n_ctx), flags=flags, scratch=scratch)
    got = out.cpu().numpy().reshape(H, dh)
    want = np.zeros((H, dh))
    for h in range(H):
        hk = h // (H // Hkv)
        K = kd[:n_kv, hk * dh:(hk + 1) * dh].astype(np.float64)
        V = vd[:n_kv, hk * dh:(hk + 1) * dh].astype(np.float64)
        s = K @ q[h].astype(np.float64) * scale
        p = np.exp(s - s.max())
        want[h] = (p / p.sum()) @ V
    err = np.abs(got - want).max()
    assert err <= 2e-3 * np.abs(want).max(), (err, np.abs(want).max())


@pytest.mark.parametrize("kq8,vq8", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("mode", [0, 2])
def test_long_context_q8_0_token_step_matches_the_one_workgroup_per_head_kernel(P, oracle, kq8, vq8, mode):
    """pm355_attn_token with Q8_0 caches beyond the long-context threshold: rope + quantizing KV store (attn_q8.hip q8_token_prep_kernel) + the matrix-core
    kernel over the cached cells, against the one-workgroup-per-head kernel that serves short contexts (attn_q8_token_kernel: the reference's
    block-wise integer K.q, f32 P.V): the same bytes in both caches (quantize_row_q8_0_ref of the rotated k / of v), outputs within the F16-operand
    tier of each other and of float64."""
    torch = P.torch
    H, Hkv, dh, n_kv = 32, 8, 128, 1500
    Ekv, n_ctx = Hkv * dh, 2048
    rng = np.random.default_rng(500 + 2 * kq8 + vq8 + mode)
    def cache(q8):
        if q8:
            b, _ = _q8_cache(oracle, rng, n_ctx, Ekv)
            return torch.from_numpy(b).cuda()
        return torch.from_numpy(rng.normal(0, 1, (n_ctx, Ekv)).astype(np.float16)).cuda()
    kc0, vc0 = cache(kq8), cache(vq8)
    q = torch.from_numpy(rng.normal(0, 1, (1, H * dh)).astype(np.float32)).cuda()
    k = torch.from_numpy(rng.normal(0, 1, (1, Ekv)).astype(np.float32)).cuda()
    v = torch.from_numpy(rng.normal(0, 1, (1, Ekv)).astype(np.float32)).cuda()
    pos = torch.tensor([n_kv - 1], dtype=torch.int32, device="cuda")
    dyn = torch.tensor([n_kv - 1, n_kv], dtype=torch.int32, device="cuda")
    flags = P.ATTN_V_ROWMAJOR | (P.ATTN_K_Q8_0 if kq8 else 0) | (P.ATTN_V_Q8_0 if vq8 else 0)
    scale = 1.0 / np.sqrt(dh)
    kc1, vc1, kc2, vc2 = kc0.clone(), vc0.clone(), kc0.clone(), vc0.clone()
    a1 = P.attn_token(q, k, v, kc1, vc1, pos, dyn, H, Hkv, dh, n_ctx, scale, flags=flags, mode=mode, freq_base=500000.0)
    scratch = P.attn_split_scratch(H, dh, n_ctx)
    a2 = P.attn_token(q, k, v, kc2, vc2, pos, dyn, H, Hkv, dh, n_ctx, scale, flags=flags, mode=mode, freq_base=500000.0, scratch=scratch, max_keys=2048)
    torch.cuda.synchronize()
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    d = (a2 - a1).abs().max().item()
    assert d <= 2e-3 * a1.abs().max().item(), (d, a1.abs().max().item())
    nm = float(((a2 - a1).double() ** 2).sum() / (a1.double() ** 2).sum())
    assert nm < 1e-5, nm


def test_small_batch_attention_fallback_is_indifferent_to_the_pre_rounded_query(P):
    """small batches store q F16-rounded (rope + KV store with round_q) for pm355_attn_cached; when that kernel refuses a shape the older
    pm355_attn_decode consumes the same rows. It rounds q to F16 itself (MUL_MAT's src1 conversion, ggml.c:12445-12473), so an already rounded q gives
    the same bits as the raw one - shown here at head_dim 96, a shape attn_cached does not serve."""
    torch = P.torch
    rng = np.random.default_rng(9)
    T, H, Hkv, dh, n_ctx, pos0 = 3, 8, 4, 96, 64, 20
    q = torch.from_numpy(rng.normal(0, 1, (T, H * dh)).astype(np.float32)).cuda()
    kc = torch.from_numpy(rng.normal(0, 1, (n_ctx, Hkv * dh)).astype(np.float16)).cuda().view(torch.int16)
    vc = torch.from_numpy(rng.normal(0, 1, (Hkv * dh, n_ctx)).astype(np.float16)).cuda().view(torch.int16)
    scale = 1.0 / np.sqrt(dh)
    raw = P.attn_decode(q, kc, vc, pos0, H, Hkv, dh, n_ctx, scale)
    rounded = P.attn_decode(q.half().float(), kc, vc, pos0, H, Hkv, dh, n_ctx, scale)
    assert torch.equal(raw, rounded)
