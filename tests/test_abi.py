"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that include/*.h
declares (no compute calls without a GPU), and the product never reaches for the oracle."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"PM355_API\s+[\w\s\*]+?\b(pm355_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import prima_cpp_amd
    lib = prima_cpp_amd.load()
    names = _declared("prima_mi355.h")
    assert len(names) > 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.pm355_version().startswith(b"prima_mi355")
    lib.pm355_row_size.restype = C.c_size_t
    assert lib.pm355_row_size(12, 8192) == 4608 and lib.pm355_row_size(14, 8192) == 6720
    assert lib.pm355_row_stride(14, 8192) == 6720 and lib.pm355_row_stride(14, 768) % 16 == 0


def test_small_batch_matmul_shape_check_is_host_logic():
    """pm355_mul_mat_q_small_check decides, without touching a device, which (type, K, N, n_tokens) the integer-matrix-core mat-mul serves:
    the engine's per-layer path choice and the plug-in's MUL_MAT dispatch both rest on it."""
    import prima_cpp_amd
    lib = prima_cpp_amd.load()
    chk = lib.pm355_mul_mat_q_small_check
    chk.restype = C.c_int
    chk.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64]
    Q8_0, Q4_K, Q5_K, Q6_K = 8, 12, 13, 14
    for t in (Q4_K, Q5_K, Q6_K):
        for K, N in ((8192, 28672), (28672, 8192), (8192, 1024), (4096, 128256), (768, 70)):
            for T in (1, 8, 16, 32, 33, 64):
                assert chk(t, K, N, T) == 0, (t, K, N, T)
    # Q8_0 weights pair with Q8_0 activations (32-value blocks): served since round 4, K a multiple of 32 (Qwen2.5-72B's ffn_down: 29568)
    assert chk(Q8_0, 8192, 8192, 8) == 0 and chk(Q8_0, 29568, 8192, 33) == 0 and chk(Q8_0, 29568 + 16, 8192, 8) != 0 and chk(Q8_0, 256, 64, 8) != 0
    assert chk(Q4_K, 8192, 8192, 65) != 0 and chk(Q4_K, 8192, 8192, 0) != 0
    assert chk(Q4_K, 256, 8192, 8) != 0             # K < 512
    assert chk(Q4_K, 8192 + 64, 8192, 8) != 0       # K % 256
    assert chk(Q6_K, 65536, 8192, 8) != 0           # scale table + 8 wave tiles exceed the LDS budget


def test_plugin_header_declares_the_reference_entry_points():
    src = open(os.path.join(ROOT, "include", "ggml_backend_mi355.h")).read()
    for name in ("ggml_backend_mi355_reg", "ggml_backend_mi355_init", "ggml_backend_mi355_buffer_type",
                 "ggml_backend_mi355_host_buffer_type", "ggml_backend_is_mi355", "ggml_backend_mi355_get_device_count"):
        assert name in src, name
    plug = os.path.join(ROOT, "prima_cpp_amd", "libggml-mi355.so")
    if os.path.exists(plug):
        out = os.popen(f"nm -D --defined-only {plug}").read()
        for name in ("ggml_backend_mi355_reg", "ggml_backend_mi355_init"):
            assert name in out, name


def test_product_never_touches_the_oracle():
    """A product path that routes through oracle/ (or any CPU fallback) would void every parity claim."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "prima_cpp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"liboracle|ggml_oracle|oracle/|libggml_ref|_ref/", txt) and f not in ("engine.py",):
                    bad.append(os.path.join(dp, f))
                if f == "engine.py":       # only its smoke() helper may take an oracle OBJECT handed in by the caller
                    assert "liboracle" not in txt and "ggml_oracle" not in txt
    assert not bad, bad


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import prima_cpp_amd.lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "lib_path", lambda: str(tmp_path / "nope.so"))
    try:
        L.load()
        assert False, "load() must raise when the HIP library is missing"
    except L.PM355Error as e:
        assert "no CPU fallback" in str(e)
    finally:
        monkeypatch.undo()
        L._lib = None
        L.load()
