"""ctypes binding of libprima_mi355.so (C ABI: include/prima_mi355.h).

`import torch` happens BEFORE the library is loaded so that both resolve the same HIP runtime
(libamdhip64.so.7) and torch streams / device pointers can be handed straight to the C ABI.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below: shared HIP runtime)

F32, F16, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 8, 12, 13, 14, 15
_BLOCK = {Q8_0: (32, 34), Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210), Q8_K: (256, 292), F16: (1, 2), F32: (1, 4)}

HERE = os.path.dirname(os.path.abspath(__file__))


class PM355Error(RuntimeError):
    pass


def lib_path():
    """In-tree library; PM355_LIB selects another build of the same C ABI (A/B kernel experiments)."""
    return os.environ.get("PM355_LIB") or os.path.join(HERE, "libprima_mi355.so")


_lib = None
_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t

_SIGS = {
    "pm355_version": (C.c_char_p, []),
    "pm355_last_error": (C.c_char_p, []),
    "pm355_device_count": (_i32, []),
    "pm355_set_device": (_i32, [_i32]),
    "pm355_sync": (_i32, [_vp]),
    "pm355_memcpy_d2d": (_i32, [_vp, _vp, _sz, _vp]),
    "pm355_memcpy_h2d": (_i32, [_vp, _vp, _sz, _vp]),
    "pm355_memcpy_d2h": (_i32, [_vp, _vp, _sz, _vp]),
    "pm355_row_size": (_sz, [_i32, _i64]),
    "pm355_row_stride": (_sz, [_i32, _i64]),
    "pm355_q8_K_row_size": (_sz, [_i64]),
    "pm355_q8_0_row_size": (_sz, [_i64]),
    "pm355_repack_rows": (_i32, [_i32, _vp, _vp, _i64, _i64, _i32, _vp]),
    "pm355_uploader_new": (_vp, [_sz, _i32]),
    "pm355_uploader_free": (None, [_vp]),
    "pm355_upload": (_i32, [_vp, _i32, _i64, _vp, _vp, _sz, _i32]),
    "pm355_uploader_sync": (_i32, [_vp]),
    "pm355_uploader_bytes": (C.c_uint64, [_vp]),
    "pm355_quantize_q8_K": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "pm355_quantize_q8_0": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "pm355_act_to_ggml_blocks": (_i32, [_i32, _vp, _vp, _i64, _i64, _vp]),
    "pm355_rms_norm": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _vp]),
    "pm355_mul_mat_vec_q": (_i32, [_i32, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _i64, _vp, _vp, _vp]),
    "pm355_get_rows": (_i32, [_i32, _vp, _i64, _vp, _i32, _vp, _vp]),
    "pm355_rope_kv_store": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pm355_attn_decode": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "pm355_attn_rope_fused": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "pm355_argmax": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "pm355_silu_mul": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "pm355_add": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "pm355_mul_mat_vec_q_dbg": (_i32, [_i32, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
}


def load():
    """Load the HIP library. Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise PM355Error(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            if hasattr(_lib, name):
                f = getattr(_lib, name)
                f.restype, f.argtypes = res, args
    return _lib


_probe = None


def load_probe():
    """libprima_mi355_probe.so: measurement helpers (tools/csrc), kept out of the product library."""
    global _probe
    if _probe is None:
        path = os.path.join(HERE, "libprima_mi355_probe.so")
        if not os.path.exists(path):
            raise PM355Error(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _probe = C.CDLL(path)
        _probe.pm355_probe_stream_read.restype = C.c_int
        _probe.pm355_probe_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return _probe


def get():
    return load()


def declare(name, restype, argtypes):
    f = getattr(load(), name)
    f.restype, f.argtypes = restype, argtypes
    return f


def check(rc, what=""):
    if rc != 0:
        raise PM355Error(f"{what} failed rc={rc}: {load().pm355_last_error().decode()}")


def row_size(t, k):
    n, b = _BLOCK[t]
    assert k % n == 0
    return k // n * b


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()
