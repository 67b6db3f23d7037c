"""prima_cpp_amd — MI355X-native implementation of prima.cpp's quantized decode hot path.

The product is `libprima_mi355.so` (hand-written HIP for gfx950, C ABI in include/prima_mi355.h) plus
`libggml-mi355.so` (the ggml-backend plug-in built on it). This Python package is only the thin host
binding used by tests/ and bench.py: ctypes over the C ABI, torch for device memory / streams /
torch.distributed. There is NO CPU fallback: every op raises if the HIP library is missing.
"""
from .lib import (F16, F32, Q4_K, Q5_K, Q6_K, Q8_0, Q8_K, PM355Error, lib_path, load, row_size)  # noqa: F401
