"""Round 5's measured-slower forms of the decode layer - producer-side sums of squares, attention in the tail of the QKV launch, the persistent engine
(decode_engine.hip) - are no longer part of what prima.cpp links: they are compiled into prima_cpp_amd/libprima_mi355_exp.so (-DPM_EXPERIMENTS=1, build.py
build_experiments()). Their tests (marked `experiments`) run here, in a process of their own with that library loaded; and the product library must refuse the
entry points loudly instead of quietly doing something else."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "prima_cpp_amd", "libprima_mi355_exp.so")
pytestmark = pytest.mark.gpu


def test_product_library_refuses_the_experiment_entry_points():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import ctypes as C
    import prima_cpp_amd.lib as L
    lib = L.load()
    lib.pm355_experiments_built.restype = C.c_int
    if os.environ.get("PM355_LIB"):
        pytest.skip("PM355_LIB selects another build")
    assert lib.pm355_experiments_built() == 0
    lib.pm355_engine_run.restype = C.c_int
    lib.pm355_engine_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    assert lib.pm355_engine_run(None, 0, None) != 0           # PM355_E_UNSUPPORTED: no code behind it in the product library


def test_round5_experiments_pass_against_their_own_library():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    assert os.path.exists(EXP), "prima_cpp_amd/libprima_mi355_exp.so not built (__graft_entry__.build() builds it)"
    env = dict(os.environ, PM355_LIB=EXP, PM355_EXPERIMENTS_ACTIVE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), os.path.join(ROOT, "tests", "test_gpu_engine.py"),
                        "-q", "-x", "-m", "gpu and experiments", "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1500:]
    print(tail)
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail
