import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: exercises round 5's measured-slower forms of the decode layer, which live in prima_cpp_amd/libprima_mi355_exp.so only "
                                       "(-DPM_EXPERIMENTS=1); tests/test_gpu_experiments.py runs these tests in a process of their own with that library loaded")


def pytest_collection_modifyitems(config, items):
    # the product library carries no experiment code: the marked tests run only where PM355_LIB points at the experiments library
    if os.environ.get("PM355_EXPERIMENTS_ACTIVE") == "1":
        return
    skip = pytest.mark.skip(reason="experiments library only: run by tests/test_gpu_experiments.py (PM355_LIB=libprima_mi355_exp.so)")
    for it in items:
        if "experiments" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from _bind import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref_scalar():
    from _bind import Ref, have_ref
    if not have_ref("scalar"):
        pytest.skip("oracle/_ref/libggml_ref_scalar.so not built (needs /root/reference)")
    return Ref("scalar")


@pytest.fixture(scope="session")
def ref_avx2():
    from _bind import Ref, have_ref
    if not have_ref("avx2"):
        pytest.skip("oracle/_ref/libggml_ref_avx2.so not built (needs /root/reference)")
    return Ref("avx2")
