import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
