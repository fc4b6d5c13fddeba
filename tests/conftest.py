import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_C
    oracle_C.lib()
    return oracle_C


@pytest.fixture(scope="session")
def hip():
    """The product `_C` surface; the HIP library must be built (no fallback)."""
    import torch
    from online_lang_splatting_amd import build
    build.build()
    from online_lang_splatting_amd import _C, _lib
    _lib.lib()
    assert torch.cuda.is_available(), "GPU tests need cuda:0"
    return _C
