import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure; oracle/capital_oracle.c)."""
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hip():
    """One capi handle for the whole GPU session (fails loudly without GPU / built library)."""
    import torch
    from capital_amd import capi
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    h = capi.Handle(0)
    yield h
    h.close()
