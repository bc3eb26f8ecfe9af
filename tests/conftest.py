import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The multi-process GPU tests run FIRST, while the pytest process itself holds nothing on the card: measured on MI355X (profiles/r4_eight_thread_ranks_2x2x2.txt),
    the eight-rank rehearsals take 11 / 30 s then, 75 / 62 s after the single-GPU tests have left this process with its session handle, streams and workspace, and
    116 / 161 s after the full-size tests have grown that workspace -- the ranks' processes share the GPU with whatever their parent keeps on it."""
    first = [it for it in items if "test_gpu_multirank.py" in it.nodeid]
    if first:
        rest = [it for it in items if "test_gpu_multirank.py" not in it.nodeid]
        items[:] = first + rest


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
