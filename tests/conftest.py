import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """native pieces are built once per session (hipcc cross-compiles without a GPU)"""
    # On a GPU box: PyTorch's wheel bundles its own HIP runtime (no SONAME), so a process that uses torch tensors AND
    # libcannoles_hip.so holds two runtimes.  They coexist when torch's initialises first (what bench.py does); the other
    # order can leave the second one without a device.  Initialise torch's before the library is loaded.
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def params(built):
    from oracle import oracle as O
    return O.default_params()
