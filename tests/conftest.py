import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "stage(n): run order across modules -- lower stages first, unmarked tests are stage 0 (stable within a stage)")


def pytest_collection_modifyitems(config, items):
    """the order of the device suite is said HERE, not by file names: modules may carry `pytest.mark.stage(n)`"""
    def stage(item):
        m = item.get_closest_marker("stage")
        return int(m.args[0]) if m and m.args else 0
    items.sort(key=stage)                                           # (list.sort is stable: collection order inside a stage)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine_lib():
    """libsummerset_hip.so; built on demand with hipcc (cross-compiles without a GPU)."""
    from summerset_amd import build as B
    B.build()
    from summerset_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no device is visible (tests marked gpu must run on the GPU box)")
    return torch.device("cuda:0")
