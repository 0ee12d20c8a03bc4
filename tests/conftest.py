import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture
def sw():
    """Pin kernel-form switches of the library for one test (csrc/switches.h: the library reads its CDET_* environment once, at load; inside a
    process the forms are switched through cdet_set_switch). `sw("CDET_HALO_NG", 2)`; `sw("CDET_HALO_NG", None)` = not pinned. Restored afterwards."""
    from cerberusdet_amd import _lib as L

    before = {}

    def pin(name, value):
        if name not in before:
            before[name] = L.get_switch(name)
        L.set_switch(name, value)

    yield pin
    for name, value in before.items():
        L.set_switch(name, value)
