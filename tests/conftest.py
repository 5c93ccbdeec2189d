import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "cdmft-lanc-ed_amd"))
sys.path.insert(0, str(ROOT / "scripts"))  # harness.py: the callers' call order (measurement scaffolding, not the package)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as ge

    ge.build_engine()
    return ge
