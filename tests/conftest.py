import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _host_independent_oracle_matmul(request, monkeypatch):
    """The GPU suite runs on another host than the one the fixtures were made on: torch's native CPU bf16 matmul is host
    dependent there (oracle/sae_oracle.py, MATMUL_MODE / NORM_MODE), so the oracle evaluates bf16 matmuls by their
    definition and takes gradient norms in float64."""
    if request.node.get_closest_marker("gpu") is not None:
        from oracle import sae_oracle as O
        monkeypatch.setattr(O, "MATMUL_MODE", "fp32")
        monkeypatch.setattr(O, "NORM_MODE", "float64")
    yield

