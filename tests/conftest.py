"""pytest configuration: the `gpu` marker separates MI355X parity tests from the CPU suite."""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def lib():
    """libgprx.so through the C ABI; GPU tests fail loudly (no skip) when it is missing on a GPU box."""
    from gpras_amd import _lib

    return _lib.load()


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU device in this container (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _host_logic_kmeans(monkeypatch):
    """CPU container only: tests of the host logic replace the HIP engine by an oracle-backed stand-in (test_host_logic.py);
    the device Lloyd iterations of the k-means initialisation get the oracle's restatement the same way.  On a GPU box
    nothing is patched: every test runs the real device path."""
    if _gpu_present():
        return
    from gpras_amd import gpr
    from oracle import kmeans as okm

    monkeypatch.setattr(gpr, "kmeans_centers", lambda x, m, device=0: okm.kmeans_centers(x, m)[0])
