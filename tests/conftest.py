import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def op():
    """The test-only CPU oracle (builds it if the shared object is missing)."""
    import oracle_py
    if not os.path.exists(oracle_py.ORACLE_SO):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return oracle_py


@pytest.fixture(scope="session")
def scenes():
    return importlib.import_module("ray-tracing_amd.scenes")


@pytest.fixture(scope="session")
def capi():
    return importlib.import_module("ray-tracing_amd.capi")


@pytest.fixture(scope="session")
def rtx():
    return importlib.import_module("ray-tracing_amd")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
