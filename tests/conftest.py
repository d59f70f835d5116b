import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        manifest = json.load(f)
    npz_path = os.path.join(GOLDEN, name + ".npz")
    arrays = dict(np.load(npz_path)) if os.path.exists(npz_path) else {}
    return manifest, arrays


@pytest.fixture(scope="session")
def golden_small():
    return _load("small_cases")


@pytest.fixture(scope="session")
def golden_demo():
    return _load("demo_cases")


@pytest.fixture(scope="session")
def golden_trace():
    return _load("trace_cases")


@pytest.fixture(scope="session")
def golden_long():
    return _load("long_cases")


@pytest.fixture(scope="session")
def golden_xlong():
    return _load("xlong_cases")


@pytest.fixture(scope="session")
def golden_large():
    return _load("large_cases")[0]


@pytest.fixture(scope="session")
def golden_matching():
    return _load("matching_cases")


@pytest.fixture(scope="session")
def built_lib():
    """The loaded library for host-only entry points (no GPU needed); built in-tree if stale."""
    from sslap_amd import _lib, build
    import shutil
    if build.stale() and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        build.build()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu_lib():
    """The loaded HIP library; fails (not skips) when it is missing -- gpu tests must run native code."""
    from sslap_amd import _lib, build
    import shutil
    if build.stale() and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        build.build()  # in-tree build of the product library (never a fallback: without it the tests fail)
    return _lib.load()
