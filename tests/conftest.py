import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are not tracked by git: on a fresh checkout build them first (hipcc cross-compiles gfx950
    without a GPU).  Nothing is built when they are already there."""
    libs = [os.path.join(ROOT, "zkt-plonk_amd", n) for n in ("libzkt_plonk_hip.so", "libzkt_comm_rccl.so")]
    if not all(os.path.exists(lib) for lib in libs):
        import __graft_entry__ as g
        g.build()                                   # the RCCL transport is best effort there: its tests skip without it


def rccl_transport_or_skip():
    lib = os.path.join(ROOT, "zkt-plonk_amd", "libzkt_comm_rccl.so")
    if not os.path.exists(lib):
        pytest.skip("libzkt_comm_rccl.so was not built (no librccl on this host)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "vectors.json")) as f:
        return json.load(f)
