import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG + ".synth")


@pytest.fixture(scope="session")
def layout():
    return importlib.import_module(PKG + ".layout")


@pytest.fixture(scope="session")
def ref():
    import vsmpc_ref
    return vsmpc_ref


@pytest.fixture(scope="session")
def golden_paper():
    return np.load(os.path.join(ROOT, "tests", "golden", "vsmpc_golden_paper.npz"))


@pytest.fixture(scope="session")
def golden_h2x():
    return np.load(os.path.join(ROOT, "tests", "golden", "vsmpc_golden_horizon2x.npz"))


@pytest.fixture(scope="session")
def solver_mod():
    import __graft_entry__ as ge
    ge.build()
    return importlib.import_module(PKG + ".solver")


def relerr(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, float(np.abs(b).max())))
