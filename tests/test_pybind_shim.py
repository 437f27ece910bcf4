"""bindingsMPC: the pybind11 shim with the reference's Python class/method names (MPCPyBindings.cpp:22-90)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"

PARAMS = {  # src/config/vs_mcp_config.xml:7-43
    "useJetDynamic": True, "periodMPC": 0.005, "periodMPCLargeSteps": 0.1, "periodMPCSmallSteps": 0.005,
    "nIter": 17, "nIterSmall": 7, "controlHorizon": 12,
    "weightCoMPos": [500.0, 500.0, 5000.0], "weightCoMPosError": [25000.0, 25000.0, 50000.0],
    "weightLinMom": [1.0, 1.0, 1.5], "weightRPY": [1000.0] * 3, "weightRPYError": [10000.0] * 3,
    "weightAngMom": [80.0] * 3, "weightDeltaJoint": [65000.0] * 8, "weightThrottle": 80000.0,
    "weightInitialThrottle": 80000.0, "weightRegularizationJointPos": 20.0, "throttleMin": 0.0, "throttleMax": 100.0,
}


@pytest.fixture(scope="module")
def shim(solver_mod):
    build = importlib.import_module(PKG + ".build")
    path = build.build_bindings()
    if path is None:
        pytest.skip("pybind11 not available")
    import torch  # noqa: F401  (first: libvsmpc.so must bind to the HIP runtime torch brings along, see _lib.py)
    return importlib.import_module(PKG + ".bindingsMPC")


def test_shim_surface(shim):
    names = ["configure", "update", "solveMPC", "getMPCSolution", "getJointsReferencePosition", "getThrottleReference",
             "getThrustReference", "getThrustDotReference", "getFinalCoMPosition", "getFinalLinMom", "getFinalRPY",
             "getFinalAngMom", "getNStatesMPC", "getNInputMPC"]
    for n in names:
        assert hasattr(shim.VariableSamplingMPC, n), n
    bad = dict(PARAMS)
    del bad["nIter"]
    with pytest.raises(Exception) as e:
        shim.VariableSamplingMPC().configure(bad, np.zeros(23), np.zeros(3))
    assert "nIter" in str(e.value)                         # "Parameter 'nIter' not found" like the reference's yError


@pytest.mark.gpu
def test_shim_runs_a_tick(shim, solver_mod, synth, layout):
    cfg = layout.paper_config()
    rec = synth.make_batch(cfg, 3, workload="takeoff")
    mpc = shim.VariableSamplingMPC()
    assert mpc.configure(PARAMS, np.zeros(23), np.zeros(3))
    assert mpc.getNStatesMPC() == 26.0 and mpc.getNInputMPC() == 12.0
    ref = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=3)
    x, fm, st, it = ref.solve(rec)
    q = np.zeros(23)
    for k in range(3):
        assert mpc.update(rec[k]) and mpc.solveMPC()
        assert mpc.getQPProblemStatus() == st[k] == layout.STATUS_SOLVED
        q[3:11] += fm[k, 0:8]
        np.testing.assert_array_equal(mpc.getJointsReferencePosition(), q)
        np.testing.assert_array_equal(mpc.getThrottleReference(), fm[k, 12:16])
        np.testing.assert_array_equal(mpc.getThrustReference(), fm[k, 16:20])
        np.testing.assert_array_equal(mpc.getThrustDotReference(), fm[k, 20:24])
        np.testing.assert_array_equal(mpc.getMPCSolution(), x[k, 468:])
        np.testing.assert_array_equal(mpc.getFinalCoMPosition(), x[k, 442:445])
    assert not mpc.update(rec[0][:100])
    ref.close()
