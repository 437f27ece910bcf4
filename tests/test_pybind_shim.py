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
    names = ["configure", "configureRecord", "setTrajectoryLoader", "update", "solveMPC", "getMPCSolution", "getJointsReferencePosition", "getThrottleReference",
             "getThrustReference", "getThrustDotReference", "getFinalCoMPosition", "getFinalLinMom", "getFinalRPY",
             "getFinalAngMom", "getNStatesMPC", "getNInputMPC"]
    for n in names:
        assert hasattr(shim.VariableSamplingMPC, n), n
    bad = dict(PARAMS)
    del bad["nIter"]
    with pytest.raises(Exception) as e:
        shim.VariableSamplingMPC().configureRecord(bad, np.zeros(23), np.zeros(3))
    assert "nIter" in str(e.value)                         # "Parameter 'nIter' not found" like the reference's yError


def test_reference_import_name_resolves(shim):
    """src/variable_sampling_mpc.py:5 -- `from momentum_based_mpc.bindingsMPC import VariableSamplingMPC` -- with the
    repository root on sys.path."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    mod = importlib.import_module("momentum_based_mpc.bindingsMPC")
    assert mod.VariableSamplingMPC is shim.VariableSamplingMPC


def test_configure_with_a_dict_handler_and_a_device_reaches_the_reference_overload(shim):
    """configure(parametersHandler, mpcInput, 0) -- the INTEGRATION.md call -- must reach the reference-signature entry
    also when the handler is a mapping (round-3 advisor: it used to land in the record-level overload and raise)."""
    class NoRobot:   # a QPInput-like object whose robot lookup fails: the reference entry returns False, it does not raise
        def getRobot(self): raise RuntimeError("no robot")
    params = dict(PARAMS)
    del params["nIter"]
    assert shim.VariableSamplingMPC().configure(params, NoRobot(), 0) is False     # 'nIter' missing -> false, like yError


def test_trajectory_file_groups_go_through_the_loader(shim, tmp_path):
    """Groups as the harness' handler holds them (vs_mcp_config.xml:34-40: a trajectoryFile name) are resolved through the
    loader callable; without one configure() fails with a message, it does not guess."""
    import scipy.io
    apkg = importlib.import_module(PKG + ".trajectory_io")
    n = 30
    pos = np.linspace(0, 1, 3 * n).reshape(3, n)                      # MAT layout: dim x samples
    scipy.io.savemat(tmp_path / "pos.mat", {"positionCoM": pos, "velocityCoM": 0 * pos, "RPY": 0 * pos, "RPYDot": 0 * pos, "fps": 10.0})
    scipy.io.savemat(tmp_path / "alpha.mat", {"alphaGravity": np.linspace(0.1, 1, n)[None, :], "fps": 10.0})
    got = apkg.load_mat_trajectory(str(tmp_path / "pos.mat"))
    assert got["fps"] == 10 and got["positionCoM"].shape == (n, 3)
    np.testing.assert_array_equal(got["positionCoM"], pos.T)
    a = apkg.load_mat_trajectory(str(tmp_path / "alpha.mat"))
    assert a["alphaGravity"].shape == (n, 1)
    seen = []
    params = dict(PARAMS, useEstimatedThrust=True, jointsLambdaOption="unfiltered", controlledJoints=[f"joint_{i}" for i in range(3, 11)],
                  POSITION_TRAJECTORY={"trajectoryFile": str(tmp_path / "pos.mat")},
                  TRAJECTORY_MANAGER={"trajectoryFile": str(tmp_path / "alpha.mat")})

    class NoRobot:
        def getRobot(self): raise RuntimeError("stop after the trajectories")
    mpc = shim.VariableSamplingMPC()
    assert mpc.configure(params, NoRobot()) is False               # no loader installed on this object / module -> refused
    mpc.setTrajectoryLoader(lambda f: (seen.append(f), apkg.load_mat_trajectory(f))[1])
    assert mpc.configure(params, NoRobot()) is False               # (fails later, at the robot) ...
    assert seen == [str(tmp_path / "pos.mat"), str(tmp_path / "alpha.mat")]   # ... after both files went through the loader


@pytest.mark.gpu
def test_shim_runs_a_tick(shim, solver_mod, synth, layout):
    cfg = layout.paper_config()
    rec = synth.make_batch(cfg, 3, workload="takeoff")
    mpc = shim.VariableSamplingMPC()
    assert mpc.configureRecord(PARAMS, np.zeros(23), np.zeros(3))
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
