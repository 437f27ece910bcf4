"""The reference's surface in compiled code.

include/VariableSamplingMPC.hpp holds ONE implementation of the packer + tick state machine (TickMachine) and the
reference-signature class template VariableSamplingMPCT (configure(parametersHandler, qpInput) / update(qpInput) /
solveMPC / getters, MPCPyBindings.cpp:22-90).  It is driven here three ways through the SAME provider states
(tests/fake_provider.py): as C++ against fake classes with the members of Robot.h / QPInput.h / IParametersHandler
(tests/cpp/reference_surface_driver.cpp), through the pybind module bindingsMPC with Python provider objects, and -- the
checker -- the Python twin reference_api, whose records tests/test_gpu_reference_api.py pins against the model written
from the reference's plugins (tests/tick_model.py).  Variants: getRobot() != getRobotReference(), controlled joints
selected by NAME at other positions than 3..10, jointsLambdaOption 'constant', a non-zero RPY / RPYDot track."""
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

import fake_provider as fp
from conftest import PKG, ROOT, relerr

pytestmark = pytest.mark.gpu
PKG_DIR = os.path.join(ROOT, PKG)
DRV = os.path.join(ROOT, "tests", "cpp", "reference_surface_driver.cpp")
N_TICKS = 45


@pytest.fixture(scope="module")
def driver(tmp_path_factory, solver_mod):
    exe = str(tmp_path_factory.mktemp("refsurf") / "reference_surface_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), DRV, "-o", exe,
           "-L", PKG_DIR, "-lvsmpc", f"-Wl,-rpath,{PKG_DIR}", "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


@pytest.fixture(scope="module")
def pinned():
    consts = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))
    traj = dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_trajectories.npz")))
    return consts, traj


def params_of(consts, traj, selector, constant_lambda=False):
    p = dict(consts["VS_MPC_CONFIG"])
    p["controlledJoints"] = [f"joint_{i}" for i in selector]
    p["jointsLambdaOption"] = "constant" if constant_lambda else "unfiltered"
    p["TRAJECTORY_MANAGER"] = {"alphaGravity": traj["alphaGravity"], "fps": int(traj["alphaGravity_fps"][0])}
    p["POSITION_TRAJECTORY"] = {"positionCoM": traj["positionCoM"], "velocityCoM": traj["velocityCoM"], "RPY": traj["RPY"],
                                "RPYDot": traj["RPYDot"], "fps": int(traj["trajectory_fps"][0])}
    return p


def drive(mpc, qp, sc, get_record, n_in):
    """the harness loop of src/variable_sampling_mpc.py:106-135 over the scenario's provider states"""
    rows = []
    for k in range(1, sc.n_ticks + 1):
        qp.setEstimatedThrustDot(sc.load(k))
        assert mpc.update(qp)
        rec = np.array(get_record(mpc), dtype=float)
        assert rec.shape == (n_in,)
        assert mpc.solveMPC()
        rows.append(np.concatenate([rec, [mpc.getQPProblemStatus()], mpc.getThrustReference(), mpc.getThrustDotReference(),
                                    mpc.getThrottleReference(), mpc.getJointsReferencePosition(), qp.getPosCoMReference(),
                                    qp.getRPYReference(), [qp.getAlphaGravity()], qp.getMomentumReference()]))
        qp.setThrottleMPC(mpc.getThrottleReference())
        qp.setThrustDesMPC(mpc.getThrustReference())
        qp.setThrustDotDesMPC(mpc.getThrustDotReference())
        qp.setOutputQPJointsPosition(mpc.getJointsReferencePosition())
    sc.load(0)
    return np.array(rows)


def new_qp(api, sc):
    sc.load(0)
    qp = api.QPInput(sc.robot, sc.reference)
    q = sc.initial_qp
    qp.setThrottleMPC(q["throttle"]); qp.setThrustDesMPC(q["thrustDes"]); qp.setThrustDotDesMPC(q["thrustDotDes"])
    qp.setEstimatedThrustDot(q["estTd"]); qp.setOutputQPJointsPosition(q["joints"])
    return qp


@pytest.mark.parametrize("variant", ["shipped", "distinct-robots", "named-joints", "constant-lambda", "rpy-track"])
def test_compiled_surface_matches_the_python_twin(driver, pinned, variant, tmp_path, layout):
    consts, traj = pinned
    traj = dict(traj)
    selector = [3, 4, 5, 6, 7, 8, 9, 10]
    distinct, constant = False, False
    if variant == "distinct-robots":
        distinct = True
    if variant == "named-joints":
        selector = [5, 3, 12, 6, 7, 20, 9, 0]
    if variant == "constant-lambda":
        constant, distinct = True, True
    if variant == "rpy-track":
        n = len(traj["RPY"])
        t = np.arange(n)[:, None] / 10.0
        traj["RPY"] = 0.05 * np.sin(0.3 * t) * np.array([[1.0, -0.5, 0.2]])
        traj["RPYDot"] = 0.015 * np.cos(0.3 * t) * np.array([[1.0, -0.5, 0.2]])
    sc = fp.Scenario(n_ticks=N_TICKS, seed=17, distinct=distinct)
    params = params_of(consts, traj, selector, constant)
    n_in = layout.paper_config().n_in

    # --- the C++ template against fake C++ providers
    (tmp_path / "scenario.bin").write_bytes(sc.serialise(consts, traj, selector, constant_lambda=constant).tobytes())
    res = subprocess.run([driver, str(tmp_path / "scenario.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    cpp = np.frombuffer((tmp_path / "out.bin").read_bytes(), dtype=np.float64).reshape(N_TICKS, -1)

    # --- the Python twin (pinned against tests/tick_model.py by test_gpu_reference_api.py)
    api = importlib.import_module(PKG + ".reference_api")
    twin = api.VariableSamplingMPC()
    qp = new_qp(api, sc)
    assert twin.configure(params, qp)
    py_rows = drive(twin, qp, sc, lambda m: m._record, n_in)

    # --- the pybind module with Python provider objects
    bindings = importlib.import_module(PKG + ".bindingsMPC")
    shim = bindings.VariableSamplingMPC()
    qp2 = new_qp(api, sc)
    assert shim.configure(params, qp2)
    pb_rows = drive(shim, qp2, sc, lambda m: m.getRecord(), n_in)

    assert cpp.shape == py_rows.shape == pb_rows.shape
    assert (cpp[:, n_in] == layout.STATUS_SOLVED).all()
    # same code, same provider values, same library: the compiled front ends agree bit for bit
    np.testing.assert_array_equal(cpp, pb_rows)
    # the Python twin computes the same quantities in numpy (rotation products, up-sampling): equal to rounding
    assert relerr(cpp[:, :n_in], py_rows[:, :n_in]) < 1e-12
    assert relerr(cpp[:, n_in:], py_rows[:, n_in:]) < 1e-9
    holds = cpp[:, layout.IN_HOLD]
    np.testing.assert_array_equal(holds, [0.0 if k % 20 == 19 else 1.0 for k in range(N_TICKS)])
    if variant == "named-joints":
        # the accumulator moved the NAMED joints (variableSamplingMPC.cpp:104-108), nothing else
        q0 = sc.initial_qp["joints"]
        moved = np.abs(cpp[-1, n_in + 13:n_in + 36] - q0) > 0
        assert set(np.nonzero(moved)[0]) == set(selector)
    if variant == "rpy-track":
        assert np.abs(cpp[:, layout.IN_XREF + 9:layout.IN_XREF + 12]).max() > 0      # h_ang reference = I_G W RPYDot


def test_compiled_surface_reports_a_missing_key(driver, pinned, tmp_path):
    consts, traj = pinned
    sc = fp.Scenario(n_ticks=1, seed=3)
    (tmp_path / "scenario.bin").write_bytes(sc.serialise(consts, traj, list(range(3, 11)), drop_key=True).tobytes())
    res = subprocess.run([driver, str(tmp_path / "scenario.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert res.returncode == 10 and "weightThrottle" in res.stdout      # configure -> false, key named like the reference's yError
    bindings = importlib.import_module(PKG + ".bindingsMPC")
    api = importlib.import_module(PKG + ".reference_api")
    bad = params_of(consts, traj, list(range(3, 11)))
    del bad["weightThrottle"]
    assert not bindings.VariableSamplingMPC().configure(bad, new_qp(api, sc))


def test_blf_style_python_handler(pinned):
    """BLF's Python parameters handler spells its getters get_parameter_int / _float / ... and get_group
    (src/variable_sampling_mpc.py:37-40): both Python front ends accept it."""
    consts, traj = pinned
    flat = params_of(consts, traj, list(range(3, 11)))

    class Handler:
        def __init__(self, d): self.d = d
        def _get(self, k):
            if k not in self.d: raise ValueError(k)
            return self.d[k]
        def get_parameter_int(self, k): return int(self._get(k))
        def get_parameter_float(self, k): return float(self._get(k))
        def get_parameter_bool(self, k): return bool(self._get(k))
        def get_parameter_string(self, k): return str(self._get(k))
        def get_parameter_vector_float(self, k): return [float(v) for v in self._get(k)]
        def get_parameter_vector_string(self, k): return [str(v) for v in self._get(k)]
        def get_group(self, k):
            g = self._get(k)
            return g if k in ("TRAJECTORY_MANAGER", "POSITION_TRAJECTORY") else Handler(g)

    top = Handler({"VS_MPC_CONFIG": flat})
    api = importlib.import_module(PKG + ".reference_api")
    bindings = importlib.import_module(PKG + ".bindingsMPC")
    sc = fp.Scenario(n_ticks=2, seed=5)
    for front in (api.VariableSamplingMPC(), bindings.VariableSamplingMPC()):
        qp = new_qp(api, sc)
        assert front.configure(top.get_group("VS_MPC_CONFIG"), qp)          # as the harness passes it (:37,70)
        qp.setEstimatedThrustDot(sc.load(1))
        assert front.update(qp) and front.solveMPC() and front.getQPProblemStatus() == 1
