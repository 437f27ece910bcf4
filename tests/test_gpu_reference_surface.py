"""The reference's surface in compiled code.

include/VariableSamplingMPC.hpp holds ONE implementation of the packer + tick state machine (TickMachine) and the
reference-signature class template VariableSamplingMPCT (configure(parametersHandler, qpInput) / update(qpInput) /
solveMPC / getters, MPCPyBindings.cpp:22-90).  tests/cpp/reference_surface_driver.cpp instantiates it as
VariableSamplingMPCT<QPInput, TrajectoryManager> over SIGNATURE-EXACT stand-ins of the reference's headers
(tests/cpp/refstub/) through the INTEGRATION.md snippet verbatim, and is checked here
  * directly against the model written from the reference's plugins (tests/tick_model.py) with the oracle's kinematics
    terms (test_compiled_surface_matches_the_reference_tick_model: the shipped configuration);
  * against the pybind module bindingsMPC and the Python twin reference_api driven through the SAME provider states
    (tests/fake_provider.py) in the variants the tick model does not cover: getRobot() != getRobotReference(), controlled
    joints selected by NAME at other positions than 3..10, jointsLambdaOption 'constant', a non-zero RPY / RPYDot track
    (the twin itself is pinned against the tick model by tests/test_gpu_reference_api.py)."""
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

import fake_provider as fp
import tick_model as tm
from conftest import PKG, ROOT, relerr

pytestmark = pytest.mark.gpu
PKG_DIR = os.path.join(ROOT, PKG)
DRV = os.path.join(ROOT, "tests", "cpp", "reference_surface_driver.cpp")
N_TICKS = 45


@pytest.fixture(scope="module")
def driver(tmp_path_factory, solver_mod):
    exe = str(tmp_path_factory.mktemp("refsurf") / "reference_surface_driver")
    cpp = os.path.join(ROOT, "tests", "cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(cpp, "refstub"), "-I", cpp, DRV,
           "-o", exe, "-L", PKG_DIR, "-lvsmpc", f"-Wl,-rpath,{PKG_DIR}", "-Wl,-rpath,/opt/rocm/lib"]
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
        assert mpc.solveMPC()
        rec = np.array(get_record(mpc), dtype=float)          # the record that was solved (fused tick: complete after solveMPC)
        assert rec.shape == (n_in,)
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


def _kin_record(robot, layout):
    k = np.zeros(layout.KIN_SIZE)
    k[layout.KIN_WRB:layout.KIN_WRB + 9] = robot.getBaseRotation().reshape(-1)
    k[layout.KIN_THRUST:layout.KIN_THRUST + 4] = robot.T
    k[layout.KIN_AXES:layout.KIN_AXES + 12] = robot.axes.reshape(-1)
    k[layout.KIN_ARMS:layout.KIN_ARMS + 12] = robot.arms.reshape(-1)
    k[layout.KIN_JREL:layout.KIN_JREL + 276] = np.stack([j[3:6] for j in robot.jrel]).reshape(-1)
    k[layout.KIN_JFRAME:layout.KIN_JFRAME + 276] = np.stack([j[0:3, 6:29] for j in robot.jframe]).reshape(-1)
    k[layout.KIN_JCOM:layout.KIN_JCOM + 69] = robot.jcom[0:3, 6:29].reshape(-1)
    k[layout.KIN_MB:layout.KIN_MB + 36] = robot.M[0:6, 0:6].reshape(-1)
    k[layout.KIN_R:layout.KIN_R + 3] = robot.p - robot.base
    return k


@pytest.mark.parametrize("mode", ["fused", "two-call"])
def test_compiled_surface_matches_the_reference_tick_model(driver, pinned, mode, tmp_path, layout, solver_mod, ref):
    """The C++ instantiation over the reference-typed stand-ins against the model written from the reference's plugins
    (tests/tick_model.py; kinematics terms from the oracle) -- no product module in the checker.  Every record the binding
    builds, tick by tick, and its outputs = the C-ABI's outputs for that record.  Both tick forms: one vsmpc_tick submission
    per tick (default) and the update() / solveMPC() pair with a kinematics round trip in update()."""
    consts, traj = pinned
    sc = fp.Scenario(n_ticks=N_TICKS, seed=17)
    selector = list(range(3, 11))
    (tmp_path / "scenario.bin").write_bytes(sc.serialise(consts, traj, selector).tobytes())
    args = [driver, str(tmp_path / "scenario.bin"), str(tmp_path / "out.bin")] + (["two-call"] if mode == "two-call" else [])
    res = subprocess.run(args, capture_output=True, text=True)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    cfg = layout.paper_config()
    n_in = cfg.n_in
    cpp = np.frombuffer((tmp_path / "out.bin").read_bytes(), dtype=np.float64).reshape(N_TICKS, -1)

    robot = sc.robot
    q = sc.initial_qp
    fb = dict(throttle=q["throttle"].copy(), thrustDes=q["thrustDes"].copy(), thrustDotDes=q["thrustDotDes"].copy(),
              joints=q["joints"].copy())

    def plant_arrays(est_td, q_ref0):
        s = np.zeros(layout.PLANT_STATE); p = np.zeros(layout.PLANT_PARAMS)
        s[layout.PS_P:layout.PS_P + 3] = robot.p
        s[layout.PS_HLIN:layout.PS_HLIN + 3] = robot.h[0:3]
        s[layout.PS_RPY:layout.PS_RPY + 3] = robot.rpy
        s[layout.PS_HANG:layout.PS_HANG + 3] = robot.h[3:6]
        s[layout.PS_T:layout.PS_T + 4] = robot.T
        s[layout.PS_TD:layout.PS_TD + 4] = est_td
        s[layout.PS_Q:layout.PS_Q + 8] = fb["joints"][3:11]
        s[layout.PS_U:layout.PS_U + 4] = fb["throttle"]
        s[layout.PS_TDES:layout.PS_TDES + 4] = fb["thrustDes"]
        s[layout.PS_TDDES:layout.PS_TDDES + 4] = fb["thrustDotDes"]
        p[layout.PP_MASS] = robot.mass
        p[layout.PP_QREF0:layout.PP_QREF0 + 8] = q_ref0
        return s, p

    est0 = sc.load(0)
    q_ref0 = robot.q[3:11].copy()
    s0, p0 = plant_arrays(est0, q_ref0)
    model = tm.ReferenceTickModel(cfg, s0, p0, traj["positionCoM"], traj["velocityCoM"], traj["alphaGravity"])
    raw = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=1)
    try:
        for k in range(N_TICKS):
            est = sc.load(k + 1)
            rec = cpp[k, :n_in]
            s, p = plant_arrays(est, q_ref0)
            kin = np.zeros(n_in)
            R = robot.getBaseRotation()
            kin[layout.IN_MASS] = float(np.float32(robot.mass))
            kin[layout.IN_WRB:layout.IN_WRB + 9] = R.reshape(-1)
            kin[layout.IN_OMEGA:layout.IN_OMEGA + 3] = R.T @ robot.omega_world
            kin[layout.IN_GRAV:layout.IN_GRAV + 3] = [0.0, 0.0, -9.81]
            kin[layout.IN_AMOM:layout.IN_AMOM + 24] = robot.amom_body.reshape(-1)
            l1, l2, ig = ref.kinematics_terms(_kin_record(robot, layout))
            kin[layout.IN_LLIN:layout.IN_LLIN + 24] = l1.reshape(-1)
            kin[layout.IN_LANG:layout.IN_LANG + 24] = l2.reshape(-1)
            kin[layout.IN_INERTIA:layout.IN_INERTIA + 9] = ig.reshape(-1)
            kin[layout.IN_T0:layout.IN_T0 + 4] = robot.T
            kin[layout.IN_TD0:layout.IN_TD0 + 4] = est
            fields = model.update(s)
            exp = tm.record_from_tick(cfg, fields, kin)
            assert relerr(rec, exp) < 1e-12, (k, int(np.abs(rec - exp).argmax()))
            assert rec[layout.IN_HOLD] == (0.0 if k % 20 == 19 else 1.0)
            x, fm, st, it = raw.solve(rec[None, :])
            o = n_in
            assert cpp[k, o] == st[0] == layout.STATUS_SOLVED
            np.testing.assert_array_equal(cpp[k, o + 1:o + 5], fm[0, 16:20])       # thrust reference
            np.testing.assert_array_equal(cpp[k, o + 5:o + 9], fm[0, 20:24])       # thrust-rate reference
            np.testing.assert_array_equal(cpp[k, o + 9:o + 13], fm[0, 12:16])      # throttle reference
            model.consume(fm[0], st[0])
            np.testing.assert_allclose(cpp[k, o + 13 + 3:o + 13 + 11], model.m_jointsPositionReference, rtol=0, atol=1e-15)
            # what the binding wrote back into the QPInput (costsVSMPC.cpp:155-160, systemDynamicsVSMPC.cpp:310)
            assert cpp[k, o + 36 + 6] == fields["alpha"]
            fb = dict(throttle=cpp[k, o + 9:o + 13].copy(), thrustDes=cpp[k, o + 1:o + 5].copy(),
                      thrustDotDes=cpp[k, o + 5:o + 9].copy(), joints=cpp[k, o + 13:o + 36].copy())
    finally:
        raw.close()
        sc.load(0)



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
