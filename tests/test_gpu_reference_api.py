"""The reference-signature adapter (reference_api.VariableSamplingMPC: configure(paramHandler, qpInput), update(qpInput),
solveMPC(), get*Reference) driven like src/variable_sampling_mpc.py:68-135 with a fake Robot / QPInput: the record its
packer + tick state machine builds every tick equals the record of the model written from the reference's plugins
(tests/tick_model.py) with the oracle's kinematics terms, and its outputs are the C-ABI's outputs for that record."""
import importlib
import json
import math
import os

import numpy as np
import pytest

import rollout_model as rm
import tick_model as tm
from conftest import PKG, ROOT, relerr

pytestmark = pytest.mark.gpu


class FakeRobot:
    """Robot getters of the provider protocol, backed by plain arrays (what utils/src/Robot.cpp:198-335 caches)."""

    def __init__(self, rng):
        self.rng = rng
        self.mass = 70.3
        self.p = np.array([0.1, -0.2, 1.0])
        self.base = self.p + np.array([0.02, 0.0, -0.15])
        self.rpy = np.array([0.02, -0.03, math.pi - 0.015])
        self.h = rng.normal(0, 0.5, 6)
        self.omega_world = np.array([0.1, -0.05, 0.7])
        self.T = np.array([160.0, 170.0, 175.0, 165.0])
        self.q = rng.normal(0, 0.1, 23)
        self.axes = rng.normal(size=(4, 3)); self.axes /= np.linalg.norm(self.axes, axis=1)[:, None]
        self.arms = rng.normal(0, 0.2, size=(4, 3))
        self.jrel = [rng.normal(0, 0.3, size=(6, 23)) for _ in range(4)]
        self.jframe = [rng.normal(0, 0.3, size=(6, 29)) for _ in range(4)]
        self.jcom = rng.normal(0, 0.3, size=(3, 29))
        a = rng.normal(size=(29, 29)); self.M = a @ a.T + 29 * np.eye(29)
        self.amom_body = rng.normal(0, 0.3, size=(6, 4)); self.amom_body[2] = 0.95

    def getPositionCoM(self): return self.p
    def getBasePosition(self): return self.base
    def getBaseRotation(self): return tm.rot(self.rpy)
    def getBaseAngVel(self): return self.omega_world
    def getMomentum(self, inBodyCoord=False): assert inBodyCoord; return self.h
    def getJetThrusts(self): return self.T
    def getTotalMass(self): return self.mass
    def getGravity(self): return np.array([0.0, 0.0, -9.81])
    def getMassMatrix(self): return self.M
    def getMatrixAmomJets(self, inBodyCoord=False): assert inBodyCoord; return self.amom_body
    def getMatrixOfJetAxes(self): return self.axes
    def getMatrixOfJetArms(self): return self.arms
    def getRelativeJacobianJetsBodyFrame(self): return self.jrel
    def getJetsList(self): return ["l_arm_jet_turbine", "r_arm_jet_turbine", "chest_l_jet_turbine", "chest_r_jet_turbine"]
    def getJacobian(self, frameName): return self.jframe[self.getJetsList().index(frameName)]
    def getJacobianCoM(self): return self.jcom
    def getJointPos(self): return self.q


def _kin(robot, layout):
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


def test_reference_signatures_build_the_reference_records(solver_mod, layout, ref):
    api = importlib.import_module(PKG + ".reference_api")
    consts = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))
    traj = np.load(os.path.join(ROOT, "tests", "golden", "reference_trajectories.npz"))
    params = dict(consts["VS_MPC_CONFIG"])                     # the keys of src/config/vs_mcp_config.xml, as pinned
    params["TRAJECTORY_MANAGER"] = {"alphaGravity": traj["alphaGravity"], "fps": int(traj["alphaGravity_fps"][0])}
    params["POSITION_TRAJECTORY"] = {"positionCoM": traj["positionCoM"], "velocityCoM": traj["velocityCoM"],
                                     "RPY": traj["RPY"], "RPYDot": traj["RPYDot"], "fps": int(traj["trajectory_fps"][0])}
    rng = np.random.default_rng(17)
    robot = FakeRobot(rng)
    qp = api.QPInput(robot)
    qp.setThrottleMPC([70.0, 72.0, 74.0, 71.0])
    qp.setThrustDesMPC(robot.T)
    qp.setOutputQPJointsPosition(robot.q)
    qp.setEstimatedThrustDot([1.0, -2.0, 0.5, 0.0])
    mpc = api.VariableSamplingMPC()
    assert mpc.configure(params, qp)
    assert not api.VariableSamplingMPC().configure({"nIter": 17}, qp)          # missing keys -> false, like the reference
    cfg = mpc.cfg
    assert cfg.n_var == 588
    np.testing.assert_array_equal(qp.getPosCoMReference(), robot.p + traj["positionCoM"][0])   # set by configure's evaluation

    # the reference-derived model, fed the same robot as a "plant state"
    def plant_arrays():
        s = np.zeros(layout.PLANT_STATE); p = np.zeros(layout.PLANT_PARAMS)
        s[layout.PS_P:layout.PS_P + 3] = robot.p
        s[layout.PS_HLIN:layout.PS_HLIN + 3] = robot.h[0:3]
        s[layout.PS_RPY:layout.PS_RPY + 3] = robot.rpy
        s[layout.PS_HANG:layout.PS_HANG + 3] = robot.h[3:6]
        s[layout.PS_T:layout.PS_T + 4] = robot.T
        s[layout.PS_TD:layout.PS_TD + 4] = qp.getEstimatedThrustDot()
        s[layout.PS_Q:layout.PS_Q + 8] = qp.getOutputQPJointsPosition()[3:11]
        s[layout.PS_U:layout.PS_U + 4] = qp.getThrottleMPC()
        s[layout.PS_TDES:layout.PS_TDES + 4] = qp.getThrustDesMPC()
        s[layout.PS_TDDES:layout.PS_TDDES + 4] = qp.getThrustDotDesMPC()
        p[layout.PP_MASS] = robot.mass
        p[layout.PP_QREF0:layout.PP_QREF0 + 8] = q_ref0
        return s, p
    q_ref0 = robot.q[3:11].copy()
    # (the model is configured on the state the adapter was configured on)
    robot_cfg_state = plant_arrays()
    model = tm.ReferenceTickModel(cfg, robot_cfg_state[0], robot_cfg_state[1], traj["positionCoM"], traj["velocityCoM"],
                                  traj["alphaGravity"])
    raw = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=1)
    try:
        for k in range(45):
            # the world moves: CoM drifts, yaw runs through +pi, thrusts and Jacobians change
            robot.p = robot.p + np.array([0.001, -0.0005, 0.002])
            robot.base = robot.p + np.array([0.02, 0.0, -0.15])
            robot.rpy = robot.rpy + np.array([0.0004, -0.0002, 0.003])
            robot.h = robot.h + rng.normal(0, 0.01, 6)
            robot.T = robot.T + rng.normal(0, 0.3, 4)
            robot.jrel = [j + rng.normal(0, 0.002, j.shape) for j in robot.jrel]
            qp.setEstimatedThrustDot(rng.normal(0, 3.0, 4))
            assert mpc.update(qp)
            rec = mpc._record.copy()
            # expected record: reference-derived tick fields over oracle kinematics fields
            s, p = plant_arrays()
            kin = np.zeros(cfg.n_in)
            R = robot.getBaseRotation()
            kin[layout.IN_MASS] = float(np.float32(robot.mass))
            kin[layout.IN_WRB:layout.IN_WRB + 9] = R.reshape(-1)
            kin[layout.IN_OMEGA:layout.IN_OMEGA + 3] = R.T @ robot.omega_world
            kin[layout.IN_GRAV:layout.IN_GRAV + 3] = [0.0, 0.0, -9.81]
            kin[layout.IN_AMOM:layout.IN_AMOM + 24] = robot.amom_body.reshape(-1)
            l1, l2, ig = ref.kinematics_terms(_kin(robot, layout))
            kin[layout.IN_LLIN:layout.IN_LLIN + 24] = l1.reshape(-1)
            kin[layout.IN_LANG:layout.IN_LANG + 24] = l2.reshape(-1)
            kin[layout.IN_INERTIA:layout.IN_INERTIA + 9] = ig.reshape(-1)
            kin[layout.IN_T0:layout.IN_T0 + 4] = robot.T
            kin[layout.IN_TD0:layout.IN_TD0 + 4] = qp.getEstimatedThrustDot()
            fields = model.update(s)
            # the tick model's column uses the plant mass un-rounded, like costsVSMPC.cpp:107-109 (getTotalMass)
            exp = tm.record_from_tick(cfg, fields, kin)
            assert relerr(rec, exp) < 1e-12, (k, int(np.abs(rec - exp).argmax()))
            assert rec[layout.IN_HOLD] == (0.0 if k % 20 == 19 else 1.0)
            assert mpc.solveMPC()
            x, fm, st, it = raw.solve(rec[None, :])
            assert mpc.getQPProblemStatus() == st[0] == layout.STATUS_SOLVED
            np.testing.assert_array_equal(mpc.getThrustReference(), fm[0, 16:20])
            np.testing.assert_array_equal(mpc.getThrustDotReference(), fm[0, 20:24])
            np.testing.assert_allclose(mpc.getThrottleReference(), fm[0, 12:16], rtol=0, atol=1e-12)
            model.consume(fm[0], st[0])
            # harness feedback (variable_sampling_mpc.py:124-135)
            qp.setThrottleMPC(mpc.getThrottleReference())
            qp.setThrustDesMPC(mpc.getThrustReference())
            qp.setThrustDotDesMPC(mpc.getThrustDotReference())
            qp.setOutputQPJointsPosition(mpc.getJointsReferencePosition())
            np.testing.assert_allclose(mpc.getJointsReferencePosition()[3:11], model.m_jointsPositionReference, rtol=0, atol=1e-15)
        assert abs(mpc._m_nTurns[2]) == 1 and qp.getAlphaGravity() == fields["alpha"]
    finally:
        raw.close()
