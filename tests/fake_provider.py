"""Fake Robot / scenario for the reference-surface tests (test helper only).

`FakeRobot` exposes the Robot getters the path reads (utils/include/Robot.h; what utils/src/Robot.cpp:198-335 caches),
backed by plain arrays.  `Scenario` moves one or two such robots through `n_ticks` ticks (CoM drift, yaw through +pi,
changing thrusts and Jacobians) and can serialise itself for tests/cpp/reference_surface_driver.cpp, so that the Python
twin (reference_api), the pybind module (bindingsMPC) and the C++ template (include/VariableSamplingMPC.hpp) are driven
through the SAME sequence of provider states."""
from __future__ import annotations

import math

import numpy as np

import tick_model as tm

NJ = 23
JETS = ["l_arm_jet_turbine", "r_arm_jet_turbine", "chest_l_jet_turbine", "chest_r_jet_turbine"]


class FakeRobot:
    def __init__(self, rng, mass=70.3):
        self.mass = mass
        self.p = np.array([0.1, -0.2, 1.0])
        self.base = self.p + np.array([0.02, 0.0, -0.15])
        self.rpy = np.array([0.02, -0.03, math.pi - 0.015])
        self.h = rng.normal(0, 0.5, 6)
        self.omega_world = np.array([0.1, -0.05, 0.7])
        self.T = np.array([160.0, 170.0, 175.0, 165.0])
        self.q = rng.normal(0, 0.1, NJ)
        self.axes = rng.normal(size=(4, 3)); self.axes /= np.linalg.norm(self.axes, axis=1)[:, None]
        self.arms = rng.normal(0, 0.2, size=(4, 3))
        self.jrel = [rng.normal(0, 0.3, size=(6, NJ)) for _ in range(4)]
        self.jframe = [rng.normal(0, 0.3, size=(6, 6 + NJ)) for _ in range(4)]
        self.jcom = rng.normal(0, 0.3, size=(3, 6 + NJ))
        a = rng.normal(size=(6 + NJ, 6 + NJ)); self.M = a @ a.T + (6 + NJ) * np.eye(6 + NJ)
        self.amom_body = rng.normal(0, 0.3, size=(6, 4)); self.amom_body[2] = 0.95

    # --- the provider protocol
    def getNJoints(self): return NJ
    def getNJets(self): return 4
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
    def getJetsList(self): return list(JETS)
    def getJacobian(self, frameName): return self.jframe[JETS.index(frameName)]
    def getJacobianCoM(self): return self.jcom
    def getJointPos(self): return self.q
    def getJointName(self, i): return f"joint_{i}"

    # --- scenario support
    def advance(self, rng, scale=1.0):
        self.p = self.p + scale * np.array([0.001, -0.0005, 0.002])
        self.base = self.p + np.array([0.02, 0.0, -0.15])
        self.rpy = self.rpy + scale * np.array([0.0004, -0.0002, 0.003])
        self.h = self.h + rng.normal(0, 0.01, 6)
        self.T = self.T + rng.normal(0, 0.3, 4)
        self.jrel = [j + rng.normal(0, 0.002, j.shape) for j in self.jrel]

    def block(self):
        """the robot block of the C++ driver's scenario file (tests/cpp/reference_surface_driver.cpp)"""
        return np.concatenate([self.p, self.base, tm.rot(self.rpy).reshape(-1), self.omega_world, self.h, self.T, self.q,
                               self.axes.reshape(-1), self.arms.reshape(-1)] + [j.reshape(-1) for j in self.jrel]
                              + [j.reshape(-1) for j in self.jframe]
                              + [self.jcom.reshape(-1), self.M[0:6, 0:6].reshape(-1), self.amom_body.reshape(-1), [self.mass],
                                 [0.0, 0.0, -9.81]])

    def snapshot(self):
        return {k: (v.copy() if isinstance(v, np.ndarray) else [a.copy() for a in v] if isinstance(v, list) else v)
                for k, v in self.__dict__.items()}

    def restore(self, snap):
        for k, v in snap.items():
            setattr(self, k, v.copy() if isinstance(v, np.ndarray) else [a.copy() for a in v] if isinstance(v, list) else v)


class Scenario:
    """Pre-generated provider states: states[k] = (robot snapshot, reference-robot snapshot or None, estimatedThrustDot);
    index 0 is the configure-time state."""

    def __init__(self, n_ticks=45, seed=17, distinct=False):
        rng = np.random.default_rng(seed)
        self.robot = FakeRobot(rng)
        self.reference = FakeRobot(rng, mass=69.1) if distinct else self.robot
        self.distinct = distinct
        self.n_ticks = n_ticks
        self.initial_qp = dict(throttle=np.array([70.0, 72.0, 74.0, 71.0]), thrustDes=self.robot.T.copy(), thrustDotDes=np.zeros(4),
                               estTd=np.array([1.0, -2.0, 0.5, 0.0]), joints=self.robot.q.copy())
        self.states = [(self.robot.snapshot(), self.reference.snapshot() if distinct else None, self.initial_qp["estTd"].copy())]
        for _ in range(n_ticks):
            self.robot.advance(rng)
            if distinct:
                self.reference.advance(rng, scale=0.7)
            self.states.append((self.robot.snapshot(), self.reference.snapshot() if distinct else None, rng.normal(0, 3.0, 4)))
        self.load(0)

    def load(self, k):
        r, rr, est = self.states[k]
        self.robot.restore(r)
        if self.distinct:
            self.reference.restore(rr)
        return est

    def serialise(self, consts, traj, selector, use_estimated=True, constant_lambda=False, drop_key=False):
        c = consts["VS_MPC_CONFIG"]
        pos, vel, rpy, rpyd = (np.asarray(traj[k], float) for k in ("positionCoM", "velocityCoM", "RPY", "RPYDot"))
        alpha = np.asarray(traj["alphaGravity"], float).reshape(-1)
        head = np.zeros(16)
        head[0:9] = [self.n_ticks, len(pos), int(traj["trajectory_fps"][0]), len(alpha), int(traj["alphaGravity_fps"][0]),
                     float(use_estimated), float(constant_lambda), float(self.distinct), float(drop_key)]
        cfg = [c["nIter"], c["nIterSmall"], c["controlHorizon"], float(c["useJetDynamic"]), c["periodMPC"], c["periodMPCSmallSteps"],
               c["periodMPCLargeSteps"]]
        for k in ("weightCoMPos", "weightCoMPosError", "weightLinMom", "weightRPY", "weightRPYError", "weightAngMom"):
            cfg += list(c[k])
        cfg += list(c["weightDeltaJoint"])
        cfg += [c["weightThrottle"], c["weightInitialThrottle"], c["weightRegularizationJointPos"], c["throttleMin"], c["throttleMax"]]
        q = self.initial_qp
        parts = [head, pos.reshape(-1), vel.reshape(-1), rpy.reshape(-1), rpyd.reshape(-1), alpha, np.asarray(cfg, float),
                 np.asarray(selector, float), q["throttle"], q["thrustDes"], q["thrustDotDes"], q["estTd"], q["joints"]]
        for k in range(self.n_ticks + 1):
            est = self.load(k)
            parts.append(self.robot.block())
            if self.distinct:
                parts.append(self.reference.block())
            parts.append(est)
        self.load(0)
        return np.concatenate([np.asarray(p, float).reshape(-1) for p in parts])
