"""Model of the REFERENCE's per-tick state machine, written from the reference's plugin code (not from the kernels):

    TrajectoryManager           utils/src/TrajectoryManager.cpp:23-39 (upsample), :142-153 (advance, clamped), :160-167
    ReferenceTrackingCost       momentum-based-linear-mpc-lib/src/variableSamplingMPC/costsVSMPC.cpp:68-69 (rates),
                                :103-118 (configureDynVectorsSize), :121-165 (window FIFO in computeHessianAndGradient)
    ThrottleConstraint          .../constraintsVSMPC.cpp:335 (counter start), :351-372 (hold, counter)
    LinearMomentumDynamicVS     .../systemDynamicsVSMPC.cpp:272 (des_fps = 1 / periodMPC -> int), :308-311 (use, then advance)
    ConstraintInitialState      .../constraintsVSMPC.cpp:184-200 (configure), :206-247 (X0, unwrapRPY)
    IMPCProblem::configure      .../IMPCProblem/IMPCProblem.cpp:80-132: ONE computeHessianAndGradient per cost and ONE
                                computeConstraintsMatrixAndBounds per constraint at configure time
    IMPCProblem::update         .../IMPCProblem.cpp:150-194: costs first (list order), then constraints
                                (dynamics -> initial state -> throttle, variableSamplingMPC.cpp:70-84)
    VariableSamplingMPC::solveMPC  variableSamplingMPC.cpp:88-112 (consume only if Solved, joint accumulator)
    harness loop                src/variable_sampling_mpc.py:106-135 (what is fed back into QPInput)

Each class keeps the reference's member names.  Test infrastructure: tests/test_gpu_rollout.py drives it with the plant
states of the device rollout and compares the records it builds with the device's, tick by tick.
"""
from __future__ import annotations

import importlib
import math

import numpy as np

PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
L = importlib.import_module(PKG + ".layout")


class Trajectory:
    def __init__(self, values, fps):
        self.values = [np.array(v, dtype=float) for v in values]
        self.fps = fps

    def upsample(self, des_fps):                               # TrajectoryManager.cpp:23-39
        resampled = []
        ratio = float(des_fps) / self.fps
        for i in range(len(self.values) - 1):
            k = 0
            while k < ratio:
                resampled.append(self.values[i] + (self.values[i + 1] - self.values[i]) * (k / ratio))
                k += 1
        self.values = resampled
        self.fps = des_fps


class TrajectoryManager:
    def __init__(self, tracks: dict, fps: int, des_fps: int):  # loadTrajectoryFromFile, TrajectoryManager.cpp:67-140
        self.trajectories_map = {}
        self.trajectorySize = 0
        self.trajectoryIndex = 0
        for name, arr in tracks.items():
            arr = np.atleast_2d(np.asarray(arr, dtype=float))
            if arr.shape[0] == 1 and arr.shape[1] > 1 and name == "alphaGravity":
                arr = arr.T                                    # 1 x n in the MAT file: one scalar per sample
            tr = Trajectory(list(arr), fps)
            if fps != des_fps and len(tr.values) > 1:
                tr.upsample(des_fps)
            self.trajectories_map[name] = tr
            self.trajectorySize = max(self.trajectorySize, len(tr.values))

    def advanceTrajectory(self):                               # :142-153
        if self.trajectoryIndex < self.trajectorySize - 1:
            self.trajectoryIndex += 1

    def getCurrentValue(self, key):                            # :160-167
        return self.trajectories_map[key].values[self.trajectoryIndex]


def rot(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def as_rpy(angle_continuous):
    """Rotation::asRPY() of a rotation built from continuously integrated angles: each angle wrapped to (-pi, pi]
    (small roll / pitch, so no gimbal reshuffle)."""
    a = np.asarray(angle_continuous, dtype=float)
    return a - 2.0 * math.pi * np.rint(a / (2.0 * math.pi))


class Robot:
    """The Robot getters the path consumes, backed by the synthetic plant's state / parameters."""

    def __init__(self, s, p):
        self.s, self.p = np.asarray(s, float), np.asarray(p, float)

    def getPositionCoM(self):
        return self.s[L.PS_P:L.PS_P + 3]

    def getRotation(self):
        return rot(self.s[L.PS_RPY:L.PS_RPY + 3])

    def asRPY(self):
        return as_rpy(self.s[L.PS_RPY:L.PS_RPY + 3])

    def getTotalMass(self):
        return self.p[L.PP_MASS]

    def getMomentumBody(self):
        return np.concatenate([self.s[L.PS_HLIN:L.PS_HLIN + 3], self.s[L.PS_HANG:L.PS_HANG + 3]])

    def getJetThrusts(self):
        return self.s[L.PS_T:L.PS_T + 4]

    def getLockedInertia(self):
        """(X^T M_b X).block(3, 3, 3, 3) of costsVSMPC.cpp:266-286: the world-oriented locked inertia, R I_B R^T for the
        synthetic plant"""
        R = self.getRotation()
        return R @ self.p[L.PP_INERTIA_B:L.PP_INERTIA_B + 9].reshape(3, 3) @ R.T


class QPInput:
    """Fields of the QPInput bus the path reads / writes (utils/include/QPInput.h:12-124)."""

    def __init__(self):
        self.posCoMReference = np.zeros(3)
        self.RPYReference = np.zeros(3)
        self.momentumReference = np.zeros(6)
        self.alphaGravity = 0.0
        self.throttleMPC = np.zeros(4)
        self.thrustDesMPC = np.zeros(4)
        self.thrustDotDesMPC = np.zeros(4)
        self.estimatedThrustDot = np.zeros(4)
        self.outputQPJointsPositionSel = np.zeros(8)   # the 8 controlled entries of getOutputQPJointsPosition


class ReferenceTrackingCost:
    def __init__(self, cfg, tracks_pos, fps):
        self.nIter, self.nIterSmall = cfg.n_iter, cfg.n_iter_small
        self.m_trajManager = TrajectoryManager(tracks_pos, fps, int(1 / cfg.period_large))            # costsVSMPC.cpp:68
        self.m_ratioSmallLargeStepsPeriod = int(round(cfg.period_large / cfg.period_small))          # :69

    def _column(self, robot):                                                                        # :105-112, :133-146
        p = self.m_initialCoMPos + self.m_trajManager.getCurrentValue("positionCoM")
        hl = robot.getRotation().T @ (robot.getTotalMass() * self.m_trajManager.getCurrentValue("velocityCoM"))
        rpy = self.m_initialRPY + self.m_trajManager.getCurrentValue("RPY")
        r0, p0 = robot.asRPY()[0], robot.asRPY()[1]                                                  # updateInertiaMatrix, :266-286
        W = np.array([[1.0, 0.0, -math.sin(p0)], [0.0, math.cos(r0), math.cos(p0) * math.sin(r0)],
                      [0.0, -math.sin(r0), math.cos(r0) * math.cos(p0)]])
        ha = robot.getLockedInertia() @ W @ self.m_trajManager.getCurrentValue("RPYDot")              # m_inertia * m_W * RPYDot
        return p, hl, rpy, ha

    def configureDynVectorsSize(self, robot):                                                        # :74-119
        n = self.nIter - self.nIterSmall + 1
        self.m_initialCoMPos = robot.getPositionCoM().copy()
        self.m_initialRPY = robot.asRPY().copy()
        col = self._column(robot)
        self.m_positionCoMReference = np.tile(col[0][:, None], (1, n))
        self.m_linearMomentumReference = np.tile(col[1][:, None], (1, n))
        self.m_RPYReference = np.tile(col[2][:, None], (1, n))
        self.m_angularMomentumReference = np.tile(col[3][:, None], (1, n))
        self.m_counter = self.m_ratioSmallLargeStepsPeriod - 1

    def configureDynVectorsSize_refill(self, robot):
        """initial fill of the window with externally given m_initialCoMPos / m_initialRPY (:103-113)"""
        n = self.nIter - self.nIterSmall + 1
        col = self._column(robot)
        self.m_positionCoMReference = np.tile(col[0][:, None], (1, n))
        self.m_linearMomentumReference = np.tile(col[1][:, None], (1, n))
        self.m_RPYReference = np.tile(col[2][:, None], (1, n))
        self.m_angularMomentumReference = np.tile(col[3][:, None], (1, n))

    def computeHessianAndGradient(self, robot, qp: QPInput):                                         # :121-165
        if self.m_counter == self.m_ratioSmallLargeStepsPeriod - 1:
            self.m_trajManager.advanceTrajectory()
            p, hl, rpy, ha = self._column(robot)
            self.m_positionCoMReference = np.column_stack([self.m_positionCoMReference[:, 1:], p])
            self.m_linearMomentumReference = np.column_stack([self.m_linearMomentumReference[:, 1:], hl])
            self.m_RPYReference = np.column_stack([self.m_RPYReference[:, 1:], rpy])
            self.m_angularMomentumReference = np.column_stack([self.m_angularMomentumReference[:, 1:], ha])
            qp.posCoMReference = self.m_positionCoMReference[:, 0].copy()
            qp.RPYReference = self.m_RPYReference[:, 0].copy()
            qp.momentumReference = np.concatenate([self.m_linearMomentumReference[:, 0], self.m_angularMomentumReference[:, 0]])
            self.m_counter = 0
        else:
            self.m_counter += 1

    def window(self):
        """[n, 12] rows = columns of the four reference matrices (CoM, h_lin, RPY, h_ang)."""
        return np.vstack([self.m_positionCoMReference, self.m_linearMomentumReference, self.m_RPYReference,
                          self.m_angularMomentumReference]).T


class ThrottleConstraint:
    def __init__(self, cfg):
        self.m_ratio = int(round(cfg.period_large / cfg.period_small))                               # constraintsVSMPC.cpp:322
        self.m_counter = self.m_ratio - 1                                                            # :335

    def computeConstraintsMatrixAndBounds(self) -> bool:
        """Returns True when block 0 is pinned to the previous throttle this call (:351-364)."""
        pinned = self.m_counter != self.m_ratio - 1
        self.m_counter = 0 if self.m_counter == self.m_ratio - 1 else self.m_counter + 1             # :366-372
        return pinned


class LinearMomentumDynamicVS:
    def __init__(self, cfg, alpha_track, fps):
        self.m_trajectoryManager = TrajectoryManager({"alphaGravity": np.asarray(alpha_track, float)[None, :]}, fps,
                                                     int(1 / cfg.period_mpc))                        # systemDynamicsVSMPC.cpp:272

    def computeLinearMomentumMatrices(self, qp: QPInput) -> float:                                   # :308-311
        a = float(self.m_trajectoryManager.getCurrentValue("alphaGravity")[0])
        qp.alphaGravity = a
        self.m_trajectoryManager.advanceTrajectory()
        return a


class ConstraintInitialState:
    def configureDynVectorsSize(self, robot):                                                        # constraintsVSMPC.cpp:184-200
        self.m_initialRPY = robot.asRPY().copy()
        self.m_rpyOld = self.m_initialRPY.copy()
        self.m_nTurns = np.zeros(3)

    def unwrapRPY(self, robot):                                                                      # :232-247
        rpy = robot.asRPY()
        for i in range(3):
            if rpy[i] - self.m_rpyOld[i] > math.pi:
                self.m_nTurns[i] -= 1
            elif rpy[i] - self.m_rpyOld[i] < -math.pi:
                self.m_nTurns[i] += 1
        self.m_rpyUnwrapped = rpy + 2 * math.pi * self.m_nTurns
        self.m_rpyOld = rpy.copy()

    def updateInitialState(self, robot, qp: QPInput):                                                # :206-230, useEstimatedThrust
        self.unwrapRPY(robot)
        x0 = np.zeros(26)
        x0[0:3] = robot.getPositionCoM()
        x0[3:6] = robot.getMomentumBody()[0:3]
        x0[6:9] = self.m_rpyUnwrapped
        x0[9:12] = robot.getMomentumBody()[3:6]
        x0[12:16] = robot.getJetThrusts()
        x0[16:20] = qp.estimatedThrustDot
        x0[20:23] = robot.getPositionCoM() - qp.posCoMReference
        x0[23:26] = self.m_rpyUnwrapped - qp.RPYReference
        return x0


class ReferenceTickModel:
    """VariableSamplingMPC + the harness's feedback into QPInput, for one instance of the synthetic plant."""

    def __init__(self, cfg, s0, p, traj_pos, traj_vel, traj_alpha, fps_traj=10, fps_alpha=10, ticks_before=0,
                 configured_elsewhere=False, traj_rpy=None, traj_rpy_dot=None):
        """`ticks_before` > 0 with `configured_elsewhere`: the loop was configured earlier, at CoM PP_PINIT / attitude
        PP_RPYINIT, and has run `ticks_before` ticks with the CURRENT attitude and state (the definition the device uses
        for loops that start mid-trajectory); reproduced by running the state machine that many times on `s0`."""
        self.cfg, self.p = cfg, np.asarray(p, float)
        n = len(traj_pos)
        tracks = {"positionCoM": traj_pos, "velocityCoM": traj_vel,
                  "RPY": np.zeros((n, 3)) if traj_rpy is None else traj_rpy,
                  "RPYDot": np.zeros((n, 3)) if traj_rpy_dot is None else traj_rpy_dot}
        self.cost = ReferenceTrackingCost(cfg, tracks, fps_traj)
        self.throttle = ThrottleConstraint(cfg)
        self.linmom = LinearMomentumDynamicVS(cfg, traj_alpha, fps_alpha)
        self.init_state = ConstraintInitialState()
        self.qp = QPInput()
        s0 = np.asarray(s0, float)
        # harness state before configure (variable_sampling_mpc.py:49-71): previous commands = current plant values
        self.qp.throttleMPC = s0[L.PS_U:L.PS_U + 4].copy()
        self.qp.thrustDesMPC = s0[L.PS_TDES:L.PS_TDES + 4].copy()
        self.qp.thrustDotDesMPC = s0[L.PS_TDDES:L.PS_TDDES + 4].copy()
        self.m_jointsPositionReference = s0[L.PS_Q:L.PS_Q + 8].copy()                                # variableSamplingMPC.cpp:59-60
        self.qp.outputQPJointsPositionSel = self.m_jointsPositionReference.copy()
        self.m_jointPosReference = self.p[L.PP_QREF0:L.PP_QREF0 + 8].copy()                          # costsVSMPC.cpp:539-550
        robot = Robot(s0, self.p)
        self.m_rpyInit = robot.asRPY().copy()                                                        # systemDynamicsVSMPC.cpp:67
        # IMPCProblem::configure: plugins sized, then ONE evaluation of each (IMPCProblem.cpp:80-132)
        self.cost.configureDynVectorsSize(robot)
        self.init_state.configureDynVectorsSize(robot)
        if configured_elsewhere:
            self.m_rpyInit = self.p[L.PP_RPYINIT:L.PP_RPYINIT + 3].copy()
            self.cost.m_initialCoMPos = self.p[L.PP_PINIT:L.PP_PINIT + 3].copy()
            self.cost.m_initialRPY = self.p[L.PP_RPYINIT:L.PP_RPYINIT + 3].copy()
            self.cost.configureDynVectorsSize_refill(robot)
            cont = s0[L.PS_RPY:L.PS_RPY + 3]
            self.init_state.m_nTurns = np.rint((cont - robot.asRPY()) / (2 * math.pi))
        self.qp.estimatedThrustDot = s0[L.PS_TD:L.PS_TD + 4].copy()
        self.cost.computeHessianAndGradient(robot, self.qp)
        self.linmom.computeLinearMomentumMatrices(self.qp)
        self.init_state.updateInitialState(robot, self.qp)
        self.throttle.computeConstraintsMatrixAndBounds()
        self.status = 0
        for _ in range(int(ticks_before)):
            self.update(s0)

    def update(self, s) -> dict:
        """IMPCProblem::update on the plant state `s`: the tick-state-dependent record fields."""
        robot = Robot(s, self.p)
        self.qp.estimatedThrustDot = np.asarray(s, float)[L.PS_TD:L.PS_TD + 4].copy()               # harness :108-109
        self.cost.computeHessianAndGradient(robot, self.qp)                 # costs first
        alpha = self.linmom.computeLinearMomentumMatrices(self.qp)          # dynamics
        x0 = self.init_state.updateInitialState(robot, self.qp)             # initial state
        hold = self.throttle.computeConstraintsMatrixAndBounds()            # throttle box
        return {"xref": self.cost.window(), "pref": self.qp.posCoMReference.copy(), "x0": x0, "alpha": alpha,
                "hold": 1.0 if hold else 0.0, "rpy": robot.asRPY(), "rpy_init": self.m_rpyInit.copy(),
                "uprev": self.qp.throttleMPC.copy(), "tdes": self.qp.thrustDesMPC.copy(),
                "tddes": self.qp.thrustDotDesMPC.copy(),
                "qerr": self.qp.outputQPJointsPositionSel - self.m_jointPosReference}

    def consume(self, fm, status):
        """solveMPC + the harness's set* calls (variableSamplingMPC.cpp:91-108, variable_sampling_mpc.py:124-135)."""
        self.status = int(status)
        if self.status == L.STATUS_SOLVED:
            self.m_jointsPositionReference = self.m_jointsPositionReference + fm[L.FM_DQ:L.FM_DQ + 8]
            self.qp.throttleMPC = np.array(fm[L.FM_THROTTLE:L.FM_THROTTLE + 4])
            self.qp.thrustDesMPC = np.array(fm[L.FM_THRUST:L.FM_THRUST + 4])
            self.qp.thrustDotDesMPC = np.array(fm[L.FM_THRUSTDOT:L.FM_THRUSTDOT + 4])
        self.qp.outputQPJointsPositionSel = self.m_jointsPositionReference.copy()


def record_from_tick(cfg, fields: dict, kin_record: np.ndarray) -> np.ndarray:
    """Full vsmpc input record: the tick-state fields of ReferenceTickModel.update over the kinematics-derived fields
    of `kin_record` (A_mom, Lambda, I_G, R, omega, mass, gravity: functions of the current plant state only)."""
    rec = kin_record.copy()
    rec[L.IN_XREF:L.IN_XREF + 12 * cfg.n_ref_cols] = fields["xref"].reshape(-1)
    rec[L.IN_PREF:L.IN_PREF + 3] = fields["pref"]
    rec[L.IN_X0:L.IN_X0 + 26] = fields["x0"]
    rec[L.IN_ALPHA] = fields["alpha"]
    rec[L.IN_HOLD] = fields["hold"]
    rec[L.IN_RPY:L.IN_RPY + 3] = fields["rpy"]
    rec[L.IN_RPYINIT:L.IN_RPYINIT + 3] = fields["rpy_init"]
    rec[L.IN_UPREV:L.IN_UPREV + 4] = fields["uprev"]
    rec[L.IN_TDES:L.IN_TDES + 4] = fields["tdes"]
    rec[L.IN_TDDES:L.IN_TDDES + 4] = fields["tddes"]
    rec[L.IN_QERR:L.IN_QERR + 8] = fields["qerr"]
    return rec
