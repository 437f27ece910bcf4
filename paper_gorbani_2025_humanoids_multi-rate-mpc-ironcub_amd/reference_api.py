"""The reference's own call signatures on top of the MI355X path:

    mpc = VariableSamplingMPC()
    mpc.configure(paramHandler, qpInput)      # MPCPyBindings.cpp:24-32  -> IMPCProblem::configure
    mpc.update(qpInput)                       # MPCPyBindings.cpp:33-37  -> IMPCProblem::update
    mpc.solveMPC(); mpc.getThrustReference() ...                          # variableSamplingMPC.cpp:88-151

so that a harness written like src/variable_sampling_mpc.py:68-135 runs against it.  `qpInput` is any object with the
QPInput getters / setters the path uses (utils/include/QPInput.h:12-124) whose getRobot() / getRobotReference() return
objects with the Robot getters the path reads (utils/include/Robot.h; the list is the one include/vsmpc.h cites per
field).  Provider protocol (names of the reference's own Python binding, flightCtrlPyBindings.cpp:66-92, where it has
one): getPositionCoM, getMomentum(inBodyCoord), getBasePosition, getBaseOrientation (RPY) or getBaseRotation (3x3),
getBaseAngVel, getJetThrusts, getTotalMass, getGravity, getMassMatrix, getMatrixAmomJets(inBodyCoord),
getMatrixOfJetAxes, getMatrixOfJetArms, getRelativeJacobianJetsBodyFrame, getJetsList / getJacobian(frameName) (= getJacobian of jet
frame i), getJacobianCoM, getJointPos.  `QPInput` below is such an object for callers that do not have the reference's bindings; `paramHandler` is a
mapping with the VS_MPC_CONFIG keys of src/config/vs_mcp_config.xml:7-43 (or any object with getParameter(name)), with
the trajectories passed as arrays under "TRAJECTORY_MANAGER" / "POSITION_TRAJECTORY" (MAT-file reading stays outside).

What runs where: the PACKER (QPInput/Robot -> vsmpc_input record) and the per-instance tick state machine (reference
window FIFO costsVSMPC.cpp:103-165, throttle-hold counter constraintsVSMPC.cpp:335-372, alpha-gravity cursor
systemDynamicsVSMPC.cpp:308-311 + TrajectoryManager.cpp:23-39,142-153, RPY unwrap constraintsVSMPC.cpp:232-247, joint
accumulator variableSamplingMPC.cpp:104-108) are host bookkeeping here, exactly the state the reference keeps inside
its plugins; everything numeric -- Lambda_lin / Lambda_ang / I_G from the Jacobians (vsmpc_kinematics_batch) and the
whole update()+solveMPC() arithmetic (vsmpc_solve_batch) -- runs in libvsmpc.so on the GPU.  No CPU solve path.
"""
from __future__ import annotations

import math

import numpy as np

from . import layout as L
from .jet_model import JetModel
from .solver import BatchedVSMPC

N_ROBOT_JOINTS = 23      # MPCPyBindings.cpp:43
JOINT_OFFSET = 3         # controlled joints = robot joints 3..10 (systemDynamicsVSMPC.cpp:348)


class QPInput:
    """Data bus between harness and MPC with the reference's accessor names (utils/include/QPInput.h:12-124); only the
    members the path touches."""

    def __init__(self, robot=None, robot_reference=None):
        self._robot, self._robotReference = robot, robot_reference if robot_reference is not None else robot
        self._throttleMPC = np.zeros(4)
        self._thrustDesMPC = np.zeros(4)
        self._thrustDotDesMPC = np.zeros(4)
        self._estimatedThrustDot = np.zeros(4)
        self._outputQPJointsPosition = np.zeros(N_ROBOT_JOINTS)
        self._posCoMReference = np.zeros(3)
        self._RPYReference = np.zeros(3)
        self._momentumReference = np.zeros(6)
        self._alphaGravity = 0.0
        self._jetModel = JetModel()

    def getRobot(self): return self._robot
    def getRobotReference(self): return self._robotReference
    def getJetModel(self): return self._jetModel
    def setThrottleMPC(self, v): self._throttleMPC = np.array(v, dtype=float)
    def getThrottleMPC(self): return self._throttleMPC
    def setThrustDesMPC(self, v): self._thrustDesMPC = np.array(v, dtype=float)
    def getThrustDesMPC(self): return self._thrustDesMPC
    def setThrustDotDesMPC(self, v): self._thrustDotDesMPC = np.array(v, dtype=float)
    def getThrustDotDesMPC(self): return self._thrustDotDesMPC
    def setEstimatedThrustDot(self, v): self._estimatedThrustDot = np.array(v, dtype=float)
    def getEstimatedThrustDot(self): return self._estimatedThrustDot
    def setOutputQPJointsPosition(self, v): self._outputQPJointsPosition = np.array(v, dtype=float)
    def getOutputQPJointsPosition(self): return self._outputQPJointsPosition
    def setPosCoMReference(self, v): self._posCoMReference = np.array(v, dtype=float)
    def getPosCoMReference(self): return self._posCoMReference
    def setRPYReference(self, v): self._RPYReference = np.array(v, dtype=float)
    def getRPYReference(self): return self._RPYReference
    def setMomentumReference(self, v): self._momentumReference = np.array(v, dtype=float)
    def getMomentumReference(self): return self._momentumReference
    def setAlphaGravity(self, v): self._alphaGravity = float(v)
    def getAlphaGravity(self): return self._alphaGravity


_BLF_GETTERS = {"int": "get_parameter_int", "float": "get_parameter_float", "bool": "get_parameter_bool",
                "str": "get_parameter_string", "vec": "get_parameter_vector_float", "strvec": "get_parameter_vector_string"}


def _param(handler, name, default=None, kind=None):
    """One key of a parameters handler: BLF's Python handler (get_parameter_int / _float / _bool / _string /
    _vector_float / _vector_string, as src/variable_sampling_mpc.py:38-40 uses it), the C++ spelling getParameter(name),
    or a mapping."""
    v = None
    blf = _BLF_GETTERS.get(kind or "", None)
    if blf is not None and hasattr(handler, blf):
        try:
            v = getattr(handler, blf)(name)
        except (ValueError, KeyError, RuntimeError):     # BLF raises when the key is missing
            v = None
    elif hasattr(handler, "getParameter"):
        v = handler.getParameter(name)
        if isinstance(v, tuple):                         # (ok, value)
            ok, v = v
            if not ok:
                v = None
    elif hasattr(handler, "get"):
        v = handler.get(name)
    if v is None:
        if default is None:
            raise KeyError(f"Parameter '{name}' not found in the config file.")
        return default
    return v


def _group(handler, name):
    for getter in ("get_group", "getGroup"):
        if hasattr(handler, getter):
            return getattr(handler, getter)(name)
    return handler[name]


class _Track:
    """TrajectoryManager for one file (TrajectoryManager.cpp): linear up-sampling to `des_fps` (the last original sample
    is dropped, :23-39), index cursor clamped at the last sample (:142-153)."""

    def __init__(self, tracks: dict, fps: int, des_fps: int):
        self.values = {}
        self.size = 0
        self.index = 0
        for name, arr in tracks.items():
            a = np.asarray(arr, dtype=float)
            a = a.reshape(-1, 1) if a.ndim == 1 else a
            if fps != des_fps and len(a) > 1:
                ratio = float(des_fps) / fps
                steps = int(math.ceil(ratio))
                k = np.arange(steps) / ratio
                a = (a[:-1, None, :] + (a[1:, None, :] - a[:-1, None, :]) * k[None, :, None]).reshape(-1, a.shape[1])
            self.values[name] = a
            self.size = max(self.size, len(a))

    def advance(self):
        if self.index < self.size - 1:
            self.index += 1

    def current(self, name):
        return self.values[name][self.index]


def _base_rotation(robot):
    """wR_b as a 3x3 array.  The reference's Python Robot returns RPY from getBaseOrientation()
    (flightCtrlPyBindings.cpp:75-78); a provider may instead offer getBaseRotation() (the matrix itself)."""
    if hasattr(robot, "getBaseRotation"):
        return np.asarray(robot.getBaseRotation(), dtype=float).reshape(3, 3)
    r, p, y = (float(v) for v in robot.getBaseOrientation())
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _rpy_of(R):
    """iDynTree Rotation::asRPY (R = Rz(y) Ry(p) Rx(r))."""
    R = np.asarray(R, dtype=float).reshape(3, 3)
    return np.array([math.atan2(R[2, 1], R[2, 2]), math.atan2(-R[2, 0], math.hypot(R[2, 1], R[2, 2])),
                     math.atan2(R[1, 0], R[0, 0])])


class VariableSamplingMPC:
    """momentum_based_mpc.bindingsMPC.VariableSamplingMPC with the reference's signatures."""

    def __init__(self, device: int = 0):
        self._device = device
        self._solver = None
        self._status = 0

    trajectory_loader = None   # loader(trajectoryFile) -> {variable: array, "fps": int}; trajectory_io.load_mat_trajectory

    def setTrajectoryLoader(self, loader):
        self.trajectory_loader = loader

    def _resolve_group(self, group):
        """a trajectory group as a mapping of arrays: the group itself, or -- when it holds a `trajectoryFile` name like the
        handler the harness reads from XML (src/config/vs_mcp_config.xml:34-40) -- what the loader returns for it"""
        try:
            name = _param(group, "trajectoryFile", kind="str")
        except (KeyError, ValueError, TypeError):
            return group
        if self.trajectory_loader is None:
            raise ValueError(f"group holds trajectoryFile '{name}' but no trajectory loader is set (setTrajectoryLoader)")
        return self.trajectory_loader(name)

    # ------------------------------------------------------------------ configure (IMPCProblem.cpp:3-148)
    def configure(self, parametersHandler, qpInput) -> bool:
        try:
            h = _group(parametersHandler, "VS_MPC_CONFIG") if _has_group(parametersHandler, "VS_MPC_CONFIG") else parametersHandler
            cfg = L.MPCConfig(
                n_iter=int(_param(h, "nIter", kind="int")), n_iter_small=int(_param(h, "nIterSmall", kind="int")),
                control_horizon=int(_param(h, "controlHorizon", kind="int")),
                use_jet_dynamic=bool(_param(h, "useJetDynamic", kind="bool")),
                period_mpc=float(_param(h, "periodMPC", kind="float")),
                period_small=float(_param(h, "periodMPCSmallSteps", kind="float")),
                period_large=float(_param(h, "periodMPCLargeSteps", kind="float")),
                w_com_pos=tuple(_param(h, "weightCoMPos", kind="vec")), w_com_pos_err=tuple(_param(h, "weightCoMPosError", kind="vec")),
                w_lin_mom=tuple(_param(h, "weightLinMom", kind="vec")), w_rpy=tuple(_param(h, "weightRPY", kind="vec")),
                w_rpy_err=tuple(_param(h, "weightRPYError", kind="vec")), w_ang_mom=tuple(_param(h, "weightAngMom", kind="vec")),
                w_delta_joint=tuple(_param(h, "weightDeltaJoint", kind="vec")), w_throttle=float(_param(h, "weightThrottle", kind="float")),
                w_initial_throttle=float(_param(h, "weightInitialThrottle", kind="float")),
                w_reg_joint_pos=float(_param(h, "weightRegularizationJointPos", kind="float")),
                throttle_min=float(_param(h, "throttleMin", kind="float")), throttle_max=float(_param(h, "throttleMax", kind="float")))
            self._useEstimatedThrust = bool(_param(h, "useEstimatedThrust", True, kind="bool"))
            option = str(_param(h, "jointsLambdaOption", "unfiltered", kind="str"))
            if option not in ("unfiltered", "constant"):             # systemDynamicsVSMPC.cpp:33-46
                raise ValueError("Parameter 'jointsLambdaOption' should be 'unfiltered' or 'constant'.")
            self._constantLambda = option == "constant"
            self._controlledJoints = list(_param(h, "controlledJoints", [], kind="strvec"))
            tm, pt = self._resolve_group(_group(h, "TRAJECTORY_MANAGER")), self._resolve_group(_group(h, "POSITION_TRAJECTORY"))
            self._alpha = _Track({"alphaGravity": np.asarray(tm["alphaGravity"], float).reshape(-1)}, int(tm.get("fps", 10)),
                                 int(1 / cfg.period_mpc))                                    # systemDynamicsVSMPC.cpp:272
            n = len(pt["positionCoM"])
            self._traj = _Track({"positionCoM": pt["positionCoM"], "velocityCoM": pt["velocityCoM"],
                                 "RPY": pt.get("RPY", np.zeros((n, 3))), "RPYDot": pt.get("RPYDot", np.zeros((n, 3)))},
                                int(pt.get("fps", 10)), int(1 / cfg.period_large))           # costsVSMPC.cpp:68
        except (KeyError, ValueError, TypeError) as exc:
            print(f"[VariableSamplingMPC::configure] {exc}")
            return False
        self.cfg = cfg
        self._solver = BatchedVSMPC(cfg, device=self._device, max_batch=1)
        self._ratio = cfg.ratio
        robot = qpInput.getRobot()
        # controlled joints by NAME where the robot can name its joints (variableSamplingMPC.cpp:47-58,
        # costsVSMPC.cpp:539-550, systemDynamicsVSMPC.cpp:57-66); the shipped robot has them at 3..10
        self._sel = list(range(JOINT_OFFSET, JOINT_OFFSET + 8))
        if self._controlledJoints and hasattr(robot, "getJointName"):
            names = [robot.getJointName(i) for i in range(N_ROBOT_JOINTS)]
            self._sel = [names.index(n) for n in self._controlledJoints]
            if len(self._sel) != 8:
                print("[VariableSamplingMPC::configure] 'controlledJoints' must name the 8 controlled joints")
                return False
        self._solver.set_kinematics_options(self._sel, self._constantLambda)
        if self._constantLambda:                                     # systemDynamicsVSMPC.cpp:53-55,276-277
            self._relJacInit = [np.array(j, dtype=float) for j in robot.getRelativeJacobianJetsBodyFrame()]
            self._axesInit = np.array(robot.getMatrixOfJetAxes(), dtype=float)
            self._armsInit = np.array(robot.getMatrixOfJetArms(), dtype=float)
        # plugin members set at configure time
        self._m_initialCoMPos = np.array(robot.getPositionCoM(), dtype=float)                # costsVSMPC.cpp:101
        self._m_initialRPY = _rpy_of(_base_rotation(robot))                                # costsVSMPC.cpp:102
        self._m_rpyInit = _rpy_of(_base_rotation(qpInput.getRobotReference()))             # systemDynamicsVSMPC.cpp:67
        self._m_rpyOld = self._m_initialRPY.copy()                                           # constraintsVSMPC.cpp:198
        self._m_nTurns = np.zeros(3)
        q = np.array(robot.getJointPos(), dtype=float)
        self._m_jointsPositionReference = q.copy()                                           # variableSamplingMPC.cpp:59-60
        self._m_jointPosReference = q[self._sel].copy()                                      # costsVSMPC.cpp:539-550
        ncol = cfg.n_ref_cols
        col = self._reference_column(robot)
        self._window = np.tile(col[None, :], (ncol, 1))                                      # costsVSMPC.cpp:103-113
        self._refCounter = self._ratio - 1                                                   # costsVSMPC.cpp:118
        self._throttleCounter = self._ratio - 1                                              # constraintsVSMPC.cpp:335
        self._thrustReference = np.zeros(4)
        self._thrustDotReference = np.zeros(4)
        self._throttleReference = np.zeros(4)                                                # warped v (variableSamplingMPC.cpp:100)
        self._deltaJoints = np.zeros(8)
        self._QPSolution = np.zeros(cfg.n_var)
        self._finalState = np.zeros(26)
        self._record = None
        # IMPCProblem::configure evaluates every cost and constraint ONCE (IMPCProblem.cpp:80-132)
        self._assemble(qpInput)
        return True

    # ------------------------------------------------------------------ per tick
    def update(self, qpInput) -> bool:
        if self._solver is None:
            return False
        self._record = self._assemble(qpInput)
        return True

    def solveMPC(self) -> bool:
        x, fm, status, _ = self._solver.solve(self._record[None, :])
        self._status = int(status[0])
        if self._status == L.STATUS_SOLVED:                                                  # variableSamplingMPC.cpp:91
            self._QPSolution = x[0]
            self._deltaJoints = fm[0, L.FM_DQ:L.FM_DQ + 8].copy()
            self._throttleReference = fm[0, L.FM_V0:L.FM_V0 + 4].copy()
            self._thrustReference = fm[0, L.FM_THRUST:L.FM_THRUST + 4].copy()
            self._thrustDotReference = fm[0, L.FM_THRUSTDOT:L.FM_THRUSTDOT + 4].copy()
            self._finalState = x[0, 26 * self.cfg.n_iter:26 * (self.cfg.n_iter + 1)].copy()
            self._m_jointsPositionReference[self._sel] += self._deltaJoints                  # :104-108
        return True                                                                           # :111

    # ------------------------------------------------------------------ getters (variableSamplingMPC.cpp:114-227)
    def getQPProblemStatus(self): return self._status
    def getJointsReferencePosition(self): return self._m_jointsPositionReference.copy()
    def getThrustReference(self): return self._thrustReference.copy()
    def getThrustDotReference(self): return self._thrustDotReference.copy()
    def getMPCSolution(self): return self._QPSolution[self.cfg.off_joints:].copy()
    def getNStatesMPC(self): return 26.0
    def getNInputMPC(self): return 12.0
    def getFinalCoMPosition(self): return self._finalState[0:3].copy()
    def getFinalLinMom(self): return self._finalState[3:6].copy()
    def getFinalRPY(self): return self._finalState[6:9].copy()
    def getFinalAngMom(self): return self._finalState[9:12].copy()

    def getThrottleReference(self):
        """destandardizeThrottle_u2T of the stored warped throttle, clamped to [0, 100] (variableSamplingMPC.cpp:138-151)."""
        return np.array([JetModel().destandardizeThrottle_u2T(v) for v in self._throttleReference])

    # ------------------------------------------------------------------ packer + tick state machine
    def _reference_column(self, robot):
        """one window column from the CURRENT trajectory sample, attitude and mass (costsVSMPC.cpp:105-112,127-146)."""
        R = np.asarray(_base_rotation(robot), dtype=float).reshape(3, 3)
        col = np.zeros(12)
        col[0:3] = self._m_initialCoMPos + self._traj.current("positionCoM")
        col[3:6] = R.T @ (float(robot.getTotalMass()) * self._traj.current("velocityCoM"))
        col[6:9] = self._m_initialRPY + self._traj.current("RPY")
        rpy_dot = self._traj.current("RPYDot")
        if np.any(rpy_dot):                                  # m_inertia * m_W * RPYDot (costsVSMPC.cpp:111-112,266-286)
            rpy = _rpy_of(R)
            W = np.array([[1.0, 0.0, -math.sin(rpy[1])], [0.0, math.cos(rpy[0]), math.cos(rpy[1]) * math.sin(rpy[0])],
                          [0.0, -math.sin(rpy[0]), math.cos(rpy[0]) * math.cos(rpy[1])]])
            col[9:12] = self._locked_inertia(robot) @ W @ rpy_dot
        return col

    def _locked_inertia(self, robot):
        """I_G of getRobot() (costsVSMPC.cpp:266-286) on the device: a kinematics record that carries only what I_G needs."""
        k = np.zeros(L.KIN_SIZE)
        k[L.KIN_WRB:L.KIN_WRB + 9] = np.asarray(_base_rotation(robot), float).reshape(-1)
        k[L.KIN_MB:L.KIN_MB + 36] = np.asarray(robot.getMassMatrix(), float)[0:6, 0:6].reshape(-1)
        k[L.KIN_R:L.KIN_R + 3] = np.asarray(robot.getPositionCoM(), float) - np.asarray(robot.getBasePosition(), float)
        return self._solver.kinematics(k[None, :])[2][0]

    def _kin_record(self, robot, plant=None):
        """raw Robot quantities in the VSMPC_KIN_* layout (include/vsmpc.h) for vsmpc_kinematics_batch; `robot` is
        getRobotReference(), `plant` getRobot() (its thrusts scale the angular term of the 'constant' option)."""
        k = np.zeros(L.KIN_SIZE)
        k[L.KIN_WRB:L.KIN_WRB + 9] = np.asarray(_base_rotation(robot), float).reshape(-1)
        k[L.KIN_THRUST:L.KIN_THRUST + 4] = robot.getJetThrusts()
        if self._constantLambda:                            # configure-time geometry (systemDynamicsVSMPC.cpp:186-200,329-337)
            k[L.KIN_AXES:L.KIN_AXES + 12] = self._axesInit.reshape(-1)
            k[L.KIN_ARMS:L.KIN_ARMS + 12] = self._armsInit.reshape(-1)
            k[L.KIN_JREL:L.KIN_JREL + 276] = np.stack([j[3:6, :] for j in self._relJacInit]).reshape(-1)
            k[L.KIN_JFRAME:L.KIN_JFRAME + 276] = np.stack([j[0:3, :] for j in self._relJacInit]).reshape(-1)
            k[L.KIN_JCOM:L.KIN_JCOM + 4] = (plant if plant is not None else robot).getJetThrusts()
        else:
            k[L.KIN_AXES:L.KIN_AXES + 12] = np.asarray(robot.getMatrixOfJetAxes(), float).reshape(-1)   # 4 x 3
            k[L.KIN_ARMS:L.KIN_ARMS + 12] = np.asarray(robot.getMatrixOfJetArms(), float).reshape(-1)   # 4 x 3
            jrel = robot.getRelativeJacobianJetsBodyFrame()                                              # 4 x (6 x 23)
            k[L.KIN_JREL:L.KIN_JREL + 276] = np.stack([np.asarray(j, float)[3:6, :] for j in jrel]).reshape(-1)
            jac = [robot.getJacobian(name) for name in robot.getJetsList()]         # systemDynamicsVSMPC.cpp:167,208-212
            k[L.KIN_JFRAME:L.KIN_JFRAME + 276] = np.stack([np.asarray(j, float)[0:3, 6:29] for j in jac]).reshape(-1)
            k[L.KIN_JCOM:L.KIN_JCOM + 69] = np.asarray(robot.getJacobianCoM(), float)[0:3, 6:29].reshape(-1)
        k[L.KIN_MB:L.KIN_MB + 36] = np.asarray(robot.getMassMatrix(), float)[0:6, 0:6].reshape(-1)
        k[L.KIN_R:L.KIN_R + 3] = np.asarray(robot.getPositionCoM(), float) - np.asarray(robot.getBasePosition(), float)
        return k

    def _assemble(self, qp) -> np.ndarray:
        """One IMPCProblem::update worth of plugin evaluations (costs first, then dynamics, initial state, throttle box:
        IMPCProblem.cpp:150-194 with the plugin order of variableSamplingMPC.cpp:70-84) -> the vsmpc input record."""
        cfg, robot, ref = self.cfg, qp.getRobot(), qp.getRobotReference()
        rec = np.zeros(cfg.n_in)
        # --- ReferenceTrackingCost::computeHessianAndGradient (costsVSMPC.cpp:121-165)
        if self._refCounter == self._ratio - 1:
            self._traj.advance()
            self._window = np.vstack([self._window[1:], self._reference_column(robot)[None, :]])
            qp.setPosCoMReference(self._window[0, 0:3])
            qp.setRPYReference(self._window[0, 6:9])
            qp.setMomentumReference(np.concatenate([self._window[0, 3:6], self._window[0, 9:12]]))
            self._refCounter = 0
        else:
            self._refCounter += 1
        rec[L.IN_XREF:L.IN_XREF + 12 * cfg.n_ref_cols] = self._window.reshape(-1)
        # --- JointPositionRegularizationCost (costsVSMPC.cpp:574-589)
        qcmd = np.asarray(qp.getOutputQPJointsPosition(), float)
        rec[L.IN_QERR:L.IN_QERR + 8] = qcmd[self._sel] - self._m_jointPosReference
        # --- dynamics (systemDynamicsVSMPC.cpp:79-103,288-319,384-429); the kinematics terms on the device
        R = np.asarray(_base_rotation(ref), float).reshape(3, 3)
        rec[L.IN_MASS] = float(np.float32(ref.getTotalMass()))                               # Robot.h:338 keeps a float
        rec[L.IN_WRB:L.IN_WRB + 9] = R.reshape(-1)
        rec[L.IN_OMEGA:L.IN_OMEGA + 3] = R.T @ np.asarray(robot.getBaseAngVel(), float)  # m_robot's: systemDynamicsVSMPC.cpp:108,325
        alpha = float(self._alpha.current("alphaGravity")[0])                                # :308-311: use, then advance
        qp.setAlphaGravity(alpha)
        self._alpha.advance()
        rec[L.IN_ALPHA] = alpha
        rec[L.IN_GRAV:L.IN_GRAV + 3] = ref.getGravity()
        rec[L.IN_AMOM:L.IN_AMOM + 24] = np.asarray(ref.getMatrixAmomJets(True), float).reshape(-1)
        Llin, Lang, IG = self._solver.kinematics(self._kin_record(ref, robot)[None, :])
        rec[L.IN_LLIN:L.IN_LLIN + 24] = Llin[0].reshape(-1)
        rec[L.IN_LANG:L.IN_LANG + 24] = Lang[0].reshape(-1)
        rec[L.IN_INERTIA:L.IN_INERTIA + 9] = IG[0].reshape(-1)
        rec[L.IN_RPY:L.IN_RPY + 3] = _rpy_of(R)
        rec[L.IN_PREF:L.IN_PREF + 3] = qp.getPosCoMReference()
        rec[L.IN_RPYINIT:L.IN_RPYINIT + 3] = self._m_rpyInit
        if self._useEstimatedThrust:                                                         # :401-409
            T0, Td0 = np.asarray(robot.getJetThrusts(), float), np.asarray(qp.getEstimatedThrustDot(), float)   # m_robot's (:403)
        else:
            T0, Td0 = np.asarray(qp.getThrustDesMPC(), float), np.asarray(qp.getThrustDotDesMPC(), float)
        rec[L.IN_T0:L.IN_T0 + 4] = T0
        rec[L.IN_TD0:L.IN_TD0 + 4] = Td0
        rec[L.IN_UPREV:L.IN_UPREV + 4] = qp.getThrottleMPC()
        rec[L.IN_TDES:L.IN_TDES + 4] = qp.getThrustDesMPC()
        rec[L.IN_TDDES:L.IN_TDDES + 4] = qp.getThrustDotDesMPC()
        # --- ConstraintInitialState (constraintsVSMPC.cpp:206-247)
        rpy = _rpy_of(_base_rotation(robot))
        for i in range(3):
            if rpy[i] - self._m_rpyOld[i] > math.pi:
                self._m_nTurns[i] -= 1
            elif rpy[i] - self._m_rpyOld[i] < -math.pi:
                self._m_nTurns[i] += 1
        unwrapped = rpy + 2 * math.pi * self._m_nTurns
        self._m_rpyOld = rpy.copy()
        mom = np.asarray(robot.getMomentum(True), float)
        x0 = rec[L.IN_X0:L.IN_X0 + 26]
        x0[0:3] = robot.getPositionCoM()
        x0[3:6] = mom[0:3]
        x0[6:9] = unwrapped
        x0[9:12] = mom[3:6]
        x0[12:16] = robot.getJetThrusts() if self._useEstimatedThrust else qp.getThrustDesMPC()
        x0[16:20] = qp.getEstimatedThrustDot() if self._useEstimatedThrust else qp.getThrustDotDesMPC()
        x0[20:23] = np.asarray(robot.getPositionCoM(), float) - np.asarray(qp.getPosCoMReference(), float)
        x0[23:26] = unwrapped - np.asarray(qp.getRPYReference(), float)
        # --- ThrottleConstraint (constraintsVSMPC.cpp:351-372)
        rec[L.IN_HOLD] = 1.0 if self._throttleCounter != self._ratio - 1 else 0.0
        self._throttleCounter = 0 if self._throttleCounter == self._ratio - 1 else self._throttleCounter + 1
        return rec


def _has_group(handler, name):
    try:
        return _group(handler, name) is not None
    except (KeyError, TypeError, AttributeError, ValueError, RuntimeError):   # (BLF raises when the group is missing)
        return False
