"""Record layouts of the C-ABI in include/vsmpc.h (offsets in doubles) and the MPC configuration.

Mirrors the *data content* of the reference's QPInput bus (utils/include/QPInput.h:12-124) and
config group VS_MPC_CONFIG (src/config/vs_mcp_config.xml:5-45); sizes follow
variableSamplingMPC.cpp:42-45 and VSconstant.h:6-16.
"""
from __future__ import annotations

import ctypes
import dataclasses

N_STATES = 26
N_JOINTS = 8
N_THRUSTS = 4

# vsmpc_input record (must match include/vsmpc.h)
IN_X0 = 0
IN_MASS = 26
IN_WRB = 27
IN_OMEGA = 36
IN_ALPHA = 39
IN_GRAV = 40
IN_AMOM = 43
IN_LLIN = 67
IN_LANG = 91
IN_INERTIA = 115
IN_RPY = 124
IN_PREF = 127
IN_RPYINIT = 130
IN_T0 = 133
IN_TD0 = 137
IN_UPREV = 141
IN_TDES = 145
IN_TDDES = 149
IN_QERR = 153
IN_HOLD = 161
IN_XREF = 162

# kinematics record of vsmpc_kinematics_batch (VSMPC_KIN_* in include/vsmpc.h)
KIN_NJ = 23
KIN_WRB = 0
KIN_THRUST = 9
KIN_AXES = 13
KIN_ARMS = 25
KIN_JREL = 37
KIN_JFRAME = 313
KIN_JCOM = 589
KIN_MB = 658
KIN_R = 694
KIN_SIZE = 697
KIN_OUT = 57

# first-move block (24 doubles): dq(8) v0(4) throttle%(4) T1(4) Tdot1(4)
FM_DQ = 0
FM_V0 = 8
FM_THROTTLE = 12
FM_THRUST = 16
FM_THRUSTDOT = 20
FM_SIZE = 24

# closed-loop rollout: plant state / parameter layouts (VSMPC_PS_* / VSMPC_PP_* in include/vsmpc.h)
PS_P, PS_HLIN, PS_RPY, PS_HANG, PS_T, PS_TD, PS_Q, PS_U, PS_TDES, PS_TDDES = 0, 3, 6, 9, 12, 16, 20, 28, 32, 36
PS_TNN, PS_EST, PS_EKFP = 40, 44, 52     # jet plant option (LSTM thrust, EKF estimates (T, Tdot) x 4, covariances 2x2 x 4)
PLANT_STATE = 68
PP_MASS, PP_INERTIA_B, PP_AMOM0, PP_DJ, PP_QREF0, PP_PINIT, PP_RPYINIT = 0, 1, 10, 34, 226, 234, 237
PP_DIST_F, PP_DIST_TAU, PP_DIST_T0, PP_DIST_T1, PP_TICK0 = 240, 243, 246, 247, 248
PLANT_PARAMS = 249
ROLLOUT_LOG = 16

# per-instance status (mirrors the OsqpEigen::Status values the reference distinguishes,
# IMPCProblem.cpp:285-294, variableSamplingMPC.cpp:91)
STATUS_SOLVED = 1
STATUS_MAX_ITER = 2
STATUS_NUMERICAL = 3


class CConfig(ctypes.Structure):
    """ctypes image of `vsmpc_config` in include/vsmpc.h."""
    _fields_ = [
        ("n_iter", ctypes.c_int),
        ("n_iter_small", ctypes.c_int),
        ("control_horizon", ctypes.c_int),
        ("use_jet_dynamic", ctypes.c_int),
        ("period_mpc", ctypes.c_double),
        ("period_small", ctypes.c_double),
        ("period_large", ctypes.c_double),
        ("w_com_pos", ctypes.c_double * 3),
        ("w_com_pos_err", ctypes.c_double * 3),
        ("w_lin_mom", ctypes.c_double * 3),
        ("w_rpy", ctypes.c_double * 3),
        ("w_rpy_err", ctypes.c_double * 3),
        ("w_ang_mom", ctypes.c_double * 3),
        ("w_delta_joint", ctypes.c_double * 8),
        ("w_throttle", ctypes.c_double),
        ("w_initial_throttle", ctypes.c_double),
        ("w_reg_joint_pos", ctypes.c_double),
        ("throttle_min", ctypes.c_double),
        ("throttle_max", ctypes.c_double),
    ]


@dataclasses.dataclass
class MPCConfig:
    """Keys of VS_MPC_CONFIG (src/config/vs_mcp_config.xml:7-43); defaults are the paper values."""
    n_iter: int = 17
    n_iter_small: int = 7
    control_horizon: int = 12
    use_jet_dynamic: bool = True
    period_mpc: float = 0.005
    period_small: float = 0.005
    period_large: float = 0.1
    w_com_pos: tuple = (500.0, 500.0, 5000.0)
    w_com_pos_err: tuple = (25000.0, 25000.0, 50000.0)
    w_lin_mom: tuple = (1.0, 1.0, 1.5)
    w_rpy: tuple = (1000.0, 1000.0, 1000.0)
    w_rpy_err: tuple = (10000.0, 10000.0, 10000.0)
    w_ang_mom: tuple = (80.0, 80.0, 80.0)
    w_delta_joint: tuple = (65000.0,) * 8
    w_throttle: float = 80000.0
    w_initial_throttle: float = 80000.0
    w_reg_joint_pos: float = 20.0
    throttle_min: float = 0.0
    throttle_max: float = 100.0

    @property
    def n_vblocks(self) -> int:
        return self.control_horizon - self.n_iter_small + 1

    @property
    def n_var(self) -> int:  # variableSamplingMPC.cpp:44-45
        return (N_STATES * (self.n_iter + 1) + N_JOINTS * self.control_horizon
                + N_THRUSTS * self.n_vblocks)

    @property
    def n_con(self) -> int:  # constraintsVSMPC.cpp:7,283 ; IQPUtilsMPC.cpp:60-63
        return N_STATES * (self.n_iter + 1) + N_THRUSTS * (self.n_iter - self.n_iter_small + 1)

    @property
    def n_ref_cols(self) -> int:  # costsVSMPC.cpp:96-99
        return self.n_iter - self.n_iter_small + 1

    @property
    def n_in(self) -> int:
        return IN_XREF + 12 * self.n_ref_cols

    @property
    def n_inputs(self) -> int:
        return N_JOINTS * self.control_horizon + N_THRUSTS * self.n_vblocks

    @property
    def off_joints(self) -> int:
        return N_STATES * (self.n_iter + 1)

    @property
    def off_throttle(self) -> int:
        return self.off_joints + N_JOINTS * self.control_horizon

    @property
    def ratio(self) -> int:  # constraintsVSMPC.cpp:322
        return int(round(self.period_large / self.period_small))

    def to_c(self) -> CConfig:
        c = CConfig()
        c.n_iter, c.n_iter_small, c.control_horizon = self.n_iter, self.n_iter_small, self.control_horizon
        c.use_jet_dynamic = 1 if self.use_jet_dynamic else 0
        c.period_mpc, c.period_small, c.period_large = self.period_mpc, self.period_small, self.period_large
        for name in ("w_com_pos", "w_com_pos_err", "w_lin_mom", "w_rpy", "w_rpy_err", "w_ang_mom"):
            setattr(c, name, (ctypes.c_double * 3)(*getattr(self, name)))
        c.w_delta_joint = (ctypes.c_double * 8)(*self.w_delta_joint)
        c.w_throttle, c.w_initial_throttle = self.w_throttle, self.w_initial_throttle
        c.w_reg_joint_pos = self.w_reg_joint_pos
        c.throttle_min, c.throttle_max = self.throttle_min, self.throttle_max
        return c


def paper_config() -> MPCConfig:
    return MPCConfig()


def horizon2x_config() -> MPCConfig:
    """BASELINE.json configs[4]: 2x horizon at halved fast-rate dt."""
    return MPCConfig(n_iter=34, n_iter_small=14, control_horizon=24, period_small=0.0025)
