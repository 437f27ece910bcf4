"""
ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).

numpy restatement of the reference's multi-rate ("variable sampling") MPC hot path:
per-tick linearisation -> variable-sampling QP assembly in the reference's dense
plugin order -> exact QP optimum with a KKT optimality certificate.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and there only as the checker.  The product path (the HIP library behind
include/vsmpc.h) never imports, links or executes anything under oracle/.

Parity status: **parity unpinned**.  The reference has no tests, golden vectors or
fixtures for this path and can be neither compiled (needs Eigen, OsqpEigen/OSQP,
iDynTree, YARP, BLF, matio, boost) nor imported (needs mujoco, casadi, idyntree, ...)
in this image.  What pins this restatement instead:
  * closed-form known-answer values derived from the cited formulas (tests/test_oracle_kat.py),
  * two independent exact solution methods that must agree (null-space active set here,
    scipy BVLS in the tests) plus the KKT certificate of the *reference-ordered* dense QP,
  * the independent C restatement oracle/vsmpc_oracle.c (assembly must agree bit-for-bit in
    structure and to rounding in value; its OSQP-algorithm solve must land on the same optimum).

The QP arithmetic of the reference lives in third-party code that is not vendored:
osqp-eigen 0.11.0 -> libosqp 1.0.0 -> libqdldl 0.1.8 (pixi.lock:317,232,237), call sites
IMPCProblem.cpp:140-145,221-279,296.  OSQP stops at eps 1e-3 and then polishes; the only
reproducible target is the exact optimum of the QP, which is what solve_exact() returns.

All `file:line` citations are relative to
/root/reference/src/flight-controller/ (momentum-based-linear-mpc-lib/... and utils/...).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# Index map — variableSamplingMPC/VSconstant.h:6-40
# --------------------------------------------------------------------------------------
N_JOINTS = 8
N_THRUSTS = 4
N_STATES = 26  # variableSamplingMPC.cpp:42 (rpyErrorIdx[2] + 1)
IDX_COM = slice(0, 3)
IDX_LINMOM = slice(3, 6)
IDX_RPY = slice(6, 9)
IDX_ANGMOM = slice(9, 12)
IDX_T = slice(12, 16)
IDX_TDOT = slice(16, 20)
IDX_EPOS = slice(20, 23)
IDX_ERPY = slice(23, 26)

# --------------------------------------------------------------------------------------
# Jet model — utils/src/JetModel.cpp:10-114
# --------------------------------------------------------------------------------------
JET_COEFF = (
    -4.64730485e-01, -8.13171858e+00, -6.19539230e+00, 6.61113140e-01, 1.67673231e+00,
    -4.83287064e-01, 8.77996617e+00, -1.01096376e+00, -5.86442286e-01, 5.19093322e-01,
    -4.23782666e-01, -1.45705257e+00, -7.83052261e-03,
)  # JetModel.cpp:13-25
JET_NORM = (108.309, 65.793, 47.333, 31.483)  # JetModel.cpp:26 (mu_T, sigma_T, mu_u, sigma_u)


def jet_f(T, Td):  # JetModel.cpp:29-33
    c = JET_COEFF
    return c[0] + c[1] * T + c[2] * Td + c[3] * T * Td + c[4] * T ** 2 + c[5] * Td ** 2


def jet_g(T, Td):  # JetModel.cpp:55-59
    c = JET_COEFF
    return c[6] + c[7] * T + c[8] * Td + c[9] * T * Td + c[10] * T ** 2 + c[11] * Td ** 2


def jet_df_dT(T, Td):  # JetModel.cpp:35-38
    c = JET_COEFF
    return c[1] + c[3] * Td + 2 * c[4] * T


def jet_df_dTd(T, Td):  # JetModel.cpp:40-43
    c = JET_COEFF
    return c[2] + c[3] * T + 2 * c[5] * Td


def jet_dg_dT(T, Td):  # JetModel.cpp:45-48
    c = JET_COEFF
    return c[7] + c[9] * Td + 2 * c[10] * T


def jet_dg_dTd(T, Td):  # JetModel.cpp:50-53
    c = JET_COEFF
    return c[8] + c[9] * T + 2 * c[11] * Td


def jet_v(u_bar):  # JetModel.cpp:61-64
    return u_bar + JET_COEFF[12] * u_bar ** 2


def std_thrust(T):  # JetModel.cpp:66-69
    return (T - JET_NORM[0]) / JET_NORM[1]


def std_thrust_dot(Td):  # JetModel.cpp:71-74
    return Td / JET_NORM[1]


def std_throttle(u):  # JetModel.cpp:76-79
    return (u - JET_NORM[2]) / JET_NORM[3]


def destd_throttle(v):  # JetModel.cpp:93-109  (quadratic inverse + clamp to [0, 100])
    c12 = JET_COEFF[12]
    u = (-1.0 + np.sqrt(1.0 + 4.0 * c12 * v)) / (2.0 * c12)
    u = u * JET_NORM[3] + JET_NORM[2]
    return np.clip(u, 0.0, 100.0)


def v_of_throttle(u_percent):  # compute_v(standardizeThrottle_u2T(u)) — constraintsVSMPC.cpp:355-358
    return jet_v(std_throttle(u_percent))


# systemDynamicsVSMPC.cpp:431-461
def jetdyn_F(T, Td):
    return jet_f(std_thrust(T), std_thrust_dot(Td)) * JET_NORM[1]


def jetdyn_G(T, Td):
    return jet_g(std_thrust(T), std_thrust_dot(Td)) * JET_NORM[1]


def jetdyn_dh_dT(T, Td, throttle):
    Tb, Tdb, ub = std_thrust(T), std_thrust_dot(Td), std_throttle(throttle)
    return jet_df_dT(Tb, Tdb) + jet_dg_dT(Tb, Tdb) * jet_v(ub)


def jetdyn_dh_dTd(T, Td, throttle):
    Tb, Tdb, ub = std_thrust(T), std_thrust_dot(Td), std_throttle(throttle)
    return jet_df_dTd(Tb, Tdb) + jet_dg_dTd(Tb, Tdb) * jet_v(ub)


# --------------------------------------------------------------------------------------
# Configuration — src/config/vs_mcp_config.xml:7-43
# --------------------------------------------------------------------------------------
@dataclasses.dataclass
class Config:
    n_iter: int = 17                 # nIter
    n_iter_small: int = 7            # nIterSmall
    control_horizon: int = 12        # controlHorizon
    period_mpc: float = 0.005
    period_small: float = 0.005      # periodMPCSmallSteps
    period_large: float = 0.1        # periodMPCLargeSteps
    w_com_pos: Tuple[float, float, float] = (500.0, 500.0, 5000.0)
    w_com_pos_err: Tuple[float, float, float] = (25000.0, 25000.0, 50000.0)
    w_lin_mom: Tuple[float, float, float] = (1.0, 1.0, 1.5)
    w_rpy: Tuple[float, float, float] = (1000.0, 1000.0, 1000.0)
    w_rpy_err: Tuple[float, float, float] = (10000.0, 10000.0, 10000.0)
    w_ang_mom: Tuple[float, float, float] = (80.0, 80.0, 80.0)
    w_delta_joint: Tuple[float, ...] = (65000.0,) * 8
    w_throttle: float = 80000.0
    w_initial_throttle: float = 80000.0
    w_reg_joint_pos: float = 20.0
    throttle_min: float = 0.0
    throttle_max: float = 100.0
    use_jet_dynamic: bool = True

    # derived sizes — variableSamplingMPC.cpp:42-45, constraintsVSMPC.cpp:7,283
    @property
    def n_vblocks(self) -> int:
        return self.control_horizon - self.n_iter_small + 1

    @property
    def n_var(self) -> int:
        return N_STATES * (self.n_iter + 1) + N_JOINTS * self.control_horizon + N_THRUSTS * self.n_vblocks

    @property
    def n_con(self) -> int:
        return N_STATES * self.n_iter + N_STATES + N_THRUSTS * (self.n_iter - self.n_iter_small + 1)

    @property
    def n_ref_cols(self) -> int:  # costsVSMPC.cpp:96-99
        return self.n_iter - self.n_iter_small + 1

    @property
    def off_joints(self) -> int:
        return N_STATES * (self.n_iter + 1)

    @property
    def off_throttle(self) -> int:
        return self.off_joints + N_JOINTS * self.control_horizon

    @property
    def ratio(self) -> int:  # constraintsVSMPC.cpp:322
        return int(round(self.period_large / self.period_small))

    @property
    def n_in(self) -> int:
        return IN_XREF + 12 * self.n_ref_cols


def paper_config() -> Config:
    return Config()


def horizon2x_config() -> Config:
    """BASELINE.json config 5: 2x horizon at halved fast-rate dt."""
    return Config(n_iter=34, n_iter_small=14, control_horizon=24, period_small=0.0025)


# --------------------------------------------------------------------------------------
# Per-instance input record (the boundary of include/vsmpc.h).
# Offsets in doubles; the record is IN_XREF + 12 * n_ref_cols doubles long.
# --------------------------------------------------------------------------------------
IN_X0 = 0          # 26  measured state X0 (constraintsVSMPC.cpp:206-230)
IN_MASS = 26       # 1   Robot::getTotalMass (float-rounded by the caller, Robot.h:338)
IN_WRB = 27        # 9   wR_b row-major
IN_OMEGA = 36      # 3   omega_B = wR_b^T * omega_world (systemDynamicsVSMPC.cpp:108,325)
IN_ALPHA = 39      # 1   alpha_gravity (systemDynamicsVSMPC.cpp:308)
IN_GRAV = 40       # 3   gravity vector (world)
IN_AMOM = 43       # 24  getMatrixAmomJets(true) 6x4 row-major
IN_LLIN = 67       # 24  Lambda_lin,B 3x8 row-major (systemDynamicsVSMPC.cpp:348)
IN_LANG = 91       # 24  Lambda_ang,B 3x8 row-major (systemDynamicsVSMPC.cpp:202-205)
IN_INERTIA = 115   # 9   I_G 3x3 row-major (systemDynamicsVSMPC.cpp:128-130)
IN_RPY = 124       # 3   base RPY (for W^-1, systemDynamicsVSMPC.cpp:132-147)
IN_PREF = 127      # 3   QPInput::getPosCoMReference (systemDynamicsVSMPC.cpp:316)
IN_RPYINIT = 130   # 3   configure-time RPY (systemDynamicsVSMPC.cpp:67,100)
IN_T0 = 133        # 4   linearisation thrust
IN_TD0 = 137       # 4   linearisation thrust rate
IN_UPREV = 141     # 4   QPInput::getThrottleMPC (percent)
IN_TDES = 145      # 4   QPInput::getThrustDesMPC
IN_TDDES = 149     # 4   QPInput::getThrustDotDesMPC
IN_QERR = 153      # 8   q_cmd,sel - q_ref0 (costsVSMPC.cpp:574-589)
IN_HOLD = 161      # 1   1.0 if the throttle-hold counter pins v0 this tick (constraintsVSMPC.cpp:351)
IN_XREF = 162      # 12 * n_ref_cols, column-major: xref[col*12 + row], rows = (p, h_lin, rpy, h_ang)


def from_vec_to_skew(v):  # utils/src/FlightControlUtils.cpp:77-85
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def w_inverse(rpy):  # systemDynamicsVSMPC.cpp:140-147
    r, p = rpy[0], rpy[1]
    Wi = np.zeros((3, 3))
    Wi[0, 0] = 1.0
    Wi[0, 1] = math.sin(r) * math.tan(p)
    Wi[1, 1] = math.cos(r)
    Wi[2, 1] = math.sin(r) / math.cos(p)
    Wi[0, 2] = math.cos(r) * math.tan(p)
    Wi[1, 2] = -math.sin(r)
    Wi[2, 2] = math.cos(r) / math.cos(p)
    return Wi


def w_matrix(rpy):  # systemDynamicsVSMPC.cpp:133-139 (used by the harness side for h_ang references)
    r, p = rpy[0], rpy[1]
    W = np.zeros((3, 3))
    W[0, 0] = 1.0
    W[1, 1] = math.cos(r)
    W[2, 1] = -math.sin(r)
    W[0, 2] = -math.sin(p)
    W[1, 2] = math.cos(p) * math.sin(r)
    W[2, 2] = math.cos(r) * math.cos(p)
    return W


# --------------------------------------------------------------------------------------
# Linearisation: x_dot = A x + Bj dq + Bt v + c   
# --------------------------------------------------------------------------------------
def linearize(cfg: Config, inp: np.ndarray):
    """Sum of the three DynamicTemplateVariableSampling contributions
    (systemDynamicsVSMPC.cpp:509-585): angular :79-103, linear :288-319, jets :384-429."""
    A = np.zeros((N_STATES, N_STATES))
    Bj = np.zeros((N_STATES, N_JOINTS))
    Bt = np.zeros((N_STATES, N_THRUSTS))
    c = np.zeros(N_STATES)

    m = inp[IN_MASS]
    wRb = inp[IN_WRB:IN_WRB + 9].reshape(3, 3)
    om = inp[IN_OMEGA:IN_OMEGA + 3]
    Amom = inp[IN_AMOM:IN_AMOM + 24].reshape(6, 4)
    Llin = inp[IN_LLIN:IN_LLIN + 24].reshape(3, 8)
    Lang = inp[IN_LANG:IN_LANG + 24].reshape(3, 8)
    inertia = inp[IN_INERTIA:IN_INERTIA + 9].reshape(3, 3)
    rpy = inp[IN_RPY:IN_RPY + 3]

    # --- angular momentum (systemDynamicsVSMPC.cpp:79-103)
    A[IDX_RPY, IDX_ANGMOM] = w_inverse(rpy) @ np.linalg.inv(inertia)          # :86-87
    A[IDX_ANGMOM, IDX_ANGMOM] -= from_vec_to_skew(om)                          # :90-91
    A[IDX_ANGMOM, IDX_T] = Amom[3:6, :]                                        # :92-93
    Bj[IDX_ANGMOM, :] = Lang                                                   # :94-95
    A[IDX_ERPY, IDX_RPY] = np.eye(3)                                           # :98-99
    c[IDX_ERPY] = -inp[IN_RPYINIT:IN_RPYINIT + 3]                              # :100

    # --- linear momentum (systemDynamicsVSMPC.cpp:288-319)
    A[IDX_COM, IDX_LINMOM] = (1.0 / m) * wRb                                   # :296-297
    A[IDX_LINMOM, IDX_LINMOM] -= from_vec_to_skew(om)                          # :301-302
    A[IDX_LINMOM, IDX_T] = Amom[0:3, :]                                        # :303-304
    Bj[IDX_LINMOM, :] = Llin                                                   # :305-306
    c[IDX_LINMOM] = inp[IN_ALPHA] * m * (wRb.T @ inp[IN_GRAV:IN_GRAV + 3])     # :307-309
    A[IDX_EPOS, IDX_COM] = np.eye(3)                                           # :314-315
    c[IDX_EPOS] = -inp[IN_PREF:IN_PREF + 3]                                    # :316

    # --- jets (systemDynamicsVSMPC.cpp:384-429)
    if cfg.use_jet_dynamic:
        A[IDX_T, IDX_TDOT] = np.eye(4)                                         # :393-394
        for i in range(N_THRUSTS):
            T0, Td0, up = inp[IN_T0 + i], inp[IN_TD0 + i], inp[IN_UPREV + i]
            dhT = jetdyn_dh_dT(T0, Td0, up)
            dhTd = jetdyn_dh_dTd(T0, Td0, up)
            A[16 + i, 12 + i] = dhT                                            # :410-411
            A[16 + i, 16 + i] += dhTd                                          # :412-413
            Bt[16 + i, i] = jetdyn_G(inp[IN_TDES + i], inp[IN_TDDES + i])      # :414-415 (desired thrust!)
            c[16 + i] = jetdyn_F(T0, Td0) - dhT * T0 - dhTd * Td0              # :416-420
    else:
        Bt[12:16, 0:4] = np.eye(4)                                             # :424-425
    return A, Bj, Bt, c


def dt_schedule(cfg: Config) -> np.ndarray:
    """constraintsVSMPC.cpp:45-51 (beta1, beta2), :78-84 (per-node dt), :156-159 (warp)."""
    nS = cfg.n_iter_small
    beta2 = (cfg.period_large - nS * cfg.period_small) / (nS * (nS - 1))
    beta1 = cfg.period_small - beta2

    def warp(t):
        return beta1 * t + beta2 * t * t

    dts = np.empty(cfg.n_iter)
    for i in range(cfg.n_iter):
        dts[i] = warp(i + 1) - warp(i) if i < nS else cfg.period_large
    return dts


def joint_block_of_stage(cfg: Config, k: int) -> int:  # constraintsVSMPC.cpp:89-103
    return k if k < cfg.control_horizon else cfg.control_horizon - 1


def throttle_block_of_stage(cfg: Config, k: int) -> int:  # constraintsVSMPC.cpp:104-128
    if k < cfg.n_iter_small:
        return 0
    if k < cfg.control_horizon:
        return k - (cfg.n_iter_small - 1)
    return cfg.control_horizon - cfg.n_iter_small


def state_weight(cfg: Config) -> np.ndarray:
    """Diagonal of Q — costsVSMPC.cpp:78-93."""
    q = np.zeros(N_STATES)
    q[IDX_COM] = cfg.w_com_pos
    q[IDX_LINMOM] = cfg.w_lin_mom
    q[IDX_RPY] = cfg.w_rpy
    q[IDX_ANGMOM] = cfg.w_ang_mom
    q[IDX_EPOS] = cfg.w_com_pos_err
    q[IDX_ERPY] = cfg.w_rpy_err
    return q


def throttle_bounds(cfg: Config) -> Tuple[float, float]:
    """constraintsVSMPC.cpp:329-332."""
    return v_of_throttle(cfg.throttle_min), v_of_throttle(cfg.throttle_max)


# --------------------------------------------------------------------------------------
# Dense assembly in the reference's order (IMPCProblem.cpp:150-194):
#   costs   : ReferenceTracking, Regularization, ThrottleInitialValue, JointPositionRegularization
#   rows    : dynamics (26*nIter) | initial state (26) | throttle (4*(nIter-nIterSmall+1))
# --------------------------------------------------------------------------------------
def assemble_dense(cfg: Config, inp: np.ndarray):
    N, nS, Hc = cfg.n_iter, cfg.n_iter_small, cfg.control_horizon
    nx, nj, nt = N_STATES, N_JOINTS, N_THRUSTS
    nvar, ncon = cfg.n_var, cfg.n_con
    offJ, offV = cfg.off_joints, cfg.off_throttle

    A, Bj, Bt, c = linearize(cfg, inp)
    dts = dt_schedule(cfg)

    H = np.zeros((nvar, nvar))
    g = np.zeros(nvar)

    # ReferenceTrackingCost — costsVSMPC.cpp:166-178 ; column map :191-200
    Q = np.diag(state_weight(cfg))
    xref_win = inp[IN_XREF:IN_XREF + 12 * cfg.n_ref_cols].reshape(cfg.n_ref_cols, 12)
    for i in range(1, N + 1):
        H[i * nx:(i + 1) * nx, i * nx:(i + 1) * nx] += Q
        col = 0 if (i - 1) < nS else (i - 1) - nS
        xr = np.zeros(nx)
        xr[0:12] = xref_win[col]
        g[i * nx:(i + 1) * nx] += -Q @ xr

    # RegualarizationCost — costsVSMPC.cpp:375-409
    Wdq = np.diag(np.asarray(cfg.w_delta_joint, dtype=float))
    Wt = cfg.w_throttle * np.eye(nt)
    for i in range(Hc):
        s = offJ + i * nj
        H[s:s + nj, s:s + nj] += Wdq
    for i in range(Hc - nS):
        a = offV + i * nt
        b = offV + (i + 1) * nt
        H[a:a + nt, a:a + nt] += Wt
        H[b:b + nt, a:a + nt] -= Wt
        H[a:a + nt, b:b + nt] -= Wt
        H[b:b + nt, b:b + nt] += Wt

    # ThrottleInitialValueCost — costsVSMPC.cpp:468-487
    vprev = np.array([v_of_throttle(inp[IN_UPREV + i]) for i in range(nt)])
    H[offV:offV + nt, offV:offV + nt] += cfg.w_initial_throttle * np.eye(nt)
    g[offV:offV + nt] += -cfg.w_initial_throttle * vprev

    # JointPositionRegularizationCost — costsVSMPC.cpp:558-592
    qerr = inp[IN_QERR:IN_QERR + nj]
    for i in range(Hc):
        s = offJ + i * nj
        H[s:s + nj, s:s + nj] += cfg.w_reg_joint_pos * np.eye(nj)
        g[s:s + nj] += cfg.w_reg_joint_pos * qerr

    Ac = np.zeros((ncon, nvar))
    lo = np.zeros(ncon)
    hi = np.zeros(ncon)

    # ConstraintSystemDynamicVS — constraintsVSMPC.cpp:76-131
    I = np.eye(nx)
    for i in range(N):
        dt = dts[i]
        r = i * nx
        Ac[r:r + nx, i * nx:(i + 1) * nx] = I + dt * A
        Ac[r:r + nx, (i + 1) * nx:(i + 2) * nx] = -I
        jb = joint_block_of_stage(cfg, i)
        Ac[r:r + nx, offJ + jb * nj: offJ + (jb + 1) * nj] = dt * Bj
        tb = throttle_block_of_stage(cfg, i)
        Ac[r:r + nx, offV + tb * nt: offV + (tb + 1) * nt] = dt * Bt
        lo[r:r + nx] = -dt * c
        hi[r:r + nx] = -dt * c

    # ConstraintInitialState — IQPUtilsMPC.cpp:71-92
    r0 = N * nx
    Ac[r0:r0 + nx, 0:nx] = I
    lo[r0:r0 + nx] = inp[IN_X0:IN_X0 + nx]
    hi[r0:r0 + nx] = inp[IN_X0:IN_X0 + nx]

    # ThrottleConstraint — constraintsVSMPC.cpp:338-365 (rows beyond the filled blocks stay 0 in [0,0])
    r1 = r0 + nx
    vmin, vmax = throttle_bounds(cfg)
    hold = inp[IN_HOLD] != 0.0
    for i in range(cfg.n_vblocks):
        Ac[r1 + i * nt: r1 + (i + 1) * nt, offV + i * nt: offV + (i + 1) * nt] = np.eye(nt)
        if hold and i == 0:
            lo[r1:r1 + nt] = vprev
            hi[r1:r1 + nt] = vprev
        else:
            lo[r1 + i * nt: r1 + (i + 1) * nt] = vmin
            hi[r1 + i * nt: r1 + (i + 1) * nt] = vmax
    return H, g, Ac, lo, hi


# --------------------------------------------------------------------------------------
# Exact QP optimum (replaces the un-vendored OSQP call, IMPCProblem.cpp:279) + certificate
# --------------------------------------------------------------------------------------
def _box_qp_active_set(Hr, gr, lo, hi, max_iter=500):
    """Exact strictly convex box QP by block principal pivoting with a least-index fallback
    (Judice & Pires 1994).  lo/hi may be +-inf; lo == hi pins a variable."""
    n = gr.size
    fixed = lo == hi
    state = np.zeros(n, dtype=int)  # 0 free, -1 at lower, +1 at upper
    state[fixed] = -1
    z = np.zeros(n)
    best_ninf, patience = n + 1, 10
    for it in range(max_iter):
        F = state == 0
        z[state == -1] = lo[state == -1]
        z[state == 1] = hi[state == 1]
        if F.any():
            rhs = -(gr[F] + Hr[np.ix_(F, ~F)] @ z[~F])
            z[F] = np.linalg.solve(Hr[np.ix_(F, F)], rhs)
        grad = Hr @ z + gr
        tol = 1e-12 * (1.0 + np.abs(z))
        viol_lo = F & (z < lo - tol)
        viol_hi = F & (z > hi + tol)
        gtol = 1e-10 * (1.0 + np.abs(gr).max())
        rel_lo = (state == -1) & ~fixed & (grad < -gtol)
        rel_hi = (state == 1) & ~fixed & (grad > gtol)
        infeas = viol_lo | viol_hi | rel_lo | rel_hi
        ninf = int(infeas.sum())
        if ninf == 0:
            return z, state, it + 1
        if ninf < best_ninf:
            best_ninf, patience = ninf, 10
            pick = infeas
        elif patience > 0:
            patience -= 1
            pick = infeas
        else:  # single pivot on the largest index (finite termination for P-matrices)
            pick = np.zeros(n, dtype=bool)
            pick[np.nonzero(infeas)[0].max()] = True
        state[pick & viol_lo] = -1
        state[pick & viol_hi] = 1
        state[pick & (rel_lo | rel_hi)] = 0
    raise RuntimeError("box QP active set did not terminate")


def solve_exact(cfg: Config, H, g, Ac, lo, hi):
    """Null-space elimination of the equality rows (generic dense LU on the reference-ordered
    matrices) followed by an exact active-set solve on the inputs.  Returns x, y (OSQP sign
    convention: y>0 upper-active, y<0 lower-active) and the number of active-set iterations."""
    nxs = N_STATES * (cfg.n_iter + 1)
    neq = nxs
    nvar = cfg.n_var
    nz = nvar - nxs
    Ax, Az = Ac[:neq, :nxs], Ac[:neq, nxs:]
    sol = np.linalg.solve(Ax, np.column_stack([lo[:neq], Az]))
    Xb, G = sol[:, 0], sol[:, 1:]
    Z = np.vstack([-G, np.eye(nz)])
    xp = np.concatenate([Xb, np.zeros(nz)])
    Hr = Z.T @ H @ Z
    Hr = 0.5 * (Hr + Hr.T)
    gr = Z.T @ (H @ xp + g)
    zlo = np.full(nz, -np.inf)
    zhi = np.full(nz, np.inf)
    nthr = N_THRUSTS * cfg.n_vblocks
    o = N_JOINTS * cfg.control_horizon
    zlo[o:o + nthr] = lo[neq:neq + nthr]
    zhi[o:o + nthr] = hi[neq:neq + nthr]
    z, state, iters = _box_qp_active_set(Hr, gr, zlo, zhi)
    x = xp + Z @ z
    # duals
    y = np.zeros(cfg.n_con)
    grad_r = Hr @ z + gr
    y[neq:neq + nthr] = -grad_r[o:o + nthr]
    y[:neq] = -np.linalg.solve(Ax.T, (H @ x + g)[:nxs])
    return x, y, iters


def kkt_certificate(H, g, Ac, lo, hi, x, y) -> Dict[str, float]:
    """Optimality certificate of min 1/2 x'Hx+g'x s.t. lo<=Ax<=hi, independent of how x,y were found."""
    Ax = Ac @ x
    stat = H @ x + g + Ac.T @ y
    prim = np.maximum(0.0, np.maximum(lo - Ax, Ax - hi))
    yp, ym = np.maximum(y, 0.0), np.maximum(-y, 0.0)
    ineq = lo < hi
    comp = np.where(ineq, np.maximum(yp * (hi - Ax), ym * (Ax - lo)), 0.0)
    scale = max(1.0, float(np.abs(g).max()), float(np.abs(H @ x).max()))
    return {
        "stationarity": float(np.abs(stat).max()),
        "stationarity_rel": float(np.abs(stat).max()) / scale,
        "primal": float(prim.max()),
        "complementarity": float(np.abs(comp).max()),
        "objective": float(0.5 * x @ H @ x + g @ x),
    }


def extract_outputs(cfg: Config, x: np.ndarray) -> Dict[str, np.ndarray]:
    """variableSamplingMPC.cpp:93-108,138-151 — first-move slices (node 1 for thrusts!)."""
    offJ, offV = cfg.off_joints, cfg.off_throttle
    v0 = x[offV:offV + N_THRUSTS]
    return {
        "delta_q": x[offJ:offJ + N_JOINTS].copy(),
        "v0": v0.copy(),
        "throttle": destd_throttle(v0),
        "thrust": x[N_STATES + 12:N_STATES + 16].copy(),
        "thrust_dot": x[N_STATES + 16:N_STATES + 20].copy(),
        "final_state": x[N_STATES * cfg.n_iter:N_STATES * (cfg.n_iter + 1)].copy(),
    }


def first_move_vector(cfg: Config, x: np.ndarray) -> np.ndarray:
    o = extract_outputs(cfg, x)
    return np.concatenate([o["delta_q"], o["v0"], o["throttle"], o["thrust"], o["thrust_dot"]])


# --------------------------------------------------------------------------------------
# Kinematics-derived inputs (rows a3/a4 of SURVEY.md 8a): Lambda_lin,B, Lambda_ang,B, I_G from the raw
# Robot quantities.  Record layout KIN_* = include/vsmpc.h VSMPC_KIN_*; nJ = 23 robot joints.
# --------------------------------------------------------------------------------------
KIN_NJ = 23
KIN_WRB = 0        # 9   wR_b row-major
KIN_THRUST = 9     # 4   Robot::getJetThrusts
KIN_AXES = 13      # 12  Robot::getMatrixOfJetAxes, 4x3
KIN_ARMS = 25      # 12  Robot::getMatrixOfJetArms, 4x3
KIN_JREL = 37      # 276 Robot::getRelativeJacobianJetsBodyFrame()[i].bottomRows(3), 4 x (3x23) row-major
KIN_JFRAME = 313   # 276 Robot::getJacobian(jet).topRightCorner(3, nJ), 4 x (3x23)
KIN_JCOM = 589     # 69  Robot::getJacobianCoM().topRightCorner(3, nJ)
KIN_MB = 658       # 36  Robot::getMassMatrix().block(0,0,6,6) row-major
KIN_R = 694        # 3   p_CoM - p_base
KIN_SIZE = 697
KIN_JOINT_OFFSET = 3  # systemDynamicsVSMPC.cpp:348 (middleCols(3, 8)); the name-based selector of :202-205 picks the same columns


def kinematics_terms(kin: np.ndarray, selector=None, constant: bool = False):
    """Lambda_lin,B (3x8), Lambda_ang,B (3x8), I_G (3x3).  Default: the "unfiltered" option (vs_mcp_config.xml:21) with
    the shipped robot's controlled joints 3..10.  `selector`: robot joint index of every controlled joint for Lambda_ang
    (name-based in the reference, systemDynamicsVSMPC.cpp:57-66,202-205; Lambda_lin is hard-coded to 3..10, :348).
    `constant`: jointsLambdaOption "constant" (:186-200,329-337) -- the record then carries the configure-time axes, arms
    and relative Jacobians, the relative Jacobians' top rows in the JFRAME slot and getRobot()'s thrusts in JCOM[0:4]."""
    nJ = KIN_NJ
    R = kin[KIN_WRB:KIN_WRB + 9].reshape(3, 3)
    T = kin[KIN_THRUST:KIN_THRUST + 4]
    axes = kin[KIN_AXES:KIN_AXES + 12].reshape(4, 3)
    arms = kin[KIN_ARMS:KIN_ARMS + 12].reshape(4, 3)
    Jrel = kin[KIN_JREL:KIN_JREL + 4 * 3 * nJ].reshape(4, 3, nJ)
    Jfr = kin[KIN_JFRAME:KIN_JFRAME + 4 * 3 * nJ].reshape(4, 3, nJ)
    Jcom = kin[KIN_JCOM:KIN_JCOM + 3 * nJ].reshape(3, nJ)
    Mb = kin[KIN_MB:KIN_MB + 36].reshape(6, 6)
    r = kin[KIN_R:KIN_R + 3]
    lam_lin = np.zeros((3, nJ))
    lam_ang = np.zeros((3, nJ))
    Tang = kin[KIN_JCOM:KIN_JCOM + 4] if constant else T
    for i in range(4):
        Sa = from_vec_to_skew(R.T @ axes[i])
        lam_lin -= T[i] * Sa @ Jrel[i]                                       # systemDynamicsVSMPC.cpp:329-345
        JrelCoM = Jfr[i] if constant else R.T @ (Jfr[i] - Jcom)              # :188-190 | :208-226 (getRelativeJacobianCoM)
        lam_ang -= Tang[i] * Sa @ JrelCoM                                    # :169-174 | :188-197
        lam_ang -= Tang[i] * from_vec_to_skew(R.T @ arms[i]) @ Sa @ Jrel[i]  # :176-183 | :191-197
    X = np.zeros((6, 6))                                                     # iDynTree Transform::asAdjointTransform
    X[0:3, 0:3] = R
    X[0:3, 3:6] = from_vec_to_skew(r) @ R
    X[3:6, 3:6] = R
    inertia = (X.T @ Mb @ X)[3:6, 3:6]                                       # :128-130
    o = KIN_JOINT_OFFSET
    sel = list(range(o, o + 8)) if selector is None else [int(v) for v in selector]
    return lam_lin[:, o:o + 8].copy(), lam_ang[:, sel].copy(), inertia


def solve_instance(cfg: Config, inp: np.ndarray):
    H, g, Ac, lo, hi = assemble_dense(cfg, inp)
    x, y, iters = solve_exact(cfg, H, g, Ac, lo, hi)
    return x, y, iters, (H, g, Ac, lo, hi)
