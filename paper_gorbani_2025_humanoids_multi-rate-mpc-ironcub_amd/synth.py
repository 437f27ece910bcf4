"""Synthetic "iRonCub-like" workloads for the batched MPC path.

The reference feeds its MPC from a MuJoCo simulation of a URDF that is not available here
(src/mujoco_lib/ironcub_mujoco_simulator.py:318-346 -> utils/src/Robot.cpp:198-335), so the
kinematics-derived quantities (A_mom, Lambda_lin, Lambda_ang, I_G) are synthesised with the
magnitudes listed in SURVEY.md 8(d); they are NOT taken from the URDF.  Every instance i is
generated from its own generator seeded `seed0 + i`, so any slice of a batch can be rebuilt on
any rank without communication (BASELINE.json configs[3]: per-rank seed offsets).
"""
from __future__ import annotations

import math

import numpy as np

from . import layout as L
from .jet_model import JetModel

GRAVITY = 9.81
NOMINAL_MASS = 70.0
_JET = JetModel()


def _rpy_to_rot(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rz = np.array([[cy, -sy, 0.0], [sy, cy, 0.0], [0.0, 0.0, 1.0]])
    Ry = np.array([[cp, 0.0, sp], [0.0, 1.0, 0.0], [-sp, 0.0, cp]])
    Rx = np.array([[1.0, 0.0, 0.0], [0.0, cr, -sr], [0.0, sr, cr]])
    return Rz @ Ry @ Rx


def _w_matrix(rpy):  # systemDynamicsVSMPC.cpp:133-139
    r, p = rpy[0], rpy[1]
    W = np.zeros((3, 3))
    W[0, 0] = 1.0
    W[1, 1] = math.cos(r)
    W[2, 1] = -math.sin(r)
    W[0, 2] = -math.sin(p)
    W[1, 2] = math.cos(p) * math.sin(r)
    W[2, 2] = math.cos(r) * math.cos(p)
    return W


def _unit(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def alpha_gravity_profile(t):
    """Shape of src/trajectories/alphaGravity.mat: 0.08 until 2 s, smooth ramp to
    1.0 at 20 s, then 1.0.  Re-synthesised (minimum-jerk blend), not read from the MAT file."""
    if t <= 2.0:
        return 0.08
    if t >= 20.0:
        return 1.0
    s = (t - 2.0) / 18.0
    return 0.08 + 0.92 * (10 * s ** 3 - 15 * s ** 4 + 6 * s ** 5)


def takeoff_profile(t):
    """CoM offset / velocity of a take-off: rest until 20 s, then a minimum-jerk climb to 2.5 m
    by 35 s (SURVEY.md A.6 end point (0,0,2.5)); lateral motion omitted."""
    if t <= 20.0:
        return np.zeros(3), np.zeros(3)
    if t >= 35.0:
        return np.array([0.0, 0.0, 2.5]), np.zeros(3)
    s = (t - 20.0) / 15.0
    z = 2.5 * (10 * s ** 3 - 15 * s ** 4 + 6 * s ** 5)
    zd = 2.5 * (30 * s ** 2 - 60 * s ** 3 + 30 * s ** 4) / 15.0
    return np.array([0.0, 0.0, z]), np.array([0.0, 0.0, zd])


def make_instance(cfg: L.MPCConfig, seed: int, index: int, *, sigma_scale: float = 1.0,
                  takeoff: bool = False) -> np.ndarray:
    rng = np.random.default_rng(seed)
    s = sigma_scale
    inp = np.zeros(cfg.n_in)

    mass = float(np.float32(NOMINAL_MASS + rng.normal(0.0, 0.5)))  # Robot.h:338 stores a float
    rpy = rng.normal(0.0, 0.05 * s, size=3)
    R = _rpy_to_rot(rpy)
    omega_B = rng.normal(0.0, 0.1 * s, size=3)
    p_nom = np.array([0.0, 0.0, 1.0])
    p = p_nom + rng.normal(0.0, 0.05 * s, size=3)
    h_lin = mass * rng.normal(0.0, 0.05 * s, size=3)
    h_ang = rng.normal(0.0, 0.5 * s, size=3)

    # jets: force directions ~ +z_B tilted <= 15 deg; arm jets +-0.3 m lateral, back jets +-0.1 m
    arms = np.array([[0.02, 0.30, 0.25], [0.02, -0.30, 0.25], [-0.15, 0.10, 0.20], [-0.15, -0.10, 0.20]])
    arms = arms + rng.normal(0.0, 0.01, size=(4, 3))
    axes = np.zeros((4, 3))
    for i in range(4):
        tilt = math.radians(15.0) * rng.uniform(0.0, 1.0)
        az = rng.uniform(0.0, 2 * math.pi)
        axes[i] = [math.sin(tilt) * math.cos(az), math.sin(tilt) * math.sin(az), math.cos(tilt)]
    Amom = np.zeros((6, 4))
    for i in range(4):
        Amom[0:3, i] = axes[i]
        Amom[3:6, i] = np.cross(arms[i], axes[i])

    inertia_B = np.diag([8.0, 7.0, 2.0]) + 0.05 * np.eye(3) * rng.uniform(0.0, 1.0)
    jitter = rng.normal(0.0, 0.05, size=(3, 3))
    inertia = R @ inertia_B @ R.T + jitter @ jitter.T

    # tick placement on the take-off timeline
    if takeoff:
        tick = int(rng.integers(0, 7001))
        t = tick * cfg.period_mpc
        alpha = alpha_gravity_profile(t)
    else:
        tick, t, alpha = index, 0.0, 1.0

    T0 = np.maximum(alpha * mass * GRAVITY / 4.0 * (1.0 + rng.normal(0.0, 0.05 * s, size=4)), 5.0)
    Td0 = rng.normal(0.0, 5.0 * s, size=4)
    u_prev = np.array([float(_JET.steady_state_throttle(Ti)) for Ti in T0]) + rng.normal(0.0, 1.0, size=4)
    u_prev = np.clip(u_prev, 0.0, 100.0)
    T_des = T0 * (1.0 + rng.normal(0.0, 0.01, size=4))
    Td_des = rng.normal(0.0, 2.0, size=4)

    Llin = rng.normal(0.0, 50.0, size=(3, 8))
    Lang = rng.normal(0.0, 15.0, size=(3, 8))
    rpy_init = rng.normal(0.0, 0.01, size=3)
    q_err = rng.normal(0.0, 0.02, size=8)

    if takeoff:  # disturbance impulse on the momentum
        F = rng.uniform(0.0, 50.0) * _unit(rng)
        tau = rng.uniform(0.0, 30.0) * _unit(rng)
        h_lin = h_lin + R.T @ F * 0.1
        h_ang = h_ang + R.T @ tau * 0.1

    # reference window (costsVSMPC.cpp:103-113,124-160): one column per large step
    W = _w_matrix(rpy)
    xref = np.zeros((cfg.n_ref_cols, 12))
    for j in range(cfg.n_ref_cols):
        if takeoff:
            dp, dv = takeoff_profile(t + j * cfg.period_large)
        else:
            dp, dv = np.zeros(3), np.zeros(3)
        xref[j, 0:3] = p_nom + dp
        xref[j, 3:6] = R.T @ (mass * dv)            # costsVSMPC.cpp:107-109
        xref[j, 6:9] = rpy_init                      # m_initialRPY + trajectory RPY (all zero, A.6)
        xref[j, 9:12] = inertia @ W @ np.zeros(3)    # costsVSMPC.cpp:111-112 (RPYDot trajectory is zero)
    p_ref = xref[0, 0:3]
    rpy_ref = xref[0, 6:9]

    x0 = np.zeros(L.N_STATES)
    x0[0:3] = p
    x0[3:6] = h_lin
    x0[6:9] = rpy                                    # unwrapped RPY == RPY for |rpy| < pi
    x0[9:12] = h_ang
    x0[12:16] = T0
    x0[16:20] = Td0
    x0[20:23] = p - p_ref                            # constraintsVSMPC.cpp:225-226
    x0[23:26] = rpy - rpy_ref                        # constraintsVSMPC.cpp:227-228

    inp[L.IN_X0:L.IN_X0 + 26] = x0
    inp[L.IN_MASS] = mass
    inp[L.IN_WRB:L.IN_WRB + 9] = R.reshape(-1)
    inp[L.IN_OMEGA:L.IN_OMEGA + 3] = omega_B
    inp[L.IN_ALPHA] = alpha
    inp[L.IN_GRAV:L.IN_GRAV + 3] = [0.0, 0.0, -GRAVITY]
    inp[L.IN_AMOM:L.IN_AMOM + 24] = Amom.reshape(-1)
    inp[L.IN_LLIN:L.IN_LLIN + 24] = Llin.reshape(-1)
    inp[L.IN_LANG:L.IN_LANG + 24] = Lang.reshape(-1)
    inp[L.IN_INERTIA:L.IN_INERTIA + 9] = inertia.reshape(-1)
    inp[L.IN_RPY:L.IN_RPY + 3] = rpy
    inp[L.IN_PREF:L.IN_PREF + 3] = p_ref
    inp[L.IN_RPYINIT:L.IN_RPYINIT + 3] = rpy_init
    inp[L.IN_T0:L.IN_T0 + 4] = T0
    inp[L.IN_TD0:L.IN_TD0 + 4] = Td0
    inp[L.IN_UPREV:L.IN_UPREV + 4] = u_prev
    inp[L.IN_TDES:L.IN_TDES + 4] = T_des
    inp[L.IN_TDDES:L.IN_TDDES + 4] = Td_des
    inp[L.IN_QERR:L.IN_QERR + 8] = q_err
    inp[L.IN_HOLD] = 1.0 if (tick % cfg.ratio) != 0 else 0.0   # constraintsVSMPC.cpp:351,366-372
    inp[L.IN_XREF:L.IN_XREF + 12 * cfg.n_ref_cols] = xref.reshape(-1)
    return inp


def make_batch(cfg: L.MPCConfig, batch: int, *, workload: str = "hover", seed0: int = 1234,
               first_index: int = 0) -> np.ndarray:
    """[batch, cfg.n_in] float64, C-contiguous: the array `vsmpc_solve_batch` consumes.

    workload: "hover" (configs[1]), "takeoff" (configs[2]), "montecarlo" (configs[3]: hover with
    4x wider sigma).  configs[4] is "hover" with layout.horizon2x_config().
    """
    sigma = {"hover": 1.0, "takeoff": 1.0, "montecarlo": 4.0}[workload]
    out = np.empty((batch, cfg.n_in))
    for b in range(batch):
        i = first_index + b
        out[b] = make_instance(cfg, seed0 + i, i, sigma_scale=sigma, takeoff=(workload == "takeoff"))
    return out
