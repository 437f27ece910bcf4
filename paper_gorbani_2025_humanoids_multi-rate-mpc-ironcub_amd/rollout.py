"""Closed-loop batched rollout: host side of vsmpc_rollout_* in include/vsmpc.h.

`ClosedLoopRollout` keeps `batch` plant states resident in HBM and advances them tick by tick entirely on the GPU
(record builder -> MPC solve -> first move + plant integration); only the optional log rows come back to the host.
It exposes the tick state machine the reference hides in its plugins (20-tick throttle hold, reference window cursor,
alpha-gravity cursor: SURVEY.md A.7) and the harness loop of src/variable_sampling_mpc.py:49-152.

`make_plant` synthesises a well-posed jet humanoid for it: four jets (two on the arms, moved by the 8 controlled arm
joints, two on the back), symmetric about the CoM so that equal thrusts hover.  The reference takes these quantities
from a URDF + MuJoCo (out of scope); nothing here is read from the reference's model files.
All numerics run in libvsmpc.so (HIP); there is no CPU path in this module.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _lib
from . import layout as L
from .jet_model import JetModel
from .solver import BatchedVSMPC, _ptr
from .synth import GRAVITY, NOMINAL_MASS, alpha_gravity_profile, takeoff_profile

_JET = JetModel()


def make_plant(cfg: L.MPCConfig, batch: int, *, workload: str = "hover", seed0: int = 4321, first_index: int = 0):
    """(state[batch, PLANT_STATE], params[batch, PLANT_PARAMS]) for `workload`:
    "hover"       start near the hover equilibrium (sigma as configs[1]), trajectory offsets zero
    "takeoff"     start on the reference at a random tick of the flight part (19 s .. 35 s) of the take-off timeline
                  (configs[2]); the plant has no ground contact, so the supported phase before lift-off is left out
    "montecarlo"  hover with 4x wider initial scatter and a disturbance push of up to 50 N / 30 Nm lasting 0.1 s
                  that starts within the first second (configs[3])
    Every instance is drawn from its own generator seeded seed0 + index, so any slice can be rebuilt on any rank."""
    sigma = {"hover": 1.0, "takeoff": 0.2, "montecarlo": 4.0}[workload]
    state = np.zeros((batch, L.PLANT_STATE))
    params = np.zeros((batch, L.PLANT_PARAMS))
    # joint rotation axes of the two 4-dof arms (shoulder pitch / roll / yaw, elbow) in the body frame
    joint_axes = np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 1.0, 0.0]])
    for b in range(batch):
        rng = np.random.default_rng(seed0 + first_index + b)
        mass = float(np.float32(NOMINAL_MASS + rng.normal(0.0, 0.5)))
        inertia_B = np.diag([8.0, 7.0, 2.0]) + 0.05 * np.eye(3) * rng.uniform(0.0, 1.0)
        # jets: arm jets at (+0.10, +-0.30, 0.25), back jets at (-0.10, +-0.10, 0.20) from the CoM; axes ~ +z_B
        arms = np.array([[0.10, 0.30, 0.25], [0.10, -0.30, 0.25], [-0.10, 0.10, 0.20], [-0.10, -0.10, 0.20]])
        arms = arms + rng.normal(0.0, 0.003, size=(4, 3))
        shoulders = np.array([[0.0, 0.18, 0.35], [0.0, -0.18, 0.35]])
        axes = np.zeros((4, 3))
        for i in range(4):
            tilt = math.radians(2.0) * rng.uniform(0.0, 1.0)
            az = rng.uniform(0.0, 2 * math.pi)
            axes[i] = [math.sin(tilt) * math.cos(az), math.sin(tilt) * math.sin(az), math.cos(tilt)]
        A0 = np.zeros((6, 4))
        for i in range(4):
            A0[0:3, i] = axes[i]
            A0[3:6, i] = np.cross(arms[i], axes[i])
        DJ = np.zeros((8, 6, 4))   # dA_mom/dq_j: joint j of arm a rotates jet a's axis and lever arm about n_j
        for j in range(8):
            a, n = j // 4, joint_axes[j % 4]
            da = np.cross(n, axes[a])
            dr = np.cross(n, arms[a] - shoulders[a])
            DJ[j, 0:3, a] = da
            DJ[j, 3:6, a] = np.cross(dr, axes[a]) + np.cross(arms[a], da)
        p_init = np.array([0.0, 0.0, 1.0])
        rpy_init = rng.normal(0.0, 0.01, size=3)

        tick0 = int(rng.integers(3800, 7001)) if workload == "takeoff" else int(rng.integers(0, cfg.ratio))
        t0 = tick0 * cfg.period_mpc
        if workload == "takeoff":
            alpha = alpha_gravity_profile(t0)
            # start ON the reference the loop tracks at that tick: column 0 of the reference's FIFO window, i.e. the
            # trajectory sample pushed one window length (10 columns = 1 s) earlier (costsVSMPC.cpp:121-165)
            col0 = max(0, 1 + tick0 // cfg.ratio - (cfg.n_ref_cols - 1))
            dp, dv = takeoff_profile(col0 * cfg.period_large)
        else:
            alpha, dp, dv = 1.0, np.zeros(3), np.zeros(3)
        rpy = rpy_init + rng.normal(0.0, 0.05 * sigma, size=3)
        p = p_init + dp + rng.normal(0.0, 0.05 * sigma, size=3)
        T = np.maximum(alpha * mass * GRAVITY / 4.0 * (1.0 + rng.normal(0.0, 0.05 * sigma, size=4)), 5.0)
        u = np.clip([float(_JET.steady_state_throttle(Ti)) for Ti in T], 0.0, 100.0)

        s = state[b]
        s[L.PS_P:L.PS_P + 3] = p
        s[L.PS_HLIN:L.PS_HLIN + 3] = mass * (dv + rng.normal(0.0, 0.02 * sigma, size=3))
        s[L.PS_RPY:L.PS_RPY + 3] = rpy
        s[L.PS_HANG:L.PS_HANG + 3] = rng.normal(0.0, 0.2 * sigma, size=3)
        s[L.PS_T:L.PS_T + 4] = T
        s[L.PS_TD:L.PS_TD + 4] = 0.0
        s[L.PS_Q:L.PS_Q + 8] = 0.0
        s[L.PS_U:L.PS_U + 4] = u
        s[L.PS_TDES:L.PS_TDES + 4] = T
        s[L.PS_TDDES:L.PS_TDDES + 4] = 0.0
        # jet plant option (set_jet_plant): the NN thrust and the EKF estimates start on the plant thrust, P = 0.1 I
        # (ironcub_mujoco_simulator.py:54-57); the polynomial jet plant ignores these fields
        s[L.PS_TNN:L.PS_TNN + 4] = np.asarray(T, dtype=np.float32)
        s[L.PS_EST:L.PS_EST + 8:2] = T
        s[L.PS_EKFP:L.PS_EKFP + 16] = np.tile(0.1 * np.eye(2).reshape(-1), 4)

        q = params[b]
        q[L.PP_MASS] = mass
        q[L.PP_INERTIA_B:L.PP_INERTIA_B + 9] = inertia_B.reshape(-1)
        q[L.PP_AMOM0:L.PP_AMOM0 + 24] = A0.reshape(-1)
        q[L.PP_DJ:L.PP_DJ + 192] = DJ.reshape(-1)
        q[L.PP_QREF0:L.PP_QREF0 + 8] = 0.0
        q[L.PP_PINIT:L.PP_PINIT + 3] = p_init
        q[L.PP_RPYINIT:L.PP_RPYINIT + 3] = rpy_init
        if workload == "montecarlo":
            v = rng.normal(size=3)
            w = rng.normal(size=3)
            q[L.PP_DIST_F:L.PP_DIST_F + 3] = rng.uniform(0.0, 50.0) * v / np.linalg.norm(v)
            q[L.PP_DIST_TAU:L.PP_DIST_TAU + 3] = rng.uniform(0.0, 30.0) * w / np.linalg.norm(w)
            ts = t0 + rng.uniform(0.0, 1.0)
            q[L.PP_DIST_T0], q[L.PP_DIST_T1] = ts, ts + 0.1
        q[L.PP_TICK0] = float(tick0)
    return state, params


def make_plant_tree(cfg: L.MPCConfig, batch: int, tree: dict, *, workload: str = "hover", seed0: int = 4321, first_index: int = 0):
    """Plant states / parameters for the KINEMATIC-TREE plant (ClosedLoopRollout.set_tree): make_plant's instances with the
    tree's total mass, joints at zero and initial thrusts that carry that mass; AMOM0 / DJ / INERTIA_B stay in the
    parameter record but are not read in tree mode (A_mom(q), I_B(q), Lambda come from the provider on the plant's joints)."""
    state, params = make_plant(cfg, batch, workload=workload, seed0=seed0, first_index=first_index)
    m_tree = float(np.float32(sum(tree["mass"])))
    for b in range(batch):
        scale = m_tree / params[b, L.PP_MASS]
        params[b, L.PP_MASS] = m_tree
        state[b, L.PS_HLIN:L.PS_HLIN + 3] *= scale
        T = state[b, L.PS_T:L.PS_T + 4] * scale
        u = np.clip([float(_JET.steady_state_throttle(Ti)) for Ti in T], 0.0, 100.0)
        state[b, L.PS_T:L.PS_T + 4] = T
        state[b, L.PS_U:L.PS_U + 4] = u
        state[b, L.PS_TDES:L.PS_TDES + 4] = T
        state[b, L.PS_TNN:L.PS_TNN + 4] = np.asarray(T, dtype=np.float32)
        state[b, L.PS_EST:L.PS_EST + 8:2] = T
    return state, params


def make_trajectory(cfg: L.MPCConfig, workload: str = "hover", horizon_s: float = 60.0):
    """(pos[n,3], vel[n,3], alpha[m], alpha_dt): CoM offsets / velocities sampled every periodMPCLargeSteps and the
    alpha-gravity profile sampled at 10 Hz, the rates of the reference's MAT trajectories."""
    n = int(round(horizon_s / cfg.period_large)) + 1
    pos, vel = np.zeros((n, 3)), np.zeros((n, 3))
    alpha_dt = 0.1
    m = int(round(horizon_s / alpha_dt)) + 1
    alpha = np.ones(m)
    if workload == "takeoff":
        for i in range(n):
            pos[i], vel[i] = takeoff_profile(i * cfg.period_large)
        alpha = np.array([alpha_gravity_profile(i * alpha_dt) for i in range(m)])
    return pos, vel, alpha, alpha_dt


def load_reference_trajectories(npz_path: str):
    """(pos[n,3], vel[n,3], alpha[m], alpha_dt) from an npz holding the reference's own trajectory arrays
    (`positionCoM`, `velocityCoM` [n,3], `alphaGravity` [m], `trajectory_fps`, `alphaGravity_fps`): the content of
    src/trajectories/{minimumJerkTrajectory,alphaGravity}.mat as plain arrays (tools/gen_reference_constants.py writes
    tests/golden/reference_trajectories.npz; MAT-7.3 reading itself stays outside the path)."""
    d = np.load(npz_path)
    pos = np.ascontiguousarray(d["positionCoM"], dtype=np.float64)
    vel = np.ascontiguousarray(d["velocityCoM"], dtype=np.float64)
    alpha = np.ascontiguousarray(d["alphaGravity"], dtype=np.float64).reshape(-1)
    return pos, vel, alpha, 1.0 / float(d["alphaGravity_fps"].reshape(-1)[0])


class ClosedLoopRollout:
    """`batch` closed loops resident on one GPU.  reset() uploads plant states, run(ticks) advances them."""

    def __init__(self, cfg: L.MPCConfig, batch: int, traj_pos, traj_vel, traj_alpha, alpha_dt: float, device: int = 0):
        self.cfg = cfg
        self.batch = batch
        self.mpc = BatchedVSMPC(cfg, device=device, max_batch=batch)
        self.lib = self.mpc.lib
        pos = np.ascontiguousarray(traj_pos, dtype=np.float64)
        vel = np.ascontiguousarray(traj_vel, dtype=np.float64)
        alpha = np.ascontiguousarray(traj_alpha, dtype=np.float64)
        if pos.shape != vel.shape or pos.ndim != 2 or pos.shape[1] != 3:
            raise ValueError("traj_pos / traj_vel must be [n, 3]")
        self.n_traj = pos.shape[0]
        self._r = ctypes.c_void_p()
        _lib.check(self.lib.vsmpc_rollout_create(self.mpc._h, batch, _ptr(pos), _ptr(vel), pos.shape[0], _ptr(alpha),
                                                 alpha.shape[0], float(alpha_dt), ctypes.byref(self._r)),
                   "vsmpc_rollout_create")

    def close(self):
        if getattr(self, "_r", None) is not None and self._r:
            self.lib.vsmpc_rollout_destroy(self._r)
            self._r = None
        if getattr(self, "mpc", None) is not None:
            self.mpc.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, state: np.ndarray, params: np.ndarray):
        state = np.ascontiguousarray(state, dtype=np.float64)
        params = np.ascontiguousarray(params, dtype=np.float64)
        if state.shape != (self.batch, L.PLANT_STATE) or params.shape != (self.batch, L.PLANT_PARAMS):
            raise ValueError("state / params shape does not match the rollout batch")
        _lib.check(self.lib.vsmpc_rollout_reset(self._r, _ptr(state), _ptr(params)), "vsmpc_rollout_reset")

    def set_attitude_tracks(self, traj_rpy=None, traj_rpy_dot=None):
        """RPY / RPYDot tracks of the position trajectory (vsmpc_rollout_set_attitude_tracks; costsVSMPC.cpp:110-112,
        141-146), [n_traj, 3] each or None (all zero, the shipped files).  Call reset() afterwards."""
        a = None if traj_rpy is None else np.ascontiguousarray(traj_rpy, dtype=np.float64)
        b = None if traj_rpy_dot is None else np.ascontiguousarray(traj_rpy_dot, dtype=np.float64)
        for t in (a, b):
            if t is not None and t.shape != (self.n_traj, 3):
                raise ValueError(f"attitude tracks must be [{self.n_traj}, 3]")
        _lib.check(self.lib.vsmpc_rollout_set_attitude_tracks(self._r, _ptr(a), _ptr(b)), "vsmpc_rollout_set_attitude_tracks")

    def set_tree(self, tree: dict | None):
        """Kinematic-tree plant (vsmpc_rollout_set_tree): `tree` = a robot_tree dictionary (None = the parametric plant).
        The plant's A_mom(q), I_B(q) and the MPC's Lambda terms then all come from the kinematics provider on the plant's
        own joints.  Call reset() afterwards; params[:, PP_MASS] must be the tree's total mass."""
        from . import robot_tree as RT
        self._ctree = None if tree is None else RT.to_c(tree)    # (kept alive; the library copies it anyway)
        _lib.check(self.lib.vsmpc_rollout_set_tree(self._r, None if tree is None else ctypes.byref(self._ctree)),
                   "vsmpc_rollout_set_tree")

    def set_jet_plant(self, jet_model=None, Q=None, R=None):
        """Jet plant option (vsmpc_rollout_set_jet_plant): `jet_model` = a jet_plant.JetModelTotal (the LSTM thrust model;
        None = back to the polynomial jet plant); Q, R = EKF covariances, default 0.1 I / 0.5 I
        (ironcub_mujoco_simulator.py:54-56).  Call reset() afterwards."""
        Q = np.ascontiguousarray(0.1 * np.eye(2) if Q is None else Q, dtype=np.float64)
        R = np.ascontiguousarray(0.5 * np.eye(2) if R is None else R, dtype=np.float64)
        self._jet_model = jet_model            # keeps the jet handle alive as long as the rollout uses it
        _lib.check(self.lib.vsmpc_rollout_set_jet_plant(self._r, None if jet_model is None else jet_model._h, _ptr(Q), _ptr(R)),
                   "vsmpc_rollout_set_jet_plant")

    def run(self, ticks: int, log: bool = True, stream=None):
        """Advances every instance by `ticks` MPC periods.  Returns the log [ticks, batch, ROLLOUT_LOG] or None."""
        out = np.empty((ticks, self.batch, L.ROLLOUT_LOG)) if log else None
        s = None if stream is None else ctypes.c_void_p(stream.cuda_stream)
        _lib.check(self.lib.vsmpc_rollout_run(self._r, int(ticks), _ptr(out), s), "vsmpc_rollout_run")
        return out

    def state(self) -> np.ndarray:
        out = np.empty((self.batch, L.PLANT_STATE))
        _lib.check(self.lib.vsmpc_rollout_get_state(self._r, _ptr(out)), "vsmpc_rollout_get_state")
        return out

    def next_records(self) -> np.ndarray:
        """Records the next tick will solve (after reset: tick 0; after run(k): tick k)."""
        out = np.empty((self.batch, self.cfg.n_in))
        _lib.check(self.lib.vsmpc_rollout_get_records(self._r, _ptr(out)), "vsmpc_rollout_get_records")
        return out
