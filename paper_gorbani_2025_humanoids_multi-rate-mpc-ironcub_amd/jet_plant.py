"""Host-side mirror of the reference's jet plant / estimator classes on top of include/vsmpc_jet.h.

`JetModelTotal`   src/mujoco_lib/nn_jet_model.py:33-109   same constructor idea (weights + normalisation metadata) and
                  `get_state(thrusts, throttles, dt)`, for any number of series instead of one jet at a time
`EKFJetsTotal`    src/mujoco_lib/jet_kalman_filter.py:68-81   `update(T, TDot, u, TMeas, TDotMeas)` with per-series covariance
`JetPlant`        the fused 1 kHz plant step of MujocoSim.step (ironcub_mujoco_simulator.py:128-133) for Monte-Carlo loops

All numerics run in libvsmpc.so (HIP); there is no CPU path in this module.  Weights are passed as arrays (the
checkpoint format itself, torch.load of model_7.pth, stays outside)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class JetModelTotal:
    def __init__(self, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b, norm, device: int = 0, max_series: int = 16384):
        self.lib = _lib.load()
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.hidden = int(np.asarray(w_hh).shape[1])
        self.max_series = max_series
        self._h = ctypes.c_void_p()
        w = [f32(w_ih), f32(w_hh), f32(b_ih), f32(b_hh), f32(np.asarray(fc_w).reshape(-1)), f32(np.asarray(fc_b).reshape(-1))]
        nm = np.ascontiguousarray(norm, dtype=np.float64)
        _lib.check(self.lib.vsmpc_jet_create(*[_p(a) for a in w], _p(nm), self.hidden, device, max_series,
                                             ctypes.byref(self._h)), "vsmpc_jet_create")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vsmpc_jet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_state(self, current_thrusts, current_throttles, dt, with_state=False):
        """(next_T, next_T_dot) float32, same shape as the inputs (nn_jet_model.py:86-109)."""
        T = np.ascontiguousarray(current_thrusts, dtype=np.float32)
        u = np.ascontiguousarray(current_throttles, dtype=np.float32)
        n = T.size
        Tn, Td = np.empty(n, np.float32), np.empty(n, np.float32)
        h = np.empty((n, self.hidden), np.float32) if with_state else None
        c = np.empty((n, self.hidden), np.float32) if with_state else None
        _lib.check(self.lib.vsmpc_jet_nn_step(self._h, _p(T.reshape(-1)), _p(u.reshape(-1)), n, float(dt), _p(Tn), _p(Td),
                                              _p(h), _p(c)), "vsmpc_jet_nn_step")
        out = (Tn.reshape(T.shape), Td.reshape(T.shape))
        return out + (h, c) if with_state else out

    def get_state_sequence(self, x, dt):
        """NeuralJetModel.get_state on normalised sequences x [n, L, 2] (state carried): T_next_norm, T_dot_norm, h_n, c_n."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n, L, _ = x.shape
        tn, td = np.empty(n, np.float32), np.empty(n, np.float32)
        h, c = np.empty((n, self.hidden), np.float32), np.empty((n, self.hidden), np.float32)
        _lib.check(self.lib.vsmpc_jet_nn_sequence(self._h, _p(x), n, L, float(dt), _p(tn), _p(td), _p(h), _p(c)),
                   "vsmpc_jet_nn_sequence")
        return tn, td, h, c


class EKFJetsTotal:
    """jet_kalman_filter.py:68-81 with one covariance per series; R, Q, P are 2x2 (ironcub_mujoco_simulator.py:54-57)."""

    def __init__(self, model: JetModelTotal, R, Q, P, dt, n_series: int):
        self.model, self.dt = model, float(dt)
        self.R = np.ascontiguousarray(R, dtype=np.float64).reshape(4)
        self.Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(4)
        self.P = np.tile(np.asarray(P, dtype=np.float64).reshape(1, 4), (n_series, 1))

    def update(self, T, TDot, u, TMeas, TDotMeas):
        x = np.ascontiguousarray(np.stack([np.asarray(T, float).reshape(-1), np.asarray(TDot, float).reshape(-1)], axis=1))
        z = np.ascontiguousarray(np.stack([np.asarray(TMeas, float).reshape(-1), np.asarray(TDotMeas, float).reshape(-1)], axis=1))
        uu = np.ascontiguousarray(u, dtype=np.float64).reshape(-1)
        _lib.check(self.model.lib.vsmpc_jet_ekf_update(self.model._h, _p(x), _p(self.P), _p(uu), _p(z), x.shape[0], self.dt,
                                                       _p(self.Q), _p(self.R)), "vsmpc_jet_ekf_update")
        return x[:, 0].reshape(np.shape(T)), x[:, 1].reshape(np.shape(T))


class JetPlant:
    """`steps` fused plant steps (NN thrust fed back -> EKF) for n series resident in host arrays."""

    def __init__(self, model: JetModelTotal, R, Q, dt):
        self.model, self.dt = model, float(dt)
        self.R = np.ascontiguousarray(R, dtype=np.float64).reshape(4)
        self.Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(4)

    def run(self, T_nn, x_est, P, throttle, steps, log=False):
        T_nn = np.ascontiguousarray(T_nn, dtype=np.float32).copy()
        x_est = np.ascontiguousarray(x_est, dtype=np.float64).copy()
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 4).copy()
        thr = np.ascontiguousarray(throttle, dtype=np.float32)
        n = T_nn.size
        tsteps = 1 if thr.ndim == 1 else thr.shape[0]
        lg = np.empty((steps, n, 2)) if log else None
        _lib.check(self.model.lib.vsmpc_jet_plant_run(self.model._h, _p(T_nn), _p(x_est), _p(P), _p(thr), tsteps, n, int(steps),
                                                      self.dt, _p(self.Q), _p(self.R), _p(lg)), "vsmpc_jet_plant_run")
        return T_nn, x_est, P.reshape(-1, 2, 2), lg
