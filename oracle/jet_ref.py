"""CPU restatement of the jet plant / estimator side of the reference's simulator.  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.

    NeuralJetModel / JetModelTotal   src/mujoco_lib/nn_jet_model.py:3-30 (LSTM(2 -> 80, batch_first) + Linear(80 -> 1),
                                     T_next = x[:, -1, 0] + fc(h_last) * dt), :64-109 (normalisation with the checkpoint's
                                     metadata, one jet at a time, ZERO initial state on every call, fp32 tensors)
    SecondOrderJetModel / EKF        src/mujoco_lib/jet_kalman_filter.py:4-81 (semi-implicit Euler step of the
                                     13-coefficient model, Jacobian by casadi AD, predict -> update with H = I);
                                     covariances of ironcub_mujoco_simulator.py:54-57 (P = Q = 0.1 I, R = 0.5 I, dt = 1 ms)
    plant step order                 src/mujoco_lib/ironcub_mujoco_simulator.py:128-133, 393-396

Parity status.  The LSTM part is PINNED: tests/golden/jet_lstm.npz holds outputs of the reference's own module run on
its own checkpoint (tools/gen_jet_fixtures.py), and tests/test_jet_oracle.py checks this restatement against them.  The
EKF part is UNPINNED: casadi is not in this image, so the Jacobian is the analytic derivative of the same step (checked
against finite differences) and nothing of the reference's EKF could be run here.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def sigmoid(x):
    return F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))


class JetLSTM:
    """torch.nn.LSTM(2, H, batch_first=True) + torch.nn.Linear(H, 1), gate order (i, f, g, o), all float32."""

    def __init__(self, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b, norm):
        self.w_ih = np.asarray(w_ih, F32)
        self.w_hh = np.asarray(w_hh, F32)
        self.b_ih = np.asarray(b_ih, F32)
        self.b_hh = np.asarray(b_hh, F32)
        self.fc_w = np.asarray(fc_w, F32).reshape(-1)
        self.fc_b = F32(np.asarray(fc_b).reshape(-1)[0])
        self.H = self.w_hh.shape[1]
        self.thrust_mean, self.thrust_std, self.throttle_mean, self.throttle_std = [float(v) for v in norm]

    def cell(self, x, h, c):
        """One LSTM step for a batch: x [n, 2], h, c [n, H] (float32)."""
        H = self.H
        gates = x @ self.w_ih.T + self.b_ih + h @ self.w_hh.T + self.b_hh
        i, f = sigmoid(gates[:, 0:H]), sigmoid(gates[:, H:2 * H])
        g, o = np.tanh(gates[:, 2 * H:3 * H], dtype=F32), sigmoid(gates[:, 3 * H:4 * H])
        c = f * c + i * g
        h = o * np.tanh(c, dtype=F32)
        return h.astype(F32), c.astype(F32)

    def get_state_sequence(self, x, dt):
        """NeuralJetModel.get_state (nn_jet_model.py:16-30): x [n, L, 2] normalised; returns T_next_norm, T_dot_norm,
        h_n, c_n and the hidden trajectory."""
        x = np.asarray(x, F32)
        n, Ls, _ = x.shape
        h = np.zeros((n, self.H), F32)
        c = np.zeros((n, self.H), F32)
        hs = np.zeros((n, Ls, self.H), F32)
        for t in range(Ls):
            h, c = self.cell(x[:, t, :], h, c)
            hs[:, t, :] = h
        T_dot_norm = (h @ self.fc_w + self.fc_b).astype(F32)
        T_next_norm = (x[:, -1, 0] + T_dot_norm * F32(dt)).astype(F32)
        return T_next_norm, T_dot_norm, h, c, hs

    def normalize(self, thrust, throttle):
        """_normalize_data_single_jet (nn_jet_model.py:64-73): arithmetic in Python floats, then a float32 tensor."""
        tn = (np.asarray(thrust, F32).astype(np.float64) - self.thrust_mean) / self.thrust_std
        un = (np.asarray(throttle, F32).astype(np.float64) - self.throttle_mean) / self.throttle_std
        return np.stack([tn, un], axis=-1).astype(F32)

    def get_state(self, thrust, throttle, dt):
        """JetModelTotal.get_state (nn_jet_model.py:86-109) for any number of series: (T_next [N], T_dot [N/s]) float32."""
        thrust = np.asarray(thrust, F32)
        shape = thrust.shape
        x = self.normalize(thrust.reshape(-1), np.asarray(throttle, F32).reshape(-1))[:, None, :]
        T_next_norm, T_dot_norm, h, c, _ = self.get_state_sequence(x, dt)
        T_next = (T_next_norm * F32(self.thrust_std) + F32(self.thrust_mean)).astype(F32)      # _denormalize_thrust
        T_dot = (T_dot_norm * F32(self.thrust_std)).astype(F32)                                # _denormalize_thrust_dot
        return T_next.reshape(shape), T_dot.reshape(shape), h, c


# ---- second-order polynomial jet model + EKF (jet_kalman_filter.py) -------------------------------------------------
COEFFS = (-4.64730485e-01, -8.13171858e+00, -6.19539230e+00, 6.61113140e-01, 1.67673231e+00, -4.83287064e-01,
          8.77996617e+00, -1.01096376e+00, -5.86442286e-01, 5.19093322e-01, -4.23782666e-01, -1.45705257e+00,
          -7.83052261e-03)                                   # jet_kalman_filter.py:6-18
MEAN_THRUST, STD_THRUST, MEAN_THROTTLE, STD_THROTTLE = 108.309, 65.793, 47.333, 31.483   # :19-22


def ekf_f(x, u, dt):
    """get_cs_f_fun (jet_kalman_filter.py:29-45): x = (T, T_dot) -> next state; T_dot first, then T with the NEW T_dot."""
    c = COEFFS
    T, Td = x
    a = (T - MEAN_THRUST) / STD_THRUST
    b = Td / STD_THRUST
    us = (u - MEAN_THROTTLE) / STD_THROTTLE
    f = c[0] + c[1] * a + c[2] * b + c[3] * a * b + c[4] * a ** 2 + c[5] * b ** 2
    g = c[6] + c[7] * a + c[8] * b + c[9] * a * b + c[10] * a ** 2 + c[11] * b ** 2
    v = us + c[12] * us ** 2
    Tdd = f + g * v
    Td_new = Td + Tdd * STD_THRUST * dt
    T_new = T + Td_new * dt
    return np.array([T_new, Td_new])


def ekf_jacobian(x, u, dt):
    """d ekf_f / d x (what casadi's jacobian gives for this function, :47-55), analytic."""
    c = COEFFS
    T, Td = x
    a = (T - MEAN_THRUST) / STD_THRUST
    b = Td / STD_THRUST
    us = (u - MEAN_THROTTLE) / STD_THROTTLE
    v = us + c[12] * us ** 2
    h_a = (c[1] + c[3] * b + 2 * c[4] * a) + (c[7] + c[9] * b + 2 * c[10] * a) * v     # d(f + g v)/d a
    h_b = (c[2] + c[3] * a + 2 * c[5] * b) + (c[8] + c[9] * a + 2 * c[11] * b) * v     # d(f + g v)/d b
    dTd_dT = dt * h_a                 # sigma dt * (1 / sigma) h_a
    dTd_dTd = 1.0 + dt * h_b
    return np.array([[1.0 + dt * dTd_dT, dt * dTd_dTd], [dTd_dT, dTd_dTd]])


def ekf_update(x, P, u, z, dt, Q, R):
    """SecondOrderJetModel.update (:57-66): the Jacobian is evaluated at the PREDICTED state; H = I."""
    x = ekf_f(np.asarray(x, float), float(u), dt)
    A = ekf_jacobian(x, float(u), dt)
    P = A @ np.asarray(P, float) @ A.T + Q
    err = np.asarray(z, float) - x
    S = P + R
    K = P @ np.linalg.inv(S)
    x = x + K @ err
    P = (np.eye(2) - K) @ P
    return x, P


def plant_run(lstm: JetLSTM, T_nn, x_est, P, throttle, steps, dt, Q, R):
    """MujocoSim.step with use_nn_jet_dynamics (ironcub_mujoco_simulator.py:128-133): per 1 ms step the NN plant advances
    its own thrust (T fed back, :393-396), then the EKF of every jet is updated with the NN's (T, T_dot) as measurement.
    T_nn [n] float32, x_est [n, 2], P [n, 2, 2], throttle [steps, n] or [n].  Returns the final values and the per-step
    estimates [steps, n, 2]."""
    T_nn = np.asarray(T_nn, F32).copy()
    x_est = np.asarray(x_est, float).copy()
    P = np.asarray(P, float).copy()
    thr = np.asarray(throttle, F32)
    log = np.zeros((steps, len(T_nn), 2))
    for k in range(steps):
        u = thr[k] if thr.ndim == 2 else thr
        T_nn, Td_nn, _, _ = lstm.get_state(T_nn, u, dt)
        for i in range(len(T_nn)):
            x_est[i], P[i] = ekf_update(x_est[i], P[i], float(u[i]), [float(T_nn[i]), float(Td_nn[i])], dt, Q, R)
        log[k] = x_est
    return T_nn, x_est, P, log
