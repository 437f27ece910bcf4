"""GPU tests of the batched jet plant / estimator, through the C-ABI of include/vsmpc_jet.h.
The LSTM path is compared with golden vectors produced by RUNNING the reference's own nn_jet_model.py on its own
checkpoint (tests/golden/jet_lstm.npz): float32, so the tolerance is float32 round-off of different transcendental /
summation implementations (1e-5 relative on the thrust rate, 1e-4 N on thrusts up to 250 N).  The EKF is compared with
the oracle's float64 restatement (unpinned: casadi absent) at 1e-12."""
import importlib
import json
import os

import numpy as np
import pytest

from conftest import PKG, ROOT

import jet_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "jet_lstm.npz"))


@pytest.fixture(scope="module")
def jet(solver_mod, gold):
    jp = importlib.import_module(PKG + ".jet_plant")
    m = jp.JetModelTotal(gold["w_ih"], gold["w_hh"], gold["b_ih"], gold["b_hh"], gold["fc_w"], gold["fc_b"], gold["norm"],
                         device=0, max_series=20000)
    yield jp, m
    m.close()


def test_nn_step_matches_reference_vectors(jet, gold):
    jp, m = jet
    Tn, Td, h, c = m.get_state(gold["step_thrust"], gold["step_throttle"], float(gold["step_dt"]), with_state=True)
    assert np.abs(Tn - gold["step_T_next"]).max() < 1e-4
    assert np.abs(Td - gold["step_T_dot"]).max() / np.abs(gold["step_T_dot"]).max() < 1e-5
    assert np.abs(h.reshape(gold["step_h"].shape) - gold["step_h"]).max() < 2e-6
    assert np.abs(c.reshape(gold["step_c"].shape) - gold["step_c"]).max() < 2e-6
    assert "libvsmpc.so" in open("/proc/self/maps").read()


def test_nn_sequences_match_reference_vectors(jet, gold):
    jp, m = jet
    tn, td, h, c = m.get_state_sequence(gold["seq_x"], float(gold["step_dt"]))
    assert np.abs(h - gold["seq_h"]).max() < 5e-6 and np.abs(c - gold["seq_c"]).max() < 1e-5
    assert np.abs(td - gold["seq_T_dot_norm"]).max() < 1e-5 and np.abs(tn - gold["seq_T_next_norm"]).max() < 1e-6


def test_closed_loop_plant_matches_reference_trajectory(jet, gold):
    """1,500 steps at 1 kHz with the thrust fed back (the sequence MujocoSim._simulate_thrust_nn_model produces)."""
    jp, m = jet
    T = gold["loop_T0"].copy()
    worst = 0.0
    for k in range(len(gold["loop_throttle"])):
        T, Td = m.get_state(T, gold["loop_throttle"][k], float(gold["step_dt"]))
        worst = max(worst, float(np.abs(T - gold["loop_T"][k]).max()))
        assert np.abs(Td - gold["loop_T_dot"][k]).max() < 0.05, k
    assert worst < 0.02, worst


def test_ekf_and_fused_plant_match_oracle(jet, gold):
    jp, m = jet
    c = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))["jet"]["ekf"]
    dt = c["timestep"]
    Q, R, P0 = np.eye(2) * c["Q"], np.eye(2) * c["R"], np.eye(2) * c["P"]
    rng = np.random.default_rng(11)
    n = 64
    x = np.stack([rng.uniform(5, 240, n), rng.uniform(-100, 100, n)], axis=1)
    u = rng.uniform(0, 100, n)
    z = x + rng.normal(0, 2.0, (n, 2))
    ekf = jp.EKFJetsTotal(m, R, Q, P0, dt, n)
    T2, Td2 = ekf.update(x[:, 0], x[:, 1], u, z[:, 0], z[:, 1])
    for i in range(n):
        xr, Pr = jet_ref.ekf_update(x[i], P0, u[i], z[i], dt, Q, R)
        assert abs(T2[i] - xr[0]) < 1e-11 * max(1, abs(xr[0])) and abs(Td2[i] - xr[1]) < 1e-11 * max(1, abs(xr[1]))
        assert np.abs(ekf.P[i].reshape(2, 2) - Pr).max() < 1e-13
    # fused plant: 4096 instances x 4 jets, 25 steps (5 MPC ticks of 5 x 1 ms), against the oracle on a sample
    lstm = jet_ref.JetLSTM(gold["w_ih"], gold["w_hh"], gold["b_ih"], gold["b_hh"], gold["fc_w"], gold["fc_b"], gold["norm"])
    N, steps = 4096 * 4, 25
    T0 = rng.uniform(20, 220, N).astype(np.float32)
    thr = rng.uniform(10, 95, N).astype(np.float32)
    x0 = np.stack([T0.astype(float), np.zeros(N)], axis=1)
    plant = jp.JetPlant(m, R, Q, dt)
    Tn, xe, Pe, log = plant.run(T0, x0, np.tile(P0.reshape(1, 4), (N, 1)), thr, steps, log=True)
    assert np.isfinite(xe).all() and np.isfinite(Pe).all()
    sample = rng.choice(N, size=48, replace=False)
    Tn_r, xe_r, Pe_r, log_r = jet_ref.plant_run(lstm, T0[sample], x0[sample], np.tile(P0[None], (48, 1, 1)), thr[sample], steps, dt, Q, R)
    assert np.abs(Tn[sample] - Tn_r).max() < 2e-3                   # float32 NN thrust fed back 25 times
    assert np.abs(xe[sample] - xe_r).max() < 5e-3 and np.abs(log[:, sample] - log_r).max() < 5e-3
    assert np.abs(Pe[sample] - Pe_r).max() < 1e-7                    # P depends on the state only through the Jacobian at the prediction
    # a held throttle drives every estimate towards the NN plant's thrust
    Tn2, xe2, _, _ = plant.run(Tn, xe, Pe, thr, 400)
    assert np.median(np.abs(xe2[:, 0] - Tn2)) < np.median(np.abs(xe[:, 0] - Tn)) + 1e-9


def test_argument_checks(jet, gold):
    jp, m = jet
    with pytest.raises(Exception):
        m.get_state(np.zeros(30000, np.float32), np.zeros(30000, np.float32), 0.001)     # more series than allocated
