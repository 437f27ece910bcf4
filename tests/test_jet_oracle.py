"""The jet plant oracle (oracle/jet_ref.py) against golden vectors produced by RUNNING the reference's own
nn_jet_model.py on its own checkpoint (tests/golden/jet_lstm.npz, tools/gen_jet_fixtures.py): this part of the parity
chain is pinned to reference outputs.  The EKF restatement is checked for internal consistency only (casadi absent)."""
import os

import numpy as np
import pytest

from conftest import ROOT

import jet_ref

TOL = 2e-6     # float32 LSTM: different summation order / transcendental implementations than torch's CPU kernels


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "jet_lstm.npz"))


@pytest.fixture(scope="module")
def lstm(gold):
    return jet_ref.JetLSTM(gold["w_ih"], gold["w_hh"], gold["b_ih"], gold["b_hh"], gold["fc_w"], gold["fc_b"], gold["norm"])


def test_checkpoint_shapes(gold):
    assert gold["w_ih"].shape == (320, 2) and gold["w_hh"].shape == (320, 80) and gold["fc_w"].shape == (80,)
    assert gold["w_ih"].dtype == np.float32
    assert abs(gold["norm"][0] - 106.45674270279225) < 1e-12      # the checkpoint's own statistics, not JetModel.cpp's


def test_single_step_matches_reference(lstm, gold):
    T_next, T_dot, h, c = lstm.get_state(gold["step_thrust"], gold["step_throttle"], float(gold["step_dt"]))
    assert np.abs(T_next - gold["step_T_next"]).max() < 1e-4                     # thrust in N, float32 at ~250
    assert np.abs(T_dot - gold["step_T_dot"]).max() / np.abs(gold["step_T_dot"]).max() < 1e-5
    assert np.abs(h.reshape(gold["step_h"].shape) - gold["step_h"]).max() < TOL
    assert np.abs(c.reshape(gold["step_c"].shape) - gold["step_c"]).max() < TOL


def test_sequences_with_carried_state_match_reference(lstm, gold):
    tn, td, h, c, hs = lstm.get_state_sequence(gold["seq_x"], float(gold["step_dt"]))
    assert np.abs(hs - gold["seq_h_all"]).max() < 5e-6
    assert np.abs(h - gold["seq_h"]).max() < 5e-6 and np.abs(c - gold["seq_c"]).max() < 1e-5
    assert np.abs(td - gold["seq_T_dot_norm"]).max() < 1e-5 and np.abs(tn - gold["seq_T_next_norm"]).max() < 1e-6


def test_closed_loop_plant_matches_reference(lstm, gold):
    """1,500 steps at 1 kHz with the thrust fed back (errors accumulate: fp32 round-off of different kernels)."""
    T = gold["loop_T0"].copy()
    worst = 0.0
    for k in range(len(gold["loop_throttle"])):
        T, Td, _, _ = lstm.get_state(T, gold["loop_throttle"][k], float(gold["step_dt"]))
        worst = max(worst, float(np.abs(T - gold["loop_T"][k]).max()))
        assert np.abs(Td - gold["loop_T_dot"][k]).max() < 0.05, k
    assert worst < 0.02, worst                                                    # N, over a 10..222 N trajectory


def test_ekf_restatement_is_consistent(consts=None):
    c = __import__("json").load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))
    k = c["jet"]["jet_kalman_filter.py"]
    assert list(jet_ref.COEFFS) == k["coeffs"]
    assert [jet_ref.MEAN_THRUST, jet_ref.STD_THRUST, jet_ref.MEAN_THROTTLE, jet_ref.STD_THROTTLE] == [
        k["mean_thrust"], k["std_thrust"], k["mean_throttle"], k["std_throttle"]]
    rng = np.random.default_rng(3)
    dt = c["jet"]["ekf"]["timestep"]
    for _ in range(20):
        x = np.array([rng.uniform(5, 240), rng.uniform(-100, 100)])
        u = rng.uniform(0, 100)
        A = jet_ref.ekf_jacobian(x, u, dt)
        fd = np.column_stack([(jet_ref.ekf_f(x + d, u, dt) - jet_ref.ekf_f(x - d, u, dt)) / (2 * np.linalg.norm(d))
                              for d in (np.array([1e-4, 0]), np.array([0, 1e-4]))])
        assert np.abs(A - fd).max() < 1e-7
    # steady state: with measurement == prediction the update leaves the state alone and contracts P
    Q, R = np.eye(2) * c["jet"]["ekf"]["Q"], np.eye(2) * c["jet"]["ekf"]["R"]
    x, P = np.array([120.0, 0.0]), np.eye(2) * c["jet"]["ekf"]["P"]
    xp = jet_ref.ekf_f(x, 60.0, dt)
    x2, P2 = jet_ref.ekf_update(x, P, 60.0, xp, dt, Q, R)
    np.testing.assert_allclose(x2, xp, rtol=0, atol=1e-12)
    assert np.all(np.linalg.eigvalsh(0.5 * (P2 + P2.T)) > 0) and np.trace(P2) < np.trace(P + Q)
