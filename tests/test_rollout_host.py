"""CPU tests of the closed-loop rollout's host side: layouts agree with the header, the synthetic plant hovers at
equal thrust, and the numpy model of the rollout kernels closes the loop with the oracle (tick state machine)."""
import importlib
import os
import re

import numpy as np

import rollout_model as rm
from conftest import PKG, ROOT


def _rollout():
    return importlib.import_module(PKG + ".rollout")


def test_plant_layout_matches_header(layout):
    text = open(os.path.join(ROOT, "include", "vsmpc.h")).read()
    defs = dict(re.findall(r"#define\s+VSMPC_(PS_[A-Z0-9_]+|PP_[A-Z0-9_]+|PLANT_STATE|PLANT_PARAMS|ROLLOUT_LOG)\s+(\d+)", text))
    assert len(defs) >= 25
    for key, val in defs.items():
        assert getattr(layout, key) == int(val), key


def test_synthetic_plant_is_well_posed_and_sliceable(layout):
    ro, cfg = _rollout(), layout.paper_config()
    st, pa = ro.make_plant(cfg, 6, workload="hover")
    st2, pa2 = ro.make_plant(cfg, 3, workload="hover", first_index=3)
    np.testing.assert_array_equal(st[3:], st2)
    np.testing.assert_array_equal(pa[3:], pa2)
    for b in range(6):
        m = pa[b, layout.PP_MASS]
        A = pa[b, layout.PP_AMOM0:layout.PP_AMOM0 + 24].reshape(6, 4)
        w = A @ np.full(4, m * 9.81 / 4.0)                 # wrench of equal hover thrusts, body frame
        assert abs(w[2] - m * 9.81) < 0.01 * m * 9.81       # supports the weight
        assert np.abs(w[:2]).max() < 0.03 * m * 9.81 and np.abs(w[3:]).max() < 6.0   # nearly balanced
        DJ = pa[b, layout.PP_DJ:layout.PP_DJ + 192].reshape(8, 6, 4)
        assert np.all(DJ[:4, :, 1:] == 0) and np.all(DJ[4:, :, [0, 2, 3]] == 0)       # arm joints move their own jet only
        lam = np.einsum("jrc,c->rj", DJ, st[b, layout.PS_T:layout.PS_T + 4])
        assert np.linalg.matrix_rank(lam[3:6]) == 3         # attitude authority through the joints
    stt, pat = ro.make_plant(cfg, 4, workload="takeoff")
    stm, pam = ro.make_plant(cfg, 4, workload="montecarlo")
    assert (pat[:, layout.PP_TICK0] >= 0).all() and (pam[:, layout.PP_DIST_T1] > pam[:, layout.PP_DIST_T0]).all()


def test_trajectory_shapes(layout):
    ro, cfg = _rollout(), layout.paper_config()
    pos, vel, alpha, adt = ro.make_trajectory(cfg, "takeoff", 40.0)
    assert pos.shape == vel.shape == (401, 3) and alpha.shape == (401,) and adt == 0.1
    assert alpha[0] == 0.08 and alpha[-1] == 1.0 and abs(pos[-1, 2] - 2.5) < 1e-12
    assert rm.interp_clamped(alpha, -1.0) == alpha[0] and rm.interp_clamped(alpha, 1e9) == alpha[-1]


def test_model_closed_loop_with_oracle_holds_throttle(layout, ref):
    """25 ticks of one hover instance with the numpy oracle in the loop: every solve is optimal, the throttle command
    changes only on the free tick of the 20-tick hold (constraintsVSMPC.cpp:351-372), the attitude error shrinks."""
    ro, cfg, rcfg = _rollout(), layout.paper_config(), ref.paper_config()
    st, pa = ro.make_plant(cfg, 1, workload="hover", seed0=77)
    pos, vel, alpha, adt = ro.make_trajectory(cfg, "hover", 5.0)
    s, p = st[0].copy(), pa[0]
    tick0 = int(p[layout.PP_TICK0])
    model = rm.make_tick_model(cfg, s, p, pos, vel, alpha)
    changes = []
    for tick in range(25):
        rec = rm.build_record(cfg, model, s, p)
        assert rec[layout.IN_HOLD] == (0.0 if (tick0 + tick) % cfg.ratio == cfg.ratio - 1 else 1.0)
        x, y, _, qp = ref.solve_instance(rcfg, rec)
        cert = ref.kkt_certificate(*qp, x, y)
        assert cert["stationarity_rel"] < 1e-9 and cert["primal"] < 1e-9
        u_before = s[layout.PS_U:layout.PS_U + 4].copy()
        fm = ref.first_move_vector(rcfg, x)
        model.consume(fm, 1)
        s = rm.advance(cfg, s, p, tick, fm, 1, alpha, adt)
        if np.abs(s[layout.PS_U:layout.PS_U + 4] - u_before).max() > 1e-9:
            changes.append((tick0 + tick) % cfg.ratio)
    assert changes and all(c == cfg.ratio - 1 for c in changes)
    assert np.isfinite(s).all() and np.abs(s[layout.PS_RPY:layout.PS_RPY + 3]).max() < 0.2


def test_tick_model_follows_the_reference_call_counts(layout):
    """The reference's own semantics, on the reference's own trajectory files (tests/golden/reference_trajectories.npz):
    FIFO window with a 10-column lag, push on ticks k % 20 == 19 = the throttle-release ticks, h_lin column frozen with
    the R of its push, alpha cursor one sample ahead (configure consumed sample 0), RPY unwrap across +-pi."""
    import os
    import tick_model as tm
    ro, cfg = _rollout(), layout.paper_config()
    pos, vel, alpha, adt = ro.load_reference_trajectories(os.path.join(ROOT, "tests", "golden", "reference_trajectories.npz"))
    st, pa = ro.make_plant(cfg, 1, workload="hover", seed0=5)
    s, p = st[0].copy(), pa[0].copy()
    p[layout.PP_TICK0] = 0.0
    p[layout.PP_PINIT:layout.PP_PINIT + 3] = s[layout.PS_P:layout.PS_P + 3]
    s[layout.PS_RPY + 2] = np.pi - 0.01                      # yaw just below +pi
    p[layout.PP_RPYINIT:layout.PP_RPYINIT + 3] = tm.as_rpy(s[layout.PS_RPY:layout.PS_RPY + 3])
    m = tm.ReferenceTickModel(cfg, s, p, pos, vel, alpha)
    n = cfg.n_ref_cols
    up = tm.TrajectoryManager({"alphaGravity": alpha[None, :]}, 10, 200).trajectories_map["alphaGravity"].values
    assert len(up) == 20 * (len(alpha) - 1)
    pushes = []
    for k in range(4200 + 45):
        s_k = s.copy()
        s_k[layout.PS_RPY + 2] += 0.001 * k                   # yaw drifts through +pi: wrapped measurement jumps by -2 pi
        s_k[layout.PS_RPY] = 0.02 * np.sin(0.01 * k)          # roll wobbles: R differs from push to push
        f = m.update(s_k)
        assert f["hold"] == (0.0 if k % 20 == 19 else 1.0)                               # constraintsVSMPC.cpp:351-372
        assert f["alpha"] == up[min(k + 1, len(up) - 1)][0]                              # systemDynamicsVSMPC.cpp:308-311
        ns = 1 + (k + 1) // 20                                                           # shifts so far incl. configure
        for j in range(n):
            sj = ns - (n - 1 - j)
            np.testing.assert_allclose(f["xref"][j, 0:3], p[layout.PP_PINIT:layout.PP_PINIT + 3] + pos[max(sj, 0)], rtol=0, atol=1e-15)
        if k % 20 == 19:
            pushes.append((k, f["xref"][-1, 3:6].copy(), tm.rot(s_k[layout.PS_RPY:layout.PS_RPY + 3])))
        if len(pushes) >= 2:                                  # the column pushed one release earlier kept ITS rotation
            k0, col, R0 = pushes[-2]
            if k - k0 < 20 and np.abs(vel[1 + (k0 + 1) // 20]).max() > 0:
                np.testing.assert_allclose(f["xref"][-2, 3:6], col, rtol=0, atol=0)
                np.testing.assert_allclose(col, R0.T @ (p[layout.PP_MASS] * vel[1 + (k0 + 1) // 20]), rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(f["x0"][6:9], s_k[layout.PS_RPY:layout.PS_RPY + 3], rtol=0, atol=1e-12)   # unwrapped == continuous
        assert np.abs(f["rpy"]).max() <= np.pi
    assert m.init_state.m_nTurns[2] == 1                      # one turn counted at the crossing
    assert m.cost.m_trajManager.trajectoryIndex == 1 + 4245 // 20


def test_model_jet_stage_is_the_oracle_plant_run():
    """The jet plant option of the rollout's numpy model (rollout_model.advance(jet=...)) advances thrust and estimates
    exactly like oracle/jet_ref.plant_run (the restatement of ironcub_mujoco_simulator.py:128-133,393-396) over the
    five 1 ms sub-steps of a tick with the throttle held."""
    import os
    import jet_ref
    import rollout_model as rm
    from conftest import PKG, ROOT
    import importlib
    L = importlib.import_module(PKG + ".layout")
    ro = importlib.import_module(PKG + ".rollout")
    gold = np.load(os.path.join(ROOT, "tests", "golden", "jet_lstm.npz"))
    lstm = jet_ref.JetLSTM(gold["w_ih"], gold["w_hh"], gold["b_ih"], gold["b_hh"], gold["fc_w"], gold["fc_b"], gold["norm"])
    Q, R = 0.1 * np.eye(2), 0.5 * np.eye(2)
    cfg = L.paper_config()
    st, pa = ro.make_plant(cfg, 3, workload="montecarlo")
    pos, vel, alpha, adt = ro.make_trajectory(cfg, "hover", 5.0)
    for b in range(3):
        s = st[b].copy()
        s[L.PS_U:L.PS_U + 4] = s[L.PS_U:L.PS_U + 4].astype(np.float32)   # plant_run keeps the throttle in float32
        after = rm.advance(cfg, s, pa[b], 0, np.zeros(L.FM_SIZE), 0, alpha, adt, jet=(lstm, Q, R))   # status 0: nothing consumed
        Tn, est, P, _ = jet_ref.plant_run(lstm, s[L.PS_TNN:L.PS_TNN + 4], s[L.PS_EST:L.PS_EST + 8].reshape(4, 2),
                                          s[L.PS_EKFP:L.PS_EKFP + 16].reshape(4, 2, 2), s[L.PS_U:L.PS_U + 4], 5,
                                          cfg.period_mpc / 5, Q, R)
        np.testing.assert_array_equal(after[L.PS_TNN:L.PS_TNN + 4], Tn.astype(np.float64))
        np.testing.assert_allclose(after[L.PS_EST:L.PS_EST + 8], est.reshape(-1), rtol=0, atol=1e-12)
        np.testing.assert_allclose(after[L.PS_EKFP:L.PS_EKFP + 16], P.reshape(-1), rtol=0, atol=1e-14)
        # set_thrust(self._estimated_thrust): the force the body sees is the EKF estimate (ironcub_mujoco_simulator.py:131-133)
        np.testing.assert_array_equal(after[L.PS_T:L.PS_T + 4], after[L.PS_EST:L.PS_EST + 8:2])
        np.testing.assert_array_equal(after[L.PS_TD:L.PS_TD + 4], after[L.PS_EST + 1:L.PS_EST + 8:2])
