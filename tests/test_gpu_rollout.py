"""GPU tests of the closed-loop rollout: the device's tick state machine against a model written from
the REFERENCE's plugins (tests/tick_model.py: FIFO reference window, throttle-release tick, alpha cursor, RPY unwrap),
the plant advance against its numpy model, a short closed loop against the oracle-in-the-loop model, and closed-loop
properties at batch scale whose bounds come from the linearised closed loop (tests/closed_loop_linearisation.py)."""
import importlib

import numpy as np
import pytest

import rollout_model as rm
from conftest import PKG, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ro(solver_mod):
    return importlib.import_module(PKG + ".rollout")


def _make(ro, layout, batch, workload, horizon_s=60.0):
    cfg = layout.paper_config()
    st, pa = ro.make_plant(cfg, batch, workload=workload)
    pos, vel, alpha, adt = ro.make_trajectory(cfg, "takeoff" if workload == "takeoff" else "hover", horizon_s)
    r = ro.ClosedLoopRollout(cfg, batch, pos, vel, alpha, adt, device=0)
    r.reset(st, pa)
    return cfg, st, pa, (pos, vel, alpha, adt), r


@pytest.mark.parametrize("workload", ["hover", "takeoff", "montecarlo"])
def test_record_and_advance_kernels_match_model(ro, layout, workload):
    """Three ticks of every workload, mid-trajectory starts included (PP_TICK0 up to 7000: the device fast-forwards the
    tick state, the model reaches the same state by running the reference's state machine PP_TICK0 times)."""
    B = 12
    cfg, st, pa, (pos, vel, alpha, adt), r = _make(ro, layout, B, workload)
    try:
        s_host = st.copy()
        models = [rm.make_tick_model(cfg, st[b], pa[b], pos, vel, alpha) for b in range(B)]
        recs = r.next_records()                                # record of tick 0, built on the device by reset()
        for tick in range(3):
            log = r.run(1)
            x, fm, status, iters = r.mpc.solve(recs)          # same kernel, same records -> the first move the loop used
            after = r.state()
            for b in range(B):
                rec_m = rm.build_record(cfg, models[b], s_host[b], pa[b])
                assert relerr(recs[b], rec_m) < 1e-13, (workload, tick, b)
                models[b].consume(fm[b], status[b])
                s_m = rm.advance(cfg, s_host[b], pa[b], tick, fm[b], status[b], alpha, adt)
                assert relerr(after[b], s_m) < 1e-12, (workload, tick, b)
                assert log[0, b, 14] == status[b] and log[0, b, 15] == iters[b]
                np.testing.assert_allclose(log[0, b, 0:3], after[b, 0:3], rtol=0, atol=0)
            recs = r.next_records()                            # assembled by the advance kernel from the new state
            s_host = after
    finally:
        r.close()


def test_jet_plant_option_matches_model(ro, layout):
    """N4 inside N1: the rollout with the LSTM jet plant + EKF (vsmpc_rollout_set_jet_plant) against the numpy model built
    on oracle/jet_ref.py (whose LSTM is pinned to the reference's own module, tests/golden/jet_lstm.npz): records (the
    controller sees the EKF estimates) and plant states over six ticks, then the polynomial plant again.  float32 network
    with a different summation order: 2e-6 relative on the state, records exact up to that."""
    import os
    import jet_ref
    from conftest import ROOT
    gold = np.load(os.path.join(ROOT, "tests", "golden", "jet_lstm.npz"))
    jp = importlib.import_module(PKG + ".jet_plant")
    jm = jp.JetModelTotal(gold["w_ih"], gold["w_hh"], gold["b_ih"], gold["b_hh"], gold["fc_w"], gold["fc_b"], gold["norm"],
                          device=0, max_series=64)
    lstm = jet_ref.JetLSTM(gold["w_ih"], gold["w_hh"], gold["b_ih"], gold["b_hh"], gold["fc_w"], gold["fc_b"], gold["norm"])
    Q, Rm = 0.1 * np.eye(2), 0.5 * np.eye(2)
    B = 8
    cfg, st, pa, (pos, vel, alpha, adt), r = _make(ro, layout, B, "montecarlo")
    try:
        r.set_jet_plant(jm)
        with pytest.raises(Exception):
            r.run(1)                                           # the option changes what the records hold: reset() first
        r.reset(st, pa)
        jet = (lstm, Q, Rm)
        s_host = st.copy()
        models = [rm.make_tick_model(cfg, rm.measured(st[b], jet), pa[b], pos, vel, alpha) for b in range(B)]
        recs = r.next_records()
        for tick in range(6):
            r.run(1, log=False)
            x, fm, status, iters = r.mpc.solve(recs)
            after = r.state()
            for b in range(B):
                rec_m = rm.build_record(cfg, models[b], rm.measured(s_host[b], jet), pa[b])
                assert relerr(recs[b], rec_m) < 2e-6, (tick, b)
                models[b].consume(fm[b], status[b])
                s_m = rm.advance(cfg, s_host[b], pa[b], tick, fm[b], status[b], alpha, adt, jet=jet)
                assert relerr(after[b], s_m) < 2e-6, (tick, b)
                # the estimate follows the NN thrust, and the NN thrust moved away from the polynomial model's
                assert np.abs(after[b, layout.PS_EST:layout.PS_EST + 8:2] - after[b, layout.PS_TNN:layout.PS_TNN + 4]).max() < 5.0
            recs = r.next_records()
            s_host = after
        assert (status == layout.STATUS_SOLVED).all()
        r.set_jet_plant(None)                                  # back to the polynomial jet plant
        r.reset(st, pa)
        recs = r.next_records()
        r.run(1, log=False)
        _, fm, status, _ = r.mpc.solve(recs)
        after = r.state()
        for b in range(B):
            assert relerr(after[b], rm.advance(cfg, st[b], pa[b], 0, fm[b], status[b], alpha, adt)) < 1e-12
    finally:
        r.close()
        jm.close()


def test_jet_plant_option_closed_loop_properties(ro, layout):
    """256 hover loops, 2 s, with the LSTM jet plant + EKF in the loop (the controller's own jet model is the polynomial
    one, so this is a loop with model mismatch): every solve optimal, the EKF estimates stay on the NN thrust, the
    thrusts stay in the model's range, the altitude stays near the reference."""
    import os
    from conftest import ROOT
    gold = np.load(os.path.join(ROOT, "tests", "golden", "jet_lstm.npz"))
    jp = importlib.import_module(PKG + ".jet_plant")
    jm = jp.JetModelTotal(gold["w_ih"], gold["w_hh"], gold["b_ih"], gold["b_hh"], gold["fc_w"], gold["fc_b"], gold["norm"],
                          device=0, max_series=64)
    B, T = 256, 400
    cfg, st, pa, _, r = _make(ro, layout, B, "hover")
    try:
        r.set_jet_plant(jm)
        r.reset(st, pa)
        log = r.run(T)
        final = r.state()
    finally:
        r.close()
        jm.close()
    assert (log[:, :, 14] == 1).all()
    est = final[:, layout.PS_EST:layout.PS_EST + 8:2]
    tnn = final[:, layout.PS_TNN:layout.PS_TNN + 4]
    assert np.abs(est - tnn).max() < 2.0                       # N: the estimate follows the plant thrust
    assert (tnn > 50).all() and (tnn < 260).all()
    np.testing.assert_array_equal(final[:, layout.PS_T:layout.PS_T + 4], est)     # the applied thrust IS the estimate (set_thrust)
    z_err = np.abs(log[-50:, :, 2] - pa[None, :, layout.PP_PINIT + 2])
    assert np.median(z_err) < 0.1 and z_err.max() < 0.6
    assert np.isfinite(final).all()


def test_tick_state_machine_matches_reference_model(ro, layout):
    """50 ticks (two releases of the 20-tick hold at ticks 19 and 39, window pushes on the same ticks) of freshly
    configured loops on the REFERENCE's own trajectory files, with the yaw drifting through +pi (wrapped measurement
    jumps, turn counter) and a rolling attitude (so a column frozen at its push differs from one recomputed later):
    every record the device builds equals the record of the model written from the reference's plugins."""
    import os
    import tick_model as tm
    from conftest import ROOT
    cfg = layout.paper_config()
    pos, vel, alpha, adt = ro.load_reference_trajectories(os.path.join(ROOT, "tests", "golden", "reference_trajectories.npz"))
    B = 4
    st, pa = ro.make_plant(cfg, B, workload="hover", seed0=99)
    for b in range(B):
        pa[b, layout.PP_TICK0] = 0.0
        st[b, layout.PS_RPY + 2] = (np.pi - 0.012) if b % 2 == 0 else (-np.pi + 0.012)     # +pi and -pi crossings
        st[b, layout.PS_HANG + 2] = (1.2 if b % 2 == 0 else -1.2)                         # yaw rate ~ +-0.6 rad/s
        st[b, layout.PS_HANG] = 0.8                                                        # rolling
        pa[b, layout.PP_PINIT:layout.PP_PINIT + 3] = st[b, layout.PS_P:layout.PS_P + 3]   # configured here and now
        pa[b, layout.PP_RPYINIT:layout.PP_RPYINIT + 3] = tm.as_rpy(st[b, layout.PS_RPY:layout.PS_RPY + 3])
    r = ro.ClosedLoopRollout(cfg, B, pos, vel, alpha, adt, device=0)
    try:
        r.reset(st, pa)
        models = [tm.ReferenceTickModel(cfg, st[b], pa[b], pos, vel, alpha) for b in range(B)]
        s_host = st.copy()
        crossed = np.zeros(B, dtype=bool)
        for tick in range(50):
            recs = r.next_records()
            for b in range(B):
                rec_m = tm.record_from_tick(cfg, models[b].update(s_host[b]), rm.kin_record(cfg, s_host[b], pa[b]))
                assert relerr(recs[b], rec_m) < 1e-12, (tick, b, np.abs(recs[b] - rec_m).argmax())
                assert recs[b, layout.IN_HOLD] == (0.0 if tick % cfg.ratio == cfg.ratio - 1 else 1.0)
                crossed[b] |= abs(recs[b, layout.IN_X0 + 8]) > np.pi and abs(recs[b, layout.IN_RPY + 2]) <= np.pi
            x, fm, status, iters = r.mpc.solve(recs)
            for b in range(B):
                models[b].consume(fm[b], status[b])
            r.run(1, log=False)
            s_host = r.state()
            for b in range(B):                                 # latched commands = what the harness feeds back into QPInput
                np.testing.assert_array_equal(s_host[b, layout.PS_U:layout.PS_U + 4], models[b].qp.throttleMPC)
                np.testing.assert_allclose(s_host[b, layout.PS_Q:layout.PS_Q + 8], models[b].m_jointsPositionReference, rtol=0, atol=1e-15)
        assert crossed.all()                                   # every loop went through +-pi with its turn counter
        assert all(abs(m.init_state.m_nTurns[2]) == 1 for m in models)
    finally:
        r.close()


def test_attitude_tracks_and_native_rate_alpha(ro, layout):
    """Non-zero RPY / RPYDot tracks (vsmpc_rollout_set_attitude_tracks): the RPY reference is the configure-time RPY + the
    track and the angular-momentum reference m_inertia * m_W * RPYDot at the attitude of the push
    (costsVSMPC.cpp:110-112,141-146,266-286); and an alpha-gravity track already at 1 / periodMPC, which the reference
    does NOT resample and follows to its last sample (TrajectoryManager.cpp:121-126).  Records against the reference-derived
    model over 45 ticks (two pushes)."""
    import os
    import tick_model as tm
    from conftest import ROOT
    cfg = layout.paper_config()
    pos, vel, alpha, adt = ro.load_reference_trajectories(os.path.join(ROOT, "tests", "golden", "reference_trajectories.npz"))
    n = len(pos)
    t = np.arange(n)[:, None] / 10.0
    rpy = 0.05 * np.sin(0.3 * t) * np.array([[1.0, -0.5, 0.2]])
    rpyd = 0.015 * np.cos(0.3 * t) * np.array([[1.0, -0.5, 0.2]])
    alpha200 = np.linspace(0.3, 1.0, 31)                       # 31 samples at 200 Hz: exhausted after 30 ticks
    B = 3
    st, pa = ro.make_plant(cfg, B, workload="hover", seed0=7)
    for b in range(B):
        pa[b, layout.PP_TICK0] = 0.0
        st[b, layout.PS_HANG] = 0.6
        pa[b, layout.PP_PINIT:layout.PP_PINIT + 3] = st[b, layout.PS_P:layout.PS_P + 3]
        pa[b, layout.PP_RPYINIT:layout.PP_RPYINIT + 3] = tm.as_rpy(st[b, layout.PS_RPY:layout.PS_RPY + 3])
    r = ro.ClosedLoopRollout(cfg, B, pos, vel, alpha200, cfg.period_mpc, device=0)
    try:
        r.set_attitude_tracks(rpy, rpyd)
        r.reset(st, pa)
        models = [tm.ReferenceTickModel(cfg, st[b], pa[b], pos, vel, alpha200, fps_alpha=200, traj_rpy=rpy, traj_rpy_dot=rpyd)
                  for b in range(B)]
        s_host = st.copy()
        for tick in range(45):
            recs = r.next_records()
            for b in range(B):
                rec_m = tm.record_from_tick(cfg, models[b].update(s_host[b]), rm.kin_record(cfg, s_host[b], pa[b]))
                assert relerr(recs[b], rec_m) < 1e-12, (tick, b, int(np.abs(recs[b] - rec_m).argmax()))
            x, fm, status, iters = r.mpc.solve(recs)
            for b in range(B):
                models[b].consume(fm[b], status[b])
            r.run(1, log=False)
            s_host = r.state()
        assert np.abs(recs[:, layout.IN_XREF + 9:layout.IN_XREF + 12]).max() > 1e-3      # a non-zero h_ang reference got in
        assert (recs[:, layout.IN_ALPHA] == alpha200[-1]).all()                          # held at the LAST sample of the track
    finally:
        r.close()


def test_closed_loop_matches_oracle_in_the_loop(ro, layout, ref):
    """45 ticks (two releases of the 20-tick hold) of 2 hover instances: the resident GPU loop against the numpy
    model driven by the oracle's exact QP optimum, state by state."""
    cfg, st, pa, (pos, vel, alpha, adt), r = _make(ro, layout, 2, "hover")
    rcfg = ref.paper_config()
    try:
        r.run(45, log=False)
        gpu = r.state()
    finally:
        r.close()
    for b in range(2):
        s = st[b].copy()
        model = rm.make_tick_model(cfg, s, pa[b], pos, vel, alpha)
        for tick in range(45):
            rec = rm.build_record(cfg, model, s, pa[b])
            x, _, _, _ = ref.solve_instance(rcfg, rec)
            fm = ref.first_move_vector(rcfg, x)
            model.consume(fm, 1)
            s = rm.advance(cfg, s, pa[b], tick, fm, 1, alpha, adt)
        assert relerr(gpu[b], s) < 1e-8, b


def test_hover_rollout_properties(ro, layout):
    """64 instances x 800 ticks (4 s): every solve optimal, throttle moves only on the free tick of the hold, the
    flight stays near the reference and the attitude converges."""
    B, T = 64, 800
    cfg, st, pa, _, r = _make(ro, layout, B, "hover")
    try:
        log = r.run(T)
        final = r.state()
    finally:
        r.close()
    assert (log[:, :, 14] == 1).all()
    u = log[:, :, 10:14]
    tick0 = pa[:, layout.PP_TICK0].astype(int)
    changed = np.abs(np.diff(u, axis=0)).max(axis=2) > 1e-9                   # [T-1, B]: command differs from previous tick
    phase = (np.arange(1, T)[:, None] + tick0[None, :]) % cfg.ratio
    assert changed.any() and (phase[changed] == cfg.ratio - 1).all()
    p_err = np.abs(log[:, :, 0:3] - pa[None, :, layout.PP_PINIT:layout.PP_PINIT + 3])
    assert p_err.max() < 0.6
    # envelope from the linearised closed loop (monodromy matrix of the oracle-in-the-loop model over one hold period,
    # tests/closed_loop_linearisation.py), not from the run: with spectral radius rho per period an initial deviation can
    # grow at most like rho^(T / ratio) in the long run; the transient factor of the non-normal map is bounded by its
    # largest singular value over the same number of periods
    import closed_loop_linearisation as cl
    M, _, rho = cl.load_fixture()      # derived by tests/closed_loop_linearisation.py, re-checked by the CPU suite
    assert abs(cl.spectral_radius(M) - rho) < 1e-12
    assert rho < 1.008, rho                                  # slow lateral mode: e-folding time 16 s (named in
                                                             # tests/test_closed_loop_linearisation.py)
    periods = T // cfg.ratio
    gain = float(np.linalg.norm(np.linalg.matrix_power(M, periods)[6:9, :][:, 6:9], 2))   # attitude -> attitude over the run
    rpy_err0 = np.abs(st[:, layout.PS_RPY:layout.PS_RPY + 3] - pa[:, layout.PP_RPYINIT:layout.PP_RPYINIT + 3]).max(axis=1)
    rpy_errT = np.abs(log[-200:, :, 3:6] - pa[None, :, layout.PP_RPYINIT:layout.PP_RPYINIT + 3]).max(axis=(0, 2))
    assert np.median(rpy_errT) < max(gain, rho ** periods) * np.median(rpy_err0) * 1.5 + 0.02, (rho, gain)
    assert rpy_errT.max() < 0.15
    assert np.isfinite(final).all()
    assert (final[:, layout.PS_T:layout.PS_T + 4] > 50).all() and (final[:, layout.PS_T:layout.PS_T + 4] < 260).all()


def _altitude_error(cfg, layout, log, pa, pos):
    # column 0 of the reference's FIFO window at tick k: sample max(0, shifts so far - (columns - 1)) -- the reference
    # tracks its trajectory one window length (1 s) behind the sample it pushes (costsVSMPC.cpp:121-165)
    T = log.shape[0]
    tick = pa[:, layout.PP_TICK0].astype(int)[None, :] + np.arange(1, T + 1)[:, None]
    ns = 1 + (tick + 1) // cfg.ratio
    idx = np.clip(ns - (cfg.n_ref_cols - 1), 0, len(pos) - 1)
    return np.abs(log[:, :, 2] - (pa[None, :, layout.PP_PINIT + 2] + pos[idx, 2]))


def test_takeoff_rollout_tracks_the_moving_reference(ro, layout):
    """64 loops started at random ticks of the climb (19 s .. 35 s of the take-off timeline), 4 s each: the altitude
    follows the moving reference window, every solve is optimal."""
    B, T = 64, 800
    cfg, st, pa, (pos, vel, alpha, adt), r = _make(ro, layout, B, "takeoff")
    try:
        log = r.run(T)
    finally:
        r.close()
    assert np.isfinite(log).all() and (log[:, :, 14] == 1).all()
    ez = _altitude_error(cfg, layout, log, pa, pos)
    assert ez.max() < 0.3, ez.max()
    assert np.abs(log[:, :, 3:6]).max() < 0.15
    # where the tracked (lagged) reference rises during the run, the robot must actually climb with it
    tick0 = pa[:, layout.PP_TICK0].astype(int)
    col0 = lambda tk: np.clip(1 + (tk + 1) // cfg.ratio - (cfg.n_ref_cols - 1), 0, len(pos) - 1)
    rise = pos[col0(tick0 + T), 2] - pos[col0(tick0), 2]
    climbing = rise > 0.1
    assert climbing.sum() > 10
    assert (log[-1, climbing, 2] - st[climbing, layout.PS_P + 2] > 0.5 * rise[climbing]).all()


def test_montecarlo_rollout_recovers_from_disturbances(ro, layout):
    """configs[3]-style closed loops: 4x wider initial scatter (thrust imbalance up to ~20 %) and a 0.1 s push of up to
    50 N / 30 Nm.  The jets are slow (seconds), so the excursion peaks after ~4 s; after 12 s the flight is back near
    the reference.  Bounds are loose property checks on this synthetic plant, not reference numbers."""
    B, T = 64, 2400
    cfg, st, pa, (pos, vel, alpha, adt), r = _make(ro, layout, B, "montecarlo")
    try:
        log = r.run(T)
        final = r.state()
    finally:
        r.close()
    assert np.isfinite(log).all() and np.isfinite(final).all()
    assert (log[:, :, 14] == 1).all()
    assert log[:, :, 15].max() > 1                                  # some ticks needed real active-set iterations
    ez = _altitude_error(cfg, layout, log, pa, pos)
    assert ez.max() < 6.0 and np.abs(log[:, :, 3:6]).max() < 1.0   # bounded excursion
    assert np.median(ez[-1]) < 0.25 and ez[-1].max() < 1.2         # recovered
    assert np.median(ez[-1]) < 0.5 * np.median(ez[400])            # and much better than at the 2 s mark


def test_tree_plant_matches_model(ro, layout):
    """N2 inside N1 (vsmpc_rollout_set_tree): the plant's joint vectoring is the kinematics the MPC linearises -- every tick
    the provider / kinematics kernels are evaluated on the plant's own joints (src/mujoco_lib/ironcub_mujoco_simulator.py:
    318-346 -> utils/src/Robot.cpp:198-335 in the harness), and A_mom,body(q), I_B(q), Lambda_lin / Lambda_ang replace the
    plant parameters.  Records and plant states against the numpy model on oracle/robot_tree_ref.py + the oracle's
    kinematics terms over eight ticks (the joints move every tick), a graph-replayed chunk of 30 ticks against the same
    model, then the parametric plant again."""
    RT = importlib.import_module(PKG + ".robot_tree")
    tree = RT.default_tree()
    cfg = layout.paper_config()
    B = 6
    st, pa = ro.make_plant_tree(cfg, B, tree, workload="montecarlo")
    pos, vel, alpha, adt = ro.make_trajectory(cfg, "hover", 20.0)
    r = ro.ClosedLoopRollout(cfg, B, pos, vel, alpha, adt, device=0)
    try:
        r.reset(st, pa)
        r.set_tree(tree)
        with pytest.raises(Exception):
            r.run(1)                                           # the plant changed under the records: reset() first
        r.reset(st, pa)
        rm.set_tree(tree)
        s_host = st.copy()
        models = [rm.make_tick_model(cfg, st[b], pa[b], pos, vel, alpha) for b in range(B)]
        recs = r.next_records()
        moved = 0.0
        for tick in range(8):
            r.run(1, log=False)
            x, fm, status, iters = r.mpc.solve(recs)
            after = r.state()
            for b in range(B):
                rec_m = rm.build_record(cfg, models[b], s_host[b], pa[b])
                assert relerr(recs[b], rec_m) < 1e-12, (tick, b, int(np.abs(recs[b] - rec_m).argmax()))
                models[b].consume(fm[b], status[b])
                s_m = rm.advance(cfg, s_host[b], pa[b], tick, fm[b], status[b], alpha, adt)
                assert relerr(after[b], s_m) < 1e-11, (tick, b)
            moved = max(moved, float(np.abs(after[:, layout.PS_Q:layout.PS_Q + 8]).max()))
            recs = r.next_records()
            s_host = after
        assert (status == layout.STATUS_SOLVED).all() and moved > 1e-3        # the joints (and with them A_mom, Lambda) did move
        # the record's Lambda is the tree's, not the parameter record's: column j = dA_mom/dq_j T would differ
        lam = recs[0, layout.IN_LLIN:layout.IN_LLIN + 48]
        assert np.abs(lam).max() > 1.0
        # 30 more ticks replayed from the captured graph (25) + direct launches (5), against the model driven by the
        # device's own first moves
        for tick in range(8, 38):
            x, fm, status, iters = r.mpc.solve(recs)
            for b in range(B):
                rm.build_record(cfg, models[b], s_host[b], pa[b])
                models[b].consume(fm[b], status[b])
                s_host[b] = rm.advance(cfg, s_host[b], pa[b], tick, fm[b], status[b], alpha, adt)
            r.run(1, log=False)
            recs = r.next_records()
        r2 = ro.ClosedLoopRollout(cfg, B, pos, vel, alpha, adt, device=0)
        try:
            r2.set_tree(tree)
            r2.reset(st, pa)
            r2.run(38, log=False)                              # one call: graph replay
            assert relerr(r2.state(), r.state()) < 1e-9
            assert relerr(r.state(), s_host) < 1e-8
        finally:
            r2.close()
        r.set_tree(None)                                       # back to the parametric plant
        rm.set_tree(None)
        r.reset(st, pa)
        recs = r.next_records()
        r.run(1, log=False)
        _, fm, status, _ = r.mpc.solve(recs)
        for b in range(B):
            assert relerr(r.state()[b], rm.advance(cfg, st[b], pa[b], 0, fm[b], status[b], alpha, adt)) < 1e-12
    finally:
        rm.set_tree(None)
        r.close()


def test_tree_plant_hover_properties(ro, layout):
    """64 hover loops on the kinematic-tree plant, 12 s (2400 ticks, graph-replayed): every solve optimal, the flight
    settles onto the trimmed hover the oracle-in-the-loop model finds (joints move to ~0.4 rad to trim the tree's jet
    geometry), attitude and position stay bounded.  The bound on the slow lateral mode comes from the linearised closed loop
    on THIS plant (tests/golden/hover_monodromy_tree.npz: rho = 1.0022 per 0.1 s hold period -- the same lateral-momentum /
    arm-jet thrust-rate pair as on the parametric plant, where it is 1.0062): the plant whose Lambda IS its kinematics
    still has the marginal mode, so it belongs to the formulation (no weight on the thrust split, no roll -> lateral-force
    model inside the horizon), not to an inconsistency of the synthetic plant."""
    import closed_loop_linearisation as cl
    RT = importlib.import_module(PKG + ".robot_tree")
    tree = RT.default_tree()
    cfg = layout.paper_config()
    B, T = 64, 2400
    st, pa = ro.make_plant_tree(cfg, B, tree, workload="hover")
    pos, vel, alpha, adt = ro.make_trajectory(cfg, "hover", 30.0)
    r = ro.ClosedLoopRollout(cfg, B, pos, vel, alpha, adt, device=0)
    try:
        r.set_tree(tree)
        r.reset(st, pa)
        log = r.run(T)
        final = r.state()
    finally:
        r.close()
    assert np.isfinite(log).all() and (log[:, :, 14] == 1).all()
    p_err = np.abs(log[:, :, 0:3] - pa[None, :, layout.PP_PINIT:layout.PP_PINIT + 3])
    assert p_err.max() < 0.8 and np.median(p_err[-400:].max(axis=(0, 2))) < 0.25
    assert np.abs(log[:, :, 3:6] - pa[None, :, layout.PP_RPYINIT:layout.PP_RPYINIT + 3]).max() < 0.35
    assert (final[:, layout.PS_T:layout.PS_T + 4] > 50).all() and (final[:, layout.PS_T:layout.PS_T + 4] < 260).all()
    assert np.abs(final[:, layout.PS_Q:layout.PS_Q + 8]).max() < 1.2          # the trim, not a wind-up
    M, orbit, rho = cl.load_fixture(tree=True)
    assert abs(cl.spectral_radius(M) - rho) < 1e-12 and rho < 1.003
    # the loop ends near the model's orbit point (taken after 10 s of the oracle-in-the-loop model on seed 4321 = instance 0)
    assert np.abs(final[0, layout.PS_Q:layout.PS_Q + 8] - orbit[layout.PS_Q:layout.PS_Q + 8]).max() < 0.15
