"""Host-side logic: record layout, workload generator determinism, jet model mirror, wrapper semantics."""
import numpy as np


def test_jet_model_mirror_matches_oracle(pkg, ref):
    import importlib
    jm = importlib.import_module(pkg.__name__ + ".jet_model").JetModel()
    for T, Td, u in [(170.0, 0.0, 75.0), (20.0, -30.0, 5.0), (220.0, 55.0, 99.0)]:
        Tb, Tdb = jm.standardizeThrust_u2T(T), jm.standardizeThrustDot_u2T(Td)
        assert jm.compute_f(Tb, Tdb) == ref.jet_f(ref.std_thrust(T), ref.std_thrust_dot(Td))
        assert jm.compute_g(Tb, Tdb) == ref.jet_g(ref.std_thrust(T), ref.std_thrust_dot(Td))
        assert jm.compute_df_dT(Tb, Tdb) == ref.jet_df_dT(Tb, Tdb)
        assert jm.compute_dg_dTdot(Tb, Tdb) == ref.jet_dg_dTd(Tb, Tdb)
        assert jm.compute_v(jm.standardizeThrottle_u2T(u)) == ref.v_of_throttle(u)
        assert abs(jm.destandardizeThrust_u2T(Tb) - T) < 1e-12
    assert abs(float(jm.steady_state_throttle(170.0)) - 75.4498) < 1e-3


def test_workload_generator_is_deterministic_and_sliceable(synth, layout):
    cfg = layout.paper_config()
    a = synth.make_batch(cfg, 12, workload="takeoff")
    b = synth.make_batch(cfg, 12, workload="takeoff")
    np.testing.assert_array_equal(a, b)
    tail = synth.make_batch(cfg, 4, workload="takeoff", first_index=8)     # any rank can rebuild its slice
    np.testing.assert_array_equal(a[8:], tail)
    assert a.shape == (12, 294) and a.flags["C_CONTIGUOUS"]
    h = synth.make_batch(cfg, 41, workload="hover")
    hold = h[:, layout.IN_HOLD]
    assert hold[0] == 0.0 and hold[20] == 0.0 and hold[40] == 0.0 and hold[1:20].all()   # 1 free tick in 20
    m = h[:, layout.IN_MASS]
    np.testing.assert_array_equal(m, m.astype(np.float32).astype(np.float64))           # Robot.h:338 float mass
    R = h[:, layout.IN_WRB:layout.IN_WRB + 9].reshape(-1, 3, 3)
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.broadcast_to(np.eye(3), R.shape), atol=1e-12)
    x0 = h[:, 0:26]
    np.testing.assert_allclose(x0[:, 20:23], x0[:, 0:3] - h[:, layout.IN_PREF:layout.IN_PREF + 3], atol=1e-15)
    c5 = layout.horizon2x_config()
    assert synth.make_batch(c5, 3).shape == (3, 414)


def test_takeoff_profiles(synth):
    assert synth.alpha_gravity_profile(0.0) == 0.08 and synth.alpha_gravity_profile(25.0) == 1.0
    assert 0.08 < synth.alpha_gravity_profile(11.0) < 1.0
    p, v = synth.takeoff_profile(27.5)
    assert 0 < p[2] < 2.5 and v[2] > 0
    p, v = synth.takeoff_profile(40.0)
    assert p[2] == 2.5 and v[2] == 0


def test_config_derived_sizes(layout, ref):
    for pc, rc in ((layout.paper_config(), ref.paper_config()), (layout.horizon2x_config(), ref.horizon2x_config())):
        assert (pc.n_var, pc.n_con, pc.n_in, pc.ratio) == (rc.n_var, rc.n_con, rc.n_in, rc.ratio)
        assert pc.n_inputs == pc.n_var - 26 * (pc.n_iter + 1)
