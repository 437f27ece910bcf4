"""Known-answer tests that pin the oracle to the formulas cited from the reference.
The reference ships no tests or vectors for this path, so these closed-form values are the pin."""
import math

import numpy as np
import pytest


def test_sizes(ref):
    c = ref.paper_config()
    assert (c.n_var, c.n_con, c.n_in) == (588, 512, 294)          # variableSamplingMPC.cpp:42-45
    assert (c.off_joints, c.off_throttle) == (468, 564)
    c5 = ref.horizon2x_config()
    assert (c5.n_var, c5.n_con, c5.n_in) == (1146, 994, 414)
    assert c.ratio == 20 and c5.ratio == 40                        # constraintsVSMPC.cpp:322


def test_dt_schedule(ref):
    dts = ref.dt_schedule(ref.paper_config())                      # constraintsVSMPC.cpp:45-51,78-84
    exp = [0.005, 0.008095238095238095, 0.011190476190476190, 0.014285714285714285, 0.017380952380952380,
           0.020476190476190476, 0.023571428571428570]
    np.testing.assert_allclose(dts[:7], exp, rtol=1e-12)
    np.testing.assert_allclose(dts[7:], 0.1)
    assert abs(dts[:7].sum() - 0.1) < 1e-15 and abs(dts.sum() - 1.1) < 1e-14
    d5 = ref.dt_schedule(ref.horizon2x_config())
    assert abs(d5[0] - 0.0025) < 1e-15 and abs(d5[13] - 0.011785714285714287) < 1e-12
    assert abs(d5[:14].sum() - 0.1) < 1e-14 and abs(d5.sum() - 2.1) < 1e-13


def test_throttle_bounds(ref):
    vmin, vmax = ref.throttle_bounds(ref.paper_config())           # constraintsVSMPC.cpp:329-332
    assert abs(vmin - (-1.5211460323229673)) < 1e-15
    assert abs(vmax - 1.6509573743290515) < 1e-15


def test_jet_model_values(ref):
    Tb = ref.std_thrust(170.0)                                      # JetModel.cpp:29-79
    f, g = ref.jet_f(Tb, 0.0), ref.jet_g(Tb, 0.0)
    assert abs(f - (-6.61528895777356)) < 1e-13 and abs(g - 7.459446256269752) < 1e-13
    vstar = -f / g
    assert abs(vstar - 0.8868337850431367) < 1e-13
    c12 = ref.JET_COEFF[12]
    ubar = (-1 + math.sqrt(1 + 4 * c12 * vstar)) / (2 * c12)
    assert abs(ubar - 0.8930793370710481) < 1e-12
    u = float(ref.destd_throttle(vstar))
    assert abs(u - 75.4498) < 1e-3
    assert abs(ref.jetdyn_dh_dT(170.0, 0.0, u) - (-6.588675863522759)) < 1e-10
    assert abs(ref.jetdyn_dh_dTd(170.0, 0.0, u) - (-5.663926392867382)) < 1e-10
    assert abs(ref.jetdyn_G(170.0, 0.0) - 490.7793475387558) < 1e-10
    # clamp of the quadratic inverse (JetModel.cpp:101-107)
    assert float(ref.destd_throttle(-5.0)) == 0.0 and float(ref.destd_throttle(5.0)) == 100.0
    # v(u(0%)) / v(u(100%)) round trip
    assert abs(float(ref.destd_throttle(ref.v_of_throttle(33.0))) - 33.0) < 1e-10


def test_skew_and_w(ref):
    v = np.array([1.0, 2.0, 3.0]); w = np.array([-0.3, 0.7, 0.2])
    np.testing.assert_allclose(ref.from_vec_to_skew(v) @ w, np.cross(v, w))   # FlightControlUtils.cpp:77-85
    rpy = np.array([0.3, -0.2, 1.0])
    np.testing.assert_allclose(ref.w_inverse(rpy) @ ref.w_matrix(rpy), np.eye(3), atol=1e-14)


def test_assembly_structure(ref, synth, layout):
    cfg, pc = ref.paper_config(), layout.paper_config()
    rec = synth.make_batch(pc, 2, workload="hover")
    H, g, Ac, lo, hi = ref.assemble_dense(cfg, rec[1])              # hold tick
    d = np.diag(H)
    assert np.all(d[0:26] == 0)                                     # X0 unweighted (costsVSMPC.cpp:169)
    q = ref.state_weight(cfg)
    for k in range(1, 18):
        np.testing.assert_array_equal(d[26 * k:26 * k + 26], q)
    assert np.all(q[12:20] == 0)                                    # thrust states unweighted
    np.testing.assert_array_equal(d[468:564], 65020.0)              # 65000 + 20
    np.testing.assert_array_equal(d[564:568], 160000.0)             # v0: 80000 + anchor 80000 (A.4)
    np.testing.assert_array_equal(d[568:584], 160000.0)
    np.testing.assert_array_equal(d[584:588], 80000.0)
    assert H[564, 568] == -80000.0 and H[568, 564] == -80000.0
    assert np.linalg.matrix_rank(H) == 426                          # SURVEY.md 7 "H is only PSD"
    np.testing.assert_array_equal(H, H.T)
    # rows: dynamics | initial state | throttle (+20 padding rows 0 in [0,0])
    np.testing.assert_array_equal(lo[:468], hi[:468])
    np.testing.assert_array_equal(lo[442:468], rec[1][0:26])
    assert np.all(Ac[492:512] == 0) and np.all(lo[492:512] == 0) and np.all(hi[492:512] == 0)
    vprev = np.array([ref.v_of_throttle(u) for u in rec[1][141:145]])
    np.testing.assert_array_equal(lo[468:472], vprev)               # hold: l = u = v(u_prev)
    np.testing.assert_array_equal(hi[468:472], vprev)
    H0, g0, Ac0, lo0, hi0 = ref.assemble_dense(cfg, rec[0])         # free tick (index 0)
    vmin, vmax = ref.throttle_bounds(cfg)
    assert np.all(lo0[468:492] == vmin) and np.all(hi0[468:492] == vmax)
    # move blocking (constraintsVSMPC.cpp:89-128): stage 13 uses U_11 and v_5
    blk = Ac[26 * 13:26 * 14]
    assert np.any(blk[:, 468 + 8 * 11:468 + 8 * 12] != 0) and np.all(blk[:, 468:468 + 8 * 11] == 0)
    assert np.any(blk[:, 564 + 20:564 + 24] != 0) and np.all(blk[:, 564:564 + 20] == 0)
    # stage 3 uses v_0
    assert np.any(Ac[26 * 3:26 * 4, 564:568] != 0)
    # thrust map uses desired thrust for Bt but T0 for A (systemDynamicsVSMPC.cpp:410-415)
    A, Bj, Bt, c = ref.linearize(cfg, rec[0])
    assert abs(Bt[16, 0] - ref.jetdyn_G(rec[0][145], rec[0][149])) < 1e-12
    assert abs(A[16, 12] - ref.jetdyn_dh_dT(rec[0][133], rec[0][137], rec[0][141])) < 1e-12


def test_first_node_thrust_is_input_determined(ref, synth, layout):
    """X1[12:16] = T0 + dt0 * Tdot0 exactly (rows 12-15 of A carry no input)."""
    cfg, pc = ref.paper_config(), layout.paper_config()
    rec = synth.make_batch(pc, 1, workload="takeoff")[0]
    x, y, it, _ = ref.solve_instance(cfg, rec)
    np.testing.assert_allclose(x[26 + 12:26 + 16], rec[12:16] + 0.005 * rec[16:20], rtol=1e-12)


@pytest.mark.parametrize("workload", ["hover", "takeoff", "montecarlo"])
def test_exact_solve_certificate_and_bvls(ref, synth, layout, workload):
    """Two independent exact methods must agree and the KKT certificate of the reference-ordered QP holds."""
    from scipy.optimize import lsq_linear
    cfg, pc = ref.paper_config(), layout.paper_config()
    recs = synth.make_batch(pc, 6, workload=workload)
    for rec in recs:
        H, g, Ac, lo, hi = ref.assemble_dense(cfg, rec)
        x, y, it = ref.solve_exact(cfg, H, g, Ac, lo, hi)
        k = ref.kkt_certificate(H, g, Ac, lo, hi, x, y)
        assert k["stationarity_rel"] < 1e-12 and k["primal"] < 1e-10 and k["complementarity"] < 1e-6
        # BVLS on the Cholesky-whitened condensed problem
        nxs = 468
        sol = np.linalg.solve(Ac[:nxs, :nxs], np.column_stack([lo[:nxs], Ac[:nxs, nxs:]]))
        Z = np.vstack([-sol[:, 1:], np.eye(120)])
        xp = np.concatenate([sol[:, 0], np.zeros(120)])
        Hr = Z.T @ H @ Z
        gr = Z.T @ (H @ xp + g)
        Lc = np.linalg.cholesky(0.5 * (Hr + Hr.T))
        zl = np.full(120, -np.inf); zu = np.full(120, np.inf)
        zl[96:] = lo[468:492]; zu[96:] = hi[468:492]
        pinned = zl == zu
        zu[pinned] += 1e-9                       # lsq_linear needs lb < ub
        res = lsq_linear(Lc.T, -np.linalg.solve(Lc, gr), bounds=(zl, zu), method="bvls", tol=1e-14, max_iter=500)
        xb = xp + Z @ res.x
        assert np.abs(xb - x).max() / max(1.0, np.abs(x).max()) < 1e-6


def test_oracle_reproduces_golden(ref, golden_paper, golden_h2x):
    for gold, cfg in ((golden_paper, ref.paper_config()), (golden_h2x, ref.horizon2x_config())):
        for i, rec in enumerate(gold["inputs"]):
            A, Bj, Bt, c = ref.linearize(cfg, rec)
            np.testing.assert_allclose(A, gold["A"][i], rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(c, gold["c"][i], rtol=1e-13, atol=1e-13)
            x, y, it, _ = ref.solve_instance(cfg, rec)
            assert np.abs(x - gold["x"][i]).max() / max(1.0, np.abs(gold["x"][i]).max()) < 1e-10
            np.testing.assert_allclose(ref.first_move_vector(cfg, x), gold["first_move"][i], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("which", ["paper", "h2x"])
def test_algorithm_model_matches_oracle(ref, golden_paper, golden_h2x, which):
    """The kernel's algorithm (tests/algo_model.py) reaches the oracle's optimum, including the
    instances whose active set needs several block-pivoting rounds."""
    import algo_model
    gold, cfg = (golden_paper, ref.paper_config()) if which == "paper" else (golden_h2x, ref.horizon2x_config())
    for i, rec in enumerate(gold["inputs"]):
        xm, status, iters = algo_model.solve_model(cfg, ref, rec)
        assert status == 1
        assert np.abs(xm - gold["x"][i]).max() / max(1.0, np.abs(gold["x"][i]).max()) < 1e-10
        assert iters == int(gold["iters"][i])


@pytest.mark.parametrize("which", ["paper", "h2x"])
def test_joint_reduction_reaches_the_oracles_optimum(ref, golden_paper, golden_h2x, which):
    """Kernel v24's joint reduction (6 unknowns per joint block: Householder QR of (Lambda W^-1/2)^T, closed-form null
    component; algo_model.joint_reduction) in front of either condensing form: the oracle's optimum and the oracle's
    active-set iteration counts, also with a rank-deficient Lambda (no thrust: Lambda = 0; linear = angular rows: rank 3),
    and the condensed matrices equal the oracle's dense reference-ordered QP taken through the change of variables."""
    import algo_model
    import condense_model
    gold, cfg = (golden_paper, ref.paper_config()) if which == "paper" else (golden_h2x, ref.horizon2x_config())
    for i, rec in enumerate(gold["inputs"][:6]):
        for cond in (None, condense_model.condense_structured):
            out = {}
            xm, status, iters = algo_model.solve_model(cfg, ref, rec, reduce=True, condense=cond, out=out)
            assert status == 1 and iters == int(gold["iters"][i])
            assert np.abs(xm - gold["x"][i]).max() / max(1.0, np.abs(gold["x"][i]).max()) < 1e-10
        if i == 0:
            Me, ge, Le = algo_model.reduced_condensed(cfg, ref, rec)
            nz = Me.shape[0]
            assert nz == (104 if which == "paper" else 188)
            scale = np.abs(Me).max()
            assert np.abs(out["M"][:nz, :nz] - Me).max() < 1e-12 * scale and np.abs(out["M"][nz, :nz] - ge).max() < 1e-11 * np.abs(ge).max()
            assert np.abs(np.tril(out["L"][:nz, :nz]) - Le).max() < 1e-11 * np.abs(Le).max()
    rec = gold["inputs"][0].copy()
    for variant in ("zero", "rank3"):
        r = rec.copy()
        if variant == "zero":
            r[ref.IN_LLIN:ref.IN_LLIN + 48] = 0.0
        else:
            r[ref.IN_LANG:ref.IN_LANG + 24] = r[ref.IN_LLIN:ref.IN_LLIN + 24]
        xm, status, iters = algo_model.solve_model(cfg, ref, r, reduce=True)
        xo, _, ito, _ = ref.solve_instance(cfg, r)
        assert status == 1 and iters == ito and np.abs(xm - xo).max() / max(1.0, np.abs(xo).max()) < 1e-10


def test_dual_and_primal_box_qp_iterates_agree(ref, synth, layout):
    """The dual form of the box QP (P = S_NN^-1 = X^T X, mu = P_AA^-1 (v_u,A - b_A)) that the kernel runs for few
    active bounds walks through the same active sets, throttles and multipliers as the primal form on the Schur
    complement, iteration by iteration, and ends at the oracle's optimum."""
    import algo_model
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    recs = np.concatenate([synth.make_batch(cfg, 48, workload="takeoff", seed0=9000),
                           synth.make_batch(cfg, 48, workload="montecarlo", seed0=9500)])
    multi = 0
    for rec in recs:
        tp, td = [], []
        xp, sp_, itp = algo_model.solve_model(rcfg, ref, rec, trace=tp)
        xd, sd_, itd = algo_model.solve_model(rcfg, ref, rec, dual=True, trace=td)
        assert sp_ == sd_ == 1 and itp == itd
        for (s1, v1, g1), (s2, v2, g2) in zip(tp, td):
            np.testing.assert_array_equal(s1, s2)
            np.testing.assert_allclose(v1, v2, rtol=0, atol=1e-10)
            np.testing.assert_allclose(g1, g2, rtol=0, atol=1e-7 * (1.0 + np.abs(g1).max()))
        np.testing.assert_allclose(xp, xd, rtol=0, atol=1e-9)
        if itp > 1:
            multi += 1
            xo, _, ito, _ = ref.solve_instance(rcfg, rec)
            assert ito == itp and np.abs(xd - xo).max() / max(1.0, np.abs(xo).max()) < 1e-10
    assert multi >= 5


def test_kinematics_terms_closed_form(ref):
    """Single jet, identity attitude: Lambda_lin = -T S(a) J_rel, Lambda_ang = -T (S(a)(J_f - J_c) + S(r) S(a) J_rel),
    I_G = parallel-axis form of the base block (systemDynamicsVSMPC.cpp:128-130,159-226,321-350)."""
    kin = np.zeros(ref.KIN_SIZE)
    kin[ref.KIN_WRB:ref.KIN_WRB + 9] = np.eye(3).reshape(-1)
    kin[ref.KIN_THRUST] = 100.0
    kin[ref.KIN_AXES:ref.KIN_AXES + 3] = [0.0, 0.0, 1.0]
    kin[ref.KIN_ARMS:ref.KIN_ARMS + 3] = [0.0, 0.3, 0.0]
    col = ref.KIN_JOINT_OFFSET + 2
    kin[ref.KIN_JREL + 0 * 23 + col] = 0.2               # J_rel,0[0, col]  (x row)
    kin[ref.KIN_JFRAME + 1 * 23 + col] = 0.5             # J_frame,0[1, col]
    kin[ref.KIN_JCOM + 1 * 23 + col] = 0.1               # J_CoM[1, col]
    m, r = 70.0, np.array([0.0, 0.0, 0.2])
    Mb = np.zeros((6, 6)); Mb[:3, :3] = m * np.eye(3); Mb[3:, 3:] = np.diag([8.0, 7.0, 2.0])
    kin[ref.KIN_MB:ref.KIN_MB + 36] = Mb.reshape(-1)
    kin[ref.KIN_R:ref.KIN_R + 3] = r
    Llin, Lang, IG = ref.kinematics_terms(kin)
    # S(e_z) [0.2,0,0]^T = [0, 0.2, 0]
    np.testing.assert_allclose(Llin[:, 2], [0.0, -100.0 * 0.2, 0.0], atol=1e-13)
    assert np.count_nonzero(Llin) == 1
    # S(e_z)[0,0.4,0]^T = [-0.4,0,0]; S(r_arm) S(a) Jrel = [0,0.3,0] x [0,0.2,0] = 0
    np.testing.assert_allclose(Lang[:, 2], [100.0 * 0.4, 0.0, 0.0], atol=1e-13)
    np.testing.assert_allclose(IG, np.diag([8.0, 7.0, 2.0]) + m * (r @ r * np.eye(3) - np.outer(r, r)), atol=1e-12)
