"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle on the same seeded
inputs, against the committed golden fixtures, and — at BASELINE.json's full batch sizes — through
size-independent properties of the QP (dynamics feasibility, bounds, KKT stationarity, determinism,
batch-permutation invariance).

Tolerance: north_star allows 1e-4 relative solution error; the build's own bar, used here, is
  max_i |x_gpu - x_oracle|_inf / max(1, |x_oracle|_inf) <= 1e-8        (FP64 end to end).
"""
import ctypes

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.fixture(scope="module")
def mpc(solver_mod, layout):
    m = solver_mod.BatchedVSMPC(layout.paper_config(), device=0, max_batch=4096)
    yield m
    m.close()


def test_native_library_is_loaded(mpc):
    maps = open("/proc/self/maps").read()
    assert "libvsmpc.so" in maps
    assert "solve_kernel" in mpc.kernel_name


def test_linearize_matches_oracle(mpc, ref, synth, layout):
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    recs = np.concatenate([synth.make_batch(cfg, 24, workload=w) for w in ("hover", "takeoff", "montecarlo")])
    A, Bj, Bt, c, dt = mpc.linearize(recs)
    np.testing.assert_allclose(dt, ref.dt_schedule(rcfg), rtol=0, atol=1e-17)
    for b, rec in enumerate(recs):
        Ar, Bjr, Btr, cr = ref.linearize(rcfg, rec)
        assert relerr(A[b], Ar) < 1e-13 and relerr(Bj[b], Bjr) < 1e-14
        assert relerr(Bt[b], Btr) < 1e-13 and relerr(c[b], cr) < 1e-13
        # structural zeros are exact zeros (what sparseView() would drop, IMPCProblem.cpp:211)
        assert np.array_equal(A[b] == 0, Ar == 0) and np.array_equal(Bt[b] == 0, Btr == 0)


def test_dense_assembly_matches_oracle(mpc, ref, synth, layout):
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    recs = synth.make_batch(cfg, 3, workload="takeoff")
    for rec in recs:
        H, g, Ac, lo, hi = mpc.assemble_dense(rec)
        Hr, gr, Acr, lor, hir = ref.assemble_dense(rcfg, rec)
        np.testing.assert_array_equal(H, Hr)                  # constant Hessian, exact
        assert relerr(g, gr) < 1e-14 and relerr(Ac, Acr) < 1e-13
        assert relerr(lo, lor) < 1e-13 and relerr(hi, hir) < 1e-13
        assert np.array_equal(Ac == 0, Acr == 0)


def test_condensed_hessian_and_factor(mpc, ref, synth, layout):
    """Block-by-block check of the device condensing (P1/P2) and Cholesky (P3) against the oracle's dense, reference-
    ordered QP taken through the joint reduction (tests/algo_model.py: reduced_condensed): 6 unknowns per joint block,
    8 dummy unknowns up to the tile boundary, condensed dimension 104 (+ the gradient row) instead of 120."""
    import algo_model
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    rec = synth.make_batch(cfg, 2, workload="takeoff")[1]
    assert mpc.n_p == 112
    M, Lf = mpc.debug_condensed(rec)
    Me, ge, Le = algo_model.reduced_condensed(rcfg, ref, rec)
    nz = Me.shape[0]
    assert nz == 104
    Mh = np.tril(M[:nz, :nz])
    Mh = Mh + np.tril(Mh, -1).T
    assert relerr(Mh, Me) < 1e-12 and relerr(M[nz, :nz], ge) < 1e-12
    assert relerr(np.tril(Lf[:nz, :nz]), Le) < 1e-11
    assert relerr(Lf[nz, :nz], np.linalg.solve(Le, ge)) < 1e-11


def test_box_qp_both_formulations_match_oracle(mpc, ref, synth, layout):
    """Instances whose throttles saturate, picked from 768 take-off / Monte-Carlo states so that every size of the
    first violated set from 1 up to 12+ occurs: sizes <= 4 run the dual form (P = X^T X, register solver), larger
    ones the primal form on the Schur complement.  Solution, multipliers' consequence (bounds hit exactly) and the
    active-set iteration count are compared with the oracle for each."""
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    recs = np.concatenate([synth.make_batch(cfg, 384, workload="takeoff", seed0=9000),
                           synth.make_batch(cfg, 384, workload="montecarlo", seed0=9500)])
    x, fm, status, iters = mpc.solve(recs)
    assert (status == layout.STATUS_SOLVED).all()
    vmin, vmax = ref.throttle_bounds(rcfg)
    v = x[:, 564:588]
    nact = ((v == vmin) | (v == vmax)).sum(axis=1)          # bound variables sit exactly on their bound
    qp = np.where(iters > 1)[0]
    assert len(qp) >= 40
    chosen = []
    for k in sorted(set(nact[qp])):                          # up to three instances per active-set size
        chosen += list(qp[nact[qp] == k][:3])
    sizes = sorted(set(nact[chosen]))
    assert min(sizes) <= 1 and max(sizes) >= 12 and len([k for k in sizes if 2 <= k <= 6]) >= 3, sizes
    for b in chosen:
        xr, _, itr, _ = ref.solve_instance(rcfg, recs[b])
        assert relerr(x[b], xr) < TOL, (b, nact[b], relerr(x[b], xr))
        assert iters[b] == itr, (b, nact[b], iters[b], itr)
        assert v[b].min() >= vmin and v[b].max() <= vmax


@pytest.mark.parametrize("workload", ["hover", "takeoff", "montecarlo"])
def test_solve_matches_oracle(mpc, ref, synth, layout, workload):
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    recs = synth.make_batch(cfg, 64, workload=workload)
    x, fm, status, iters = mpc.solve(recs)
    assert (status == layout.STATUS_SOLVED).all()
    worst = {"x": 0.0, "thrust": 0.0, "thrust_dot": 0.0, "v": 0.0, "dq": 0.0, "traj": 0.0, "fm": 0.0}
    multi = 0
    for b, rec in enumerate(recs):
        xr, yr, itr, _ = ref.solve_instance(rcfg, rec)
        X, Xr = x[b, :468].reshape(18, 26), xr[:468].reshape(18, 26)
        worst["x"] = max(worst["x"], relerr(x[b], xr))
        worst["thrust"] = max(worst["thrust"], relerr(X[:, 12:16], Xr[:, 12:16]))
        worst["thrust_dot"] = max(worst["thrust_dot"], relerr(X[:, 16:20], Xr[:, 16:20]))
        worst["traj"] = max(worst["traj"], relerr(X[:, 0:12], Xr[:, 0:12]))
        worst["v"] = max(worst["v"], relerr(x[b, 564:], xr[564:]))
        worst["dq"] = max(worst["dq"], relerr(x[b, 468:564], xr[468:564]))
        worst["fm"] = max(worst["fm"], relerr(fm[b], ref.first_move_vector(rcfg, xr)))
        assert iters[b] == itr            # same active-set path as the oracle's pivoting rule
        multi += itr > 1
    for k, v in worst.items():
        assert v < TOL, (k, v)
    if workload == "takeoff":
        assert multi > 0                  # the batch really exercises the active-set loop


def test_golden_fixtures(mpc, golden_paper, layout):
    x, fm, status, iters = mpc.solve(golden_paper["inputs"])
    assert (status == layout.STATUS_SOLVED).all()
    for b in range(len(x)):
        assert relerr(x[b], golden_paper["x"][b]) < TOL
        assert relerr(fm[b], golden_paper["first_move"][b]) < TOL
    np.testing.assert_array_equal(iters, golden_paper["iters"])


def _kkt_properties(ref, rcfg, recs, x, sample):
    """Size-independent optimality properties checked on the reference-ordered dense QP."""
    vmin, vmax = ref.throttle_bounds(rcfg)
    for b in sample:
        H, g, Ac, lo, hi = ref.assemble_dense(rcfg, recs[b])
        r = Ac @ x[b]
        scale = max(1.0, np.abs(x[b]).max())
        assert np.maximum(lo - r, r - hi).max() < 1e-9 * scale            # primal feasibility of all 512 rows
        # reduced-gradient optimality: project the gradient onto the null space of the equality rows
        nxs = 468
        G = np.linalg.solve(Ac[:nxs, :nxs], Ac[:nxs, nxs:])
        Z = np.vstack([-G, np.eye(120)])
        rg = Z.T @ (H @ x[b] + g)
        gscale = max(1.0, np.abs(g).max())
        assert np.abs(rg[:96]).max() < 1e-9 * gscale                       # joints are unconstrained
        v = x[b, 564:588]
        lo_v, hi_v = lo[468:492], hi[468:492]
        at_lo, at_hi = np.abs(v - lo_v) < 1e-12, np.abs(v - hi_v) < 1e-12
        free = ~(at_lo | at_hi)
        assert np.abs(rg[96:][free]).max(initial=0.0) < 1e-9 * gscale
        pinned = lo_v == hi_v
        assert (rg[96:][at_lo & ~pinned] > -1e-9 * gscale).all()           # multiplier signs
        assert (rg[96:][at_hi & ~pinned] < 1e-9 * gscale).all()


@pytest.mark.parametrize("batch,workload", [(256, "hover"), (4096, "takeoff")])
def test_full_batch_properties(mpc, ref, synth, layout, batch, workload):
    """BASELINE.json configs[1] and configs[2] at full size: properties instead of per-instance oracle solves."""
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    base = synth.make_batch(cfg, min(batch, 512), workload=workload)
    recs = np.tile(base, (batch // len(base), 1)) if batch > len(base) else base
    x, fm, status, iters = mpc.solve(recs)
    assert (status == layout.STATUS_SOLVED).all()
    # determinism and batch-position independence: tiled copies are bit-identical
    if batch > len(base):
        for k in range(1, batch // len(base)):
            np.testing.assert_array_equal(x[:len(base)], x[k * len(base):(k + 1) * len(base)])
    x2, fm2, st2, it2 = mpc.solve(recs)
    np.testing.assert_array_equal(x, x2)
    perm = np.random.default_rng(0).permutation(batch)
    xp, _, _, _ = mpc.solve(recs[perm])
    np.testing.assert_array_equal(xp, x[perm])
    # checksum of checksums across the two runs
    assert float(np.sum(np.sum(x, axis=1))) == float(np.sum(np.sum(x2, axis=1)))
    # initial state pinned exactly, first-move block consistent with the primal
    np.testing.assert_array_equal(x[:, 0:26], recs[:, 0:26])
    np.testing.assert_array_equal(fm[:, 0:8], x[:, 468:476])
    np.testing.assert_array_equal(fm[:, 8:12], x[:, 564:568])
    np.testing.assert_array_equal(fm[:, 16:24], x[:, 26 + 12:26 + 20])
    np.testing.assert_allclose(fm[:, 12:16], ref.destd_throttle(x[:, 564:568]), rtol=1e-9, atol=1e-9)  # -1+sqrt(1+4cv) cancels ~2 digits
    vmin, vmax = ref.throttle_bounds(rcfg)
    assert x[:, 564:588].min() >= vmin and x[:, 564:588].max() <= vmax
    hold = recs[:, layout.IN_HOLD] != 0
    vprev = ref.v_of_throttle(recs[:, layout.IN_UPREV:layout.IN_UPREV + 4])
    np.testing.assert_allclose(x[hold, 564:568], vprev[hold], rtol=0, atol=1e-15)   # 20-tick hold
    sample = np.random.default_rng(1).choice(len(base), size=12, replace=False)
    _kkt_properties(ref, rcfg, recs, x, sample)


def test_device_entry_matches_host_entry(mpc, synth, layout):
    import torch
    cfg = layout.paper_config()
    recs = synth.make_batch(cfg, 40, workload="takeoff")
    x, fm, st, it = mpc.solve(recs)
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(recs).to(dev)
    d_x = torch.zeros((40, cfg.n_var), dtype=torch.float64, device=dev)
    d_fm = torch.zeros((40, 24), dtype=torch.float64, device=dev)
    d_st = torch.zeros(40, dtype=torch.int32, device=dev)
    d_it = torch.zeros(40, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        mpc.solve_device(d_in, d_x, d_fm, d_st, d_it)
    side.synchronize()
    np.testing.assert_array_equal(d_x.cpu().numpy(), x)
    np.testing.assert_array_equal(d_fm.cpu().numpy(), fm)
    np.testing.assert_array_equal(d_st.cpu().numpy(), st)
    np.testing.assert_array_equal(d_it.cpu().numpy(), it)
    # optional outputs may be NULL
    mpc.solve_device(d_in, None, None, d_st, None)
    torch.cuda.synchronize()
    assert (d_st.cpu().numpy() == layout.STATUS_SOLVED).all()


def test_pinned_buffers_take_the_direct_store_path(mpc, solver_mod, synth, layout):
    """Callers that hand pinned buffers (vsmpc_alloc_host) get the results written by the kernel straight into them
    (no device-to-host copies); same bits as through pageable buffers, also with optional outputs left out."""
    import importlib
    _lib = importlib.import_module(solver_mod.__name__.rsplit(".", 1)[0] + "._lib")
    lib = _lib.load()
    cfg = layout.paper_config()
    B = 200                                     # > 8 (mapped staging), not a multiple of anything
    big = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=B)
    recs = synth.make_batch(cfg, B, workload="takeoff")
    ref_x, ref_fm, ref_st, ref_it = big.solve(recs)

    def pinned(shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = lib.vsmpc_alloc_host(n)
        assert ptr
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), (n,)).view(dtype).reshape(shape), ptr

    bufs = [pinned((B, cfg.n_in), np.float64), pinned((B, cfg.n_var), np.float64), pinned((B, 24), np.float64),
            pinned((B,), np.int32), pinned((B,), np.int32)]
    try:
        (inp, _), (x, _), (fm, _), (st, _), (it, _) = bufs
        inp[:] = recs
        x[:] = np.nan; fm[:] = np.nan; st[:] = -7; it[:] = -7
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(lib.vsmpc_solve_batch(big._h, vp(inp), B, vp(x), vp(fm), vp(st), vp(it), None), "vsmpc_solve_batch")
        np.testing.assert_array_equal(x, ref_x)
        np.testing.assert_array_equal(fm, ref_fm)
        np.testing.assert_array_equal(st, ref_st)
        np.testing.assert_array_equal(it, ref_it)
        st[:] = -7; fm[:] = np.nan
        _lib.check(lib.vsmpc_solve_batch(big._h, vp(inp), B, None, vp(fm), vp(st), None, None), "vsmpc_solve_batch")
        np.testing.assert_array_equal(fm, ref_fm)
        np.testing.assert_array_equal(st, ref_st)
    finally:
        for _, ptr in bufs:
            lib.vsmpc_free_host(ptr)
        big.close()


def test_structured_and_syrk_condensing_agree(mpc, solver_mod, synth, layout):
    """The two condensing forms of the solve kernel (include/vsmpc.h, vsmpc_set_kernel_form) are independent algorithms
    for the same matrix -- forward / adjoint recursions on generator columns vs the sensitivity recursion + SYRK on the
    matrix cores: condensed Hessian, factor, outputs, statuses and iteration counts must agree to rounding."""
    cfg = layout.paper_config()
    recs = np.concatenate([synth.make_batch(cfg, 48, workload="takeoff"), synth.make_batch(cfg, 48, workload="montecarlo")])
    prev = mpc.set_kernel_form(solver_mod.KERNEL_FORM_STRUCTURED)
    try:
        a = mpc.solve(recs)
        Ma, La = mpc.debug_condensed(recs[5])[:2]
        assert mpc.set_kernel_form(solver_mod.KERNEL_FORM_SYRK) == solver_mod.KERNEL_FORM_STRUCTURED
        b = mpc.solve(recs)
        Mb, Lb = mpc.debug_condensed(recs[5])[:2]
    finally:
        mpc.set_kernel_form(prev)
    assert (a[2] == layout.STATUS_SOLVED).all()
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[3], b[3])
    assert relerr(a[0], b[0]) < 1e-11 and relerr(a[1], b[1]) < 1e-11
    # (entry [NZ, NZ], the constant term of the cost, is not formed by the structured form: nothing reads it)
    nz = 104   # 6 x 12 reduced joint unknowns + 8 dummies + 24 throttles
    assert relerr(np.tril(Ma[:nz + 1, :nz]), np.tril(Mb[:nz + 1, :nz])) < 1e-13
    assert relerr(np.tril(La[:nz + 1, :nz]), np.tril(Lb[:nz + 1, :nz])) < 1e-12
    with pytest.raises(ValueError):
        mpc.set_kernel_form(7)


def test_solves_are_deterministic_and_independent_of_batch_position(mpc, synth, layout):
    """Every 16-lane row of a panel wavefront factors its own copy of the diagonal tile, panel 0 is shared by several wavefronts
    that each do the same, and the tiles are dealt to the wavefronts at compile time (cholesky_wave, tools/gen_panel_asm.py):
    nothing depends on timing or on where in a launch an instance sits.  Bit-identical results for the same records solved
    twice, in another order, and inside a launch with two workgroups per CU."""
    cfg = layout.paper_config()
    recs = np.concatenate([synth.make_batch(cfg, 40, workload="takeoff"), synth.make_batch(cfg, 40, workload="montecarlo")])
    a = mpc.solve(recs)
    assert (a[2] == layout.STATUS_SOLVED).all()
    b = mpc.solve(recs)
    perm = np.random.default_rng(5).permutation(len(recs))
    c = mpc.solve(recs[perm])
    big = np.concatenate([recs[perm]] * 8)          # 640 instances: more than one workgroup per CU
    d = mpc.solve(big)
    for u, v, w, z in zip(a, b, c, d):
        np.testing.assert_array_equal(u, v)
        np.testing.assert_array_equal(u[perm], w)
        for rep in range(8):
            np.testing.assert_array_equal(u[perm], z[rep * len(recs):(rep + 1) * len(recs)])


def test_edge_cases(mpc, solver_mod, synth, layout, ref):
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    # empty batch
    x, fm, st, it = mpc.solve(np.empty((0, cfg.n_in)))
    assert x.shape == (0, 588) and st.shape == (0,)
    # batch of one, ragged (non multiple of anything) batch
    for n in (1, 3, 37):
        recs = synth.make_batch(cfg, n, workload="hover", first_index=100)
        x, fm, st, it = mpc.solve(recs)
        xr, _, _, _ = ref.solve_instance(rcfg, recs[-1])
        assert (st == 1).all() and relerr(x[-1], xr) < TOL
    # batch larger than the handle
    small = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=2)
    with pytest.raises(Exception) as e:
        small.solve(synth.make_batch(cfg, 3))
    assert "max_batch" in str(e.value)
    small.close()
    with pytest.raises(ValueError):
        mpc.solve(np.zeros((2, 100)))


def test_saturated_throttles(mpc, ref, synth, layout):
    """Force many throttle bounds active on a free (non-hold) tick: CoM reference 30 m away."""
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    recs = synth.make_batch(cfg, 8, workload="hover", first_index=40)
    recs[:, layout.IN_HOLD] = 0.0
    recs[:, layout.IN_XREF + 2::12] += 30.0
    recs[4:, layout.IN_XREF + 2::12] -= 60.0
    recs[:, 22] = recs[:, 2] - recs[:, layout.IN_XREF + 2]      # keep X0's position error consistent
    x, fm, st, it = mpc.solve(recs)
    assert (st == 1).all()
    vmin, vmax = ref.throttle_bounds(rcfg)
    nact = 0
    for b, rec in enumerate(recs):
        xr, yr, itr, _ = ref.solve_instance(rcfg, rec)
        assert relerr(x[b], xr) < TOL and it[b] == itr
        nact += int((np.abs(xr[564:] - vmax) < 1e-12).sum() + (np.abs(xr[564:] - vmin) < 1e-12).sum())
    assert nact >= 24 and it.max() >= 4                 # the case really saturates and iterates


def test_status_on_non_finite_input(mpc, synth, layout):
    cfg = layout.paper_config()
    recs = synth.make_batch(cfg, 4, workload="hover")
    recs[2, layout.IN_INERTIA] = np.nan
    x, fm, st, it = mpc.solve(recs)
    assert st[2] == layout.STATUS_NUMERICAL
    assert (st[[0, 1, 3]] == layout.STATUS_SOLVED).all()        # neighbours unaffected


def test_wrapper_consumes_only_when_solved(solver_mod, synth, layout):
    """variableSamplingMPC.cpp:91,104-108 semantics of the reference-shaped wrapper."""
    cfg = layout.paper_config()
    rec = synth.make_batch(cfg, 1, workload="hover")[0]
    w = solver_mod.VariableSamplingMPC()
    assert w.configure(cfg, initial_joint_positions=np.zeros(23))
    assert w.update(rec) and w.solveMPC()
    assert w.getQPProblemStatus() == layout.STATUS_SOLVED
    q1 = w.getJointsReferencePosition()
    assert np.abs(q1[3:11]).max() > 0 and np.all(q1[:3] == 0) and np.all(q1[11:] == 0)
    thr = w.getThrottleReference()
    assert thr.shape == (4,) and (thr >= 0).all() and (thr <= 100).all()
    assert w.getThrustReference().shape == (4,) and w.getThrustDotReference().shape == (4,)
    bad = rec.copy()
    bad[layout.IN_INERTIA] = np.nan
    assert w.update(bad) and w.solveMPC()                         # returns true like the reference
    assert w.getQPProblemStatus() != layout.STATUS_SOLVED
    np.testing.assert_array_equal(w.getJointsReferencePosition(), q1)   # previous commands persist
    np.testing.assert_array_equal(w.getThrottleReference(), thr)
    assert not w.update(rec[:10])


def test_kinematics_terms_match_oracle(mpc, ref, synth, layout):
    """Rows a3/a4: Lambda_lin,B, Lambda_ang,B (unfiltered) and I_G from raw Robot quantities."""
    rng = np.random.default_rng(7)
    B = 37                                             # odd count: record stride 697 doubles alternates 16 B alignment
    kin = rng.normal(size=(B, layout.KIN_SIZE))
    for b in range(B):                                 # a proper rotation, positive thrusts, SPD base mass matrix
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        kin[b, layout.KIN_WRB:layout.KIN_WRB + 9] = (q * np.sign(np.linalg.det(q))).reshape(-1)
        kin[b, layout.KIN_THRUST:layout.KIN_THRUST + 4] = rng.uniform(20, 220, size=4)
        a = rng.normal(size=(6, 6))
        kin[b, layout.KIN_MB:layout.KIN_MB + 36] = (a @ a.T + 6 * np.eye(6)).reshape(-1)
    recs = synth.make_batch(layout.paper_config(), B, workload="hover")
    before = recs.copy()
    Llin, Lang, IG = mpc.kinematics(kin, recs)
    for b in range(B):
        l1, l2, ig = ref.kinematics_terms(kin[b])
        assert relerr(Llin[b], l1) < 1e-13 and relerr(Lang[b], l2) < 1e-13 and relerr(IG[b], ig) < 1e-13
        np.testing.assert_array_equal(recs[b, layout.IN_LLIN:layout.IN_LLIN + 24], Llin[b].reshape(-1))
        np.testing.assert_array_equal(recs[b, layout.IN_LANG:layout.IN_LANG + 24], Lang[b].reshape(-1))
        np.testing.assert_array_equal(recs[b, layout.IN_INERTIA:layout.IN_INERTIA + 9], IG[b].reshape(-1))
    untouched = np.ones(recs.shape[1], dtype=bool)
    untouched[layout.IN_LLIN:layout.IN_LLIN + 48] = False
    untouched[layout.IN_INERTIA:layout.IN_INERTIA + 9] = False
    np.testing.assert_array_equal(recs[:, untouched], before[:, untouched])
    # the patched records still solve (I_G symmetric positive definite here)
    x, fm, st, it = mpc.solve(recs)
    assert (st == layout.STATUS_SOLVED).all()


@pytest.mark.parametrize("B", [1, 5, 37])
def test_tick_is_kinematics_then_solve_in_one_submission(mpc, synth, layout, B):
    """vsmpc_tick (the drop-in tick: update()'s kinematics terms + solveMPC() in one submission) == vsmpc_kinematics_batch
    followed by vsmpc_solve_batch, bit for bit -- through the mapped staging buffer (B <= 8) and through device buffers."""
    rng = np.random.default_rng(100 + B)
    kin = rng.normal(size=(B, layout.KIN_SIZE))
    for b in range(B):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        kin[b, layout.KIN_WRB:layout.KIN_WRB + 9] = (q * np.sign(np.linalg.det(q))).reshape(-1)
        kin[b, layout.KIN_THRUST:layout.KIN_THRUST + 4] = rng.uniform(20, 220, size=4)
        a = rng.normal(size=(6, 6))
        kin[b, layout.KIN_MB:layout.KIN_MB + 36] = (a @ a.T + 6 * np.eye(6)).reshape(-1)
    recs = synth.make_batch(layout.paper_config(), B, workload="takeoff")
    two = recs.copy()
    mpc.kinematics(kin, two)
    x2, fm2, st2, it2 = mpc.solve(two)
    one = recs.copy()
    one[:, layout.IN_LLIN:layout.IN_LLIN + 48] = np.nan          # whatever the caller left there is overwritten
    one[:, layout.IN_INERTIA:layout.IN_INERTIA + 9] = np.nan
    x1, fm1, st1, it1 = mpc.tick(kin, one)
    np.testing.assert_array_equal(one, two)                       # the completed record comes back
    np.testing.assert_array_equal(x1, x2)
    np.testing.assert_array_equal(fm1, fm2)
    np.testing.assert_array_equal(st1, st2)
    np.testing.assert_array_equal(it1, it2)
    assert (st1 == layout.STATUS_SOLVED).all()


def test_kinematics_options_match_oracle(solver_mod, ref, layout):
    """vsmpc_set_kinematics_options: Lambda_ang columns selected by (name-derived) robot joint index
    (systemDynamicsVSMPC.cpp:57-66,202-205) and jointsLambdaOption 'constant' (:186-200,329-337)."""
    rng = np.random.default_rng(11)
    kin = rng.normal(size=(9, layout.KIN_SIZE))
    for b in range(len(kin)):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        kin[b, layout.KIN_WRB:layout.KIN_WRB + 9] = (q * np.sign(np.linalg.det(q))).reshape(-1)
    m = solver_mod.BatchedVSMPC(layout.paper_config(), device=0, max_batch=16)
    try:
        for sel, const in (([5, 3, 12, 6, 7, 20, 9, 0], False), (list(range(3, 11)), True), ([22, 1, 2, 4, 8, 16, 10, 11], True)):
            m.set_kinematics_options(sel, const)
            Llin, Lang, IG = m.kinematics(kin)
            for b in range(len(kin)):
                l1, l2, ig = ref.kinematics_terms(kin[b], selector=sel, constant=const)
                assert relerr(Llin[b], l1) < 1e-13 and relerr(Lang[b], l2) < 1e-13 and relerr(IG[b], ig) < 1e-13
        with pytest.raises(Exception):
            m.set_kinematics_options([0, 1, 2, 3, 4, 5, 6, 23], False)      # not a robot joint
    finally:
        m.close()


@pytest.fixture(scope="module")
def mpc2x(solver_mod, layout):
    m = solver_mod.BatchedVSMPC(layout.horizon2x_config(), device=0, max_batch=256)
    yield m
    m.close()


def test_horizon2x_structured_and_syrk_condensing_agree(mpc2x, solver_mod, synth, layout):
    """The long-horizon structured form (72 generator columns per half: eight of them ride with the throttle wavefront;
    thrust contraction by LDS atomics inside the chains; sparse rows) against the SYRK form, as at the paper horizon."""
    cfg = layout.horizon2x_config()
    recs = np.concatenate([synth.make_batch(cfg, 24, workload="takeoff"), synth.make_batch(cfg, 24, workload="montecarlo"),
                           synth.make_batch(cfg, 16, workload="hover")])
    prev = mpc2x.set_kernel_form(solver_mod.KERNEL_FORM_STRUCTURED)
    try:
        a = mpc2x.solve(recs)
        Ma, La = mpc2x.debug_condensed(recs[5])[:2]
        mpc2x.set_kernel_form(solver_mod.KERNEL_FORM_SYRK)
        b = mpc2x.solve(recs)
        Mb, Lb = mpc2x.debug_condensed(recs[5])[:2]
    finally:
        mpc2x.set_kernel_form(prev)
    nz = 188   # 6 x 24 reduced joint unknowns + 44 throttles
    assert relerr(np.tril(Ma[:nz + 1, :nz]), np.tril(Mb[:nz + 1, :nz])) < 1e-13
    assert relerr(np.tril(La[:nz + 1, :nz]), np.tril(Lb[:nz + 1, :nz])) < 1e-11
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[3], b[3])
    assert relerr(a[0], b[0]) < 1e-10 and relerr(a[1], b[1]) < 1e-10


def test_horizon2x_golden_and_oracle(mpc2x, ref, synth, layout, golden_h2x):
    """BASELINE.json configs[4]: 2x horizon at halved fast-rate dt (1146 variables, 994 rows, condensed dimension 188).
    The factor stays in registers + the LDS panel ring for this variant too (one workgroup per CU); many throttle bounds are
    active."""
    cfg, rcfg = layout.horizon2x_config(), ref.horizon2x_config()
    assert mpc2x.n_var == 1146 and mpc2x.n_con == 994 and mpc2x.n_in == 414 and mpc2x.n_p == 192
    x, fm, st, it = mpc2x.solve(golden_h2x["inputs"])
    assert (st == layout.STATUS_SOLVED).all()
    for b in range(len(x)):
        assert relerr(x[b], golden_h2x["x"][b]) < TOL
        assert relerr(fm[b], golden_h2x["first_move"][b]) < TOL
    np.testing.assert_array_equal(it, golden_h2x["iters"])
    recs = np.concatenate([synth.make_batch(cfg, 8, workload=w, first_index=30) for w in ("hover", "takeoff", "montecarlo")])
    x, fm, st, it = mpc2x.solve(recs)
    assert (st == layout.STATUS_SOLVED).all()
    nact = 0
    vmin, vmax = ref.throttle_bounds(rcfg)
    for b, rec in enumerate(recs):
        xr, yr, itr, _ = ref.solve_instance(rcfg, rec)
        assert relerr(x[b], xr) < TOL and it[b] == itr
        assert relerr(fm[b], ref.first_move_vector(rcfg, xr)) < TOL
        v = xr[rcfg.off_throttle:]
        nact += int((np.abs(v - vmax) < 1e-12).sum() + (np.abs(v - vmin) < 1e-12).sum())
    assert nact > 20 and it.max() >= 3


def test_horizon2x_blocks_and_batch(mpc2x, ref, synth, layout):
    cfg, rcfg = layout.horizon2x_config(), ref.horizon2x_config()
    recs = synth.make_batch(cfg, 2, workload="hover")
    A, Bj, Bt, c, dt = mpc2x.linearize(recs)
    np.testing.assert_allclose(dt, ref.dt_schedule(rcfg), rtol=0, atol=1e-17)
    Ar, Bjr, Btr, cr = ref.linearize(rcfg, recs[1])
    assert relerr(A[1], Ar) < 1e-13 and relerr(c[1], cr) < 1e-13
    import algo_model
    M, Lf = mpc2x.debug_condensed(recs[1])
    Me, ge, Le = algo_model.reduced_condensed(rcfg, ref, recs[1])
    nz = Me.shape[0]
    assert nz == 188
    Mh = np.tril(M[:nz, :nz]); Mh = Mh + np.tril(Mh, -1).T
    assert relerr(Mh, Me) < 1e-11 and relerr(M[nz, :nz], ge) < 1e-11
    assert relerr(np.tril(Lf[:nz, :nz]), Le) < 1e-10
    # a full 256-instance launch: determinism + feasibility properties
    big = synth.make_batch(cfg, 64, workload="takeoff")
    big = np.tile(big, (4, 1))
    x, fm, st, it = mpc2x.solve(big)
    assert (st == layout.STATUS_SOLVED).all()
    for k in range(1, 4):
        np.testing.assert_array_equal(x[:64], x[64 * k:64 * (k + 1)])
    np.testing.assert_array_equal(x[:, 0:26], big[:, 0:26])
    vmin, vmax = ref.throttle_bounds(rcfg)
    assert x[:, rcfg.off_throttle:].min() >= vmin and x[:, rcfg.off_throttle:].max() <= vmax


def test_general_horizon_21_9_15(solver_mod, ref, synth, layout):
    """A horizon that is neither BASELINE configuration (variableSamplingMPC.cpp:24-45 sizes the reference from the XML;
    here every horizon of csrc/vsmpc_horizons.def has a kernel): controlHorizon odd, so joint rows share a 16-row tile
    with throttle rows and the throttle block spans three tile rows (the general corner sweep / primal box QP paths)."""
    cfg = layout.MPCConfig(n_iter=21, n_iter_small=9, control_horizon=15)
    rcfg = ref.Config(n_iter=21, n_iter_small=9, control_horizon=15)
    assert cfg.n_var == 26 * 22 + 8 * 15 + 4 * 7 and cfg.n_con == 26 * 22 + 4 * 13
    m = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=64)
    try:
        assert "21,9,15" in m.kernel_name.replace(" ", "") and m.n_p == 128
        recs = np.concatenate([synth.make_batch(cfg, 12, workload=w, first_index=5) for w in ("hover", "takeoff", "montecarlo")])
        A, Bj, Bt, c, dt = m.linearize(recs[:2])
        np.testing.assert_allclose(dt, ref.dt_schedule(rcfg), rtol=0, atol=1e-17)
        x, fm, st, it = m.solve(recs)
        assert (st == layout.STATUS_SOLVED).all()
        multi = 0
        for b, rec in enumerate(recs):
            xr, _, itr, _ = ref.solve_instance(rcfg, rec)
            assert relerr(x[b], xr) < TOL, (b, relerr(x[b], xr))
            assert it[b] == itr, (b, it[b], itr)
            assert relerr(fm[b], ref.first_move_vector(rcfg, xr)) < TOL
            multi += itr > 1
        assert multi > 0
        # saturated throttles on a free tick: the primal box QP with joint rows inside the first corner tile
        sat = synth.make_batch(cfg, 4, workload="hover", first_index=40)
        sat[:, layout.IN_HOLD] = 0.0
        sat[:, layout.IN_XREF + 2::12] += 30.0
        sat[:, 22] = sat[:, 2] - sat[:, layout.IN_XREF + 2]
        x, fm, st, it = m.solve(sat)
        for b, rec in enumerate(sat):
            xr, _, itr, _ = ref.solve_instance(rcfg, rec)
            assert st[b] == 1 and relerr(x[b], xr) < TOL and it[b] == itr, (b, relerr(x[b], xr), it[b], itr)
        assert it.max() >= 3
    finally:
        m.close()
    with pytest.raises(Exception) as e:                      # a horizon without an instantiation is refused, not mis-solved
        solver_mod.BatchedVSMPC(layout.MPCConfig(n_iter=20, n_iter_small=5, control_horizon=9), device=0, max_batch=4)
    assert "unsupported" in str(e.value).lower()


def test_use_jet_dynamic_false(solver_mod, ref, synth, layout):
    """useJetDynamic = false (systemDynamicsVSMPC.cpp:384-429: the thrust itself is the input, Bt[12+i][i] = 1, no
    second-order jet rows): the other branch of P0 against the oracle."""
    cfg = layout.MPCConfig(use_jet_dynamic=False)
    rcfg = ref.Config(use_jet_dynamic=False)
    m = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=32)
    try:
        recs = np.concatenate([synth.make_batch(cfg, 8, workload=w) for w in ("hover", "takeoff")])
        A, Bj, Bt, c, dt = m.linearize(recs)
        for b, rec in enumerate(recs):
            Ar, Bjr, Btr, cr = ref.linearize(rcfg, rec)
            assert relerr(A[b], Ar) < 1e-13 and relerr(Bt[b], Btr) < 1e-13 and relerr(c[b], cr) < 1e-13
            assert np.array_equal(Bt[b] == 0, Btr == 0) and Bt[b][12, 0] == 1.0
        x, fm, st, it = m.solve(recs)
        assert (st == layout.STATUS_SOLVED).all()
        for b, rec in enumerate(recs):
            xr, _, itr, _ = ref.solve_instance(rcfg, rec)
            assert relerr(x[b], xr) < TOL and it[b] == itr, (b, relerr(x[b], xr))
    finally:
        m.close()


def test_batch_larger_than_handle_is_refused_on_the_device_entry(solver_mod, synth, layout):
    import torch
    cfg = layout.paper_config()
    small = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=4)
    try:
        dev = torch.device("cuda:0")
        d_in = torch.from_numpy(synth.make_batch(cfg, 8)).to(dev)
        d_st = torch.zeros(8, dtype=torch.int32, device=dev)
        with pytest.raises(Exception) as e:
            small.solve_device(d_in, None, None, d_st, None)
        assert "max_batch" in str(e.value)
    finally:
        small.close()
