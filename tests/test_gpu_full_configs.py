"""BASELINE.json configs[2], [3], [4] at FULL size on one GPU, through the C-ABI, plus the long parity sweep.

    configs[2]  batch=4096 take-off trajectory, 4096 DISTINCT tick placements / disturbance seeds
    configs[3]  32768 Monte-Carlo initial states over 8 GPUs: the 4096-instance slices of ranks 0 and 7, seeds exactly
                as sharding.rank_inputs builds them on an 8-GPU node (4x wider sigma)
    configs[4]  2x horizon at halved fast-rate dt, batch=4096 on the long-horizon kernel

Each case: size-independent properties on all 4096 instances (status, determinism, batch-permutation invariance,
X0 pinned bit-exactly, first-move block == primal slices, throttle box, 20-tick hold) + >= 32 instances picked across
the active-set sizes that occur, solved by the oracle and compared at the build's bar (1e-8 relative; north_star
allows 1e-4) with equal active-set iteration counts + KKT certificates on the reference-ordered dense QP.
"""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, relerr
from test_gpu_parity import _kkt_properties

pytestmark = pytest.mark.gpu
TOL = 1e-8
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"


def _pick_by_active_set(nact, iters, want=32):
    """Indices spread over the active-set sizes present: round-robin over the distinct sizes, largest
    iteration counts first within a size."""
    groups = {}
    for b in np.lexsort((-iters, nact)):
        groups.setdefault(int(nact[b]), []).append(int(b))
    chosen, depth = [], 0
    while len(chosen) < want and any(depth < len(g) for g in groups.values()):
        for k in sorted(groups):
            if depth < len(groups[k]) and len(chosen) < want:
                chosen.append(groups[k][depth])
        depth += 1
    return chosen


def _full_size_case(mpc, ref, rcfg, layout, recs, *, want=32, kkt=6):
    B = len(recs)
    off_j, off_v, nvar = rcfg.off_joints, rcfg.off_throttle, rcfg.n_var
    x, fm, st, it = mpc.solve(recs)
    assert (st == layout.STATUS_SOLVED).all(), np.unique(st, return_counts=True)
    # determinism, batch-position independence, checksum of checksums
    x2, fm2, st2, it2 = mpc.solve(recs)
    np.testing.assert_array_equal(x, x2)
    np.testing.assert_array_equal(it, it2)
    perm = np.random.default_rng(5).permutation(B)
    xp, _, _, itp = mpc.solve(recs[perm])
    np.testing.assert_array_equal(xp, x[perm])
    np.testing.assert_array_equal(itp, it[perm])
    assert float(np.sum(np.sum(x, axis=1))) == float(np.sum(np.sum(x2, axis=1)))
    # boundary facts of the path
    np.testing.assert_array_equal(x[:, 0:26], recs[:, 0:26])                       # X0 = x_meas (IQPUtilsMPC.cpp:71-92)
    np.testing.assert_array_equal(fm[:, 0:8], x[:, off_j:off_j + 8])               # variableSamplingMPC.cpp:99
    np.testing.assert_array_equal(fm[:, 8:12], x[:, off_v:off_v + 4])              # :100
    np.testing.assert_array_equal(fm[:, 16:24], x[:, 26 + 12:26 + 20])             # node-1 thrusts (:101-102)
    np.testing.assert_allclose(fm[:, 12:16], ref.destd_throttle(x[:, off_v:off_v + 4]), rtol=1e-9, atol=1e-9)
    vmin, vmax = ref.throttle_bounds(rcfg)
    v = x[:, off_v:nvar]
    assert v.min() >= vmin and v.max() <= vmax
    hold = recs[:, layout.IN_HOLD] != 0
    vprev = ref.v_of_throttle(recs[:, layout.IN_UPREV:layout.IN_UPREV + 4])
    np.testing.assert_allclose(x[hold, off_v:off_v + 4], vprev[hold], rtol=0, atol=1e-15)
    # oracle on instances spread over the active-set sizes
    nact = ((v == vmin) | (v == vmax)).sum(axis=1)      # bound variables sit exactly on their bound
    chosen = _pick_by_active_set(nact, it, want)
    assert len(chosen) >= min(want, B)
    worst = 0.0
    for b in chosen:
        xr, _, itr, _ = ref.solve_instance(rcfg, recs[b])
        e = relerr(x[b], xr)
        worst = max(worst, e)
        assert e < TOL, (b, int(nact[b]), e)
        assert it[b] == itr, (b, int(nact[b]), int(it[b]), itr)
        assert relerr(fm[b], ref.first_move_vector(rcfg, xr)) < TOL
    if rcfg.n_var == 588:
        _kkt_properties(ref, rcfg, recs, x, chosen[:kkt])
    return {"nact_sizes": sorted(set(int(n) for n in nact[chosen])), "max_iters": int(it.max()), "worst": worst,
            "with_qp": int((it > 1).sum())}


@pytest.fixture(scope="module")
def mpc(solver_mod, layout):
    m = solver_mod.BatchedVSMPC(layout.paper_config(), device=0, max_batch=4096)
    yield m
    m.close()


def test_config2_takeoff_4096_distinct_seeds(mpc, ref, synth, layout):
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    recs = synth.make_batch(cfg, 4096, workload="takeoff")
    assert len(np.unique(recs[:, layout.IN_ALPHA])) > 1000          # really 4096 different tick placements
    info = _full_size_case(mpc, ref, rcfg, layout, recs)
    assert info["with_qp"] > 100 and len(info["nact_sizes"]) >= 4, info


@pytest.mark.parametrize("rank", [0, 7])
def test_config3_montecarlo_rank_slice(mpc, ref, synth, layout, pkg, rank):
    sharding = importlib.import_module(PKG + ".sharding")
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    recs = sharding.rank_inputs(cfg, synth, 32768, rank, 8, workload="montecarlo")
    assert recs.shape == (4096, cfg.n_in)
    first, count = sharding.shard_range(32768, rank, 8)
    assert (first, count) == (4096 * rank, 4096)
    # the slice is what a single-process build of the whole batch would hold at those positions
    np.testing.assert_array_equal(recs[:3], synth.make_batch(cfg, 3, workload="montecarlo", first_index=first))
    info = _full_size_case(mpc, ref, rcfg, layout, recs)
    assert info["with_qp"] > 20, info


def test_config4_horizon2x_4096(solver_mod, ref, synth, layout):
    cfg, rcfg = layout.horizon2x_config(), ref.horizon2x_config()
    m = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=4096)
    try:
        assert "34,14,24" in m.kernel_name.replace(" ", "")
        recs = synth.make_batch(cfg, 4096, workload="hover")
        info = _full_size_case(m, ref, rcfg, layout, recs)
        assert info["with_qp"] > 1000 and info["max_iters"] >= 4, info   # the long horizon saturates throttles
    finally:
        m.close()


def test_parity_sweep_collected(solver_mod, layout, synth, tmp_path):
    """tests/parity_sweep.py as a collected test: 3 workloads x 768 instances at the paper horizon and x 96 at the 2x
    horizon (2,592 oracle solves in worker processes that never touch the GPU), every one compared."""
    out = tmp_path / "oracle.npz"
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "parity_sweep.py"), "--oracle-only", str(out), "768"],
                   check=True, env=env, timeout=1500)
    import parity_sweep
    report = parity_sweep.compare(str(out))
    print(json.dumps(report))
    assert report["worst_rel_err"] < TOL
    for c in report["cases"]:
        assert c["all_solved"] and c["iterations_equal"], c
    assert sum(c["instances"] for c in report["cases"]) == 2592
    assert sum(c["instances_with_active_set_iterations"] for c in report["cases"]) > 300
