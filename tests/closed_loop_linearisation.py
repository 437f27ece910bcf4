"""Linearised closed loop of the synthetic plant under the MPC, from the oracle (test infrastructure).

The tick map is periodic with the 20-tick throttle hold, so the object whose spectral radius bounds the long-run decay
is the MONODROMY matrix over one hold period: finite differences of the 20-tick map
    s -> advance(s, first_move(oracle(record(s))))
about the hover equilibrium of one instance (reference window at rest, no bound active near hover), over the plant
state (p, h_lin, rpy, h_ang, T, Tdot, q, u, T_des, Tdot_des).  `spectral_radius()` is what the GPU property test
derives its decay bound from (tests/test_gpu_rollout.py) instead of fitting a threshold to the run.

The full derivation costs ~2,200 oracle solves (minutes), so its result is committed as a fixture:
    python tests/closed_loop_linearisation.py --write     ->  tests/golden/hover_monodromy.npz  (M, orbit point, rho)
tests/test_closed_loop_linearisation.py re-derives the orbit residual and two columns from scratch on every CPU run and
compares them with the fixture; the GPU test loads the fixture."""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"


def period_map(cfg, rcfg, ref, rm, s, p, traj, start_tick=0):
    """State after one hold period (cfg.ratio ticks) of the oracle-in-the-loop model, started right after a release."""
    pos, vel, alpha, adt = traj
    q = p.copy()
    q[importlib.import_module(PKG + ".layout").PP_TICK0] = float(start_tick)
    model = rm.make_tick_model(cfg, s, q, pos, vel, alpha)
    s = s.copy()
    for tick in range(cfg.ratio):
        rec = rm.build_record(cfg, model, s, q)
        x, _, _, _ = ref.solve_instance(rcfg, rec)
        fm = ref.first_move_vector(rcfg, x)
        model.consume(fm, 1)
        s = rm.advance(cfg, s, q, tick, fm, 1, alpha, adt)
    return s


def tree_plant(seed0=4321):
    """One hover instance of the KINEMATIC-TREE plant (vsmpc_rollout_set_tree): the parametric instance of that seed with
    the tree's mass, started with the thrusts that carry it"""
    layout = importlib.import_module(PKG + ".layout")
    ro = importlib.import_module(PKG + ".rollout")
    RT = importlib.import_module(PKG + ".robot_tree")
    cfg = layout.paper_config()
    st, pa = ro.make_plant_tree(cfg, 1, RT.default_tree(), workload="hover", seed0=seed0)
    return RT.default_tree(), st, pa


def monodromy(seed0=4321, eps=1e-6, tree=False, settle=30):
    import rollout_model as rm
    import vsmpc_ref as ref
    layout = importlib.import_module(PKG + ".layout")
    ro = importlib.import_module(PKG + ".rollout")
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    if tree:
        tr, st, pa = tree_plant(seed0)
        rm.set_tree(tr)
    else:
        rm.set_tree(None)
        st, pa = ro.make_plant(cfg, 1, workload="hover", seed0=seed0)
    traj = ro.make_trajectory(cfg, "hover", 5.0)
    p = pa[0].copy()
    # settle onto the periodic orbit near hover first (a few periods), then linearise about that point
    s = st[0].copy()
    for _ in range(settle):
        s = period_map(cfg, rcfg, ref, rm, s, p, traj)
    n = N_LIN
    f0 = period_map(cfg, rcfg, ref, rm, s, p, traj)
    M = np.zeros((n, n))
    scale = fd_scale(layout)
    for i in range(n):
        M[:, i] = column(cfg, rcfg, ref, rm, s, p, traj, i, eps * scale[i])
    return M, s, f0


N_LIN = 40   # plant state proper (p, h_lin, rpy, h_ang, T, Tdot, q, u, T_des, Tdot_des); the fields of the jet plant option
             # behind it are not touched by the polynomial jet plant


def fd_scale(layout):
    scale = np.ones(N_LIN)
    scale[layout.PS_T:layout.PS_T + 4] = 10.0
    scale[layout.PS_TD:layout.PS_TD + 4] = 10.0
    scale[layout.PS_U:layout.PS_U + 4] = 1.0
    scale[layout.PS_TDES:layout.PS_TDES + 8] = 10.0
    return scale


def column(cfg, rcfg, ref, rm, s, p, traj, i, step):
    """Column i of the monodromy matrix by central differences about the orbit point `s`."""
    d = np.zeros(len(s))
    d[i] = step
    return ((period_map(cfg, rcfg, ref, rm, s + d, p, traj) - period_map(cfg, rcfg, ref, rm, s - d, p, traj)) / (2 * step))[:N_LIN]


def setup(seed0=4321):
    """(cfg, rcfg, ref, rm, layout, p, traj) of the instance the linearisation is taken about."""
    import rollout_model as rm
    import vsmpc_ref as ref
    layout = importlib.import_module(PKG + ".layout")
    ro = importlib.import_module(PKG + ".rollout")
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    st, pa = ro.make_plant(cfg, 1, workload="hover", seed0=seed0)
    return cfg, rcfg, ref, rm, layout, st[0].copy(), pa[0].copy(), ro.make_trajectory(cfg, "hover", 5.0)


FIXTURE = os.path.join(ROOT, "tests", "golden", "hover_monodromy.npz")
FIXTURE_TREE = os.path.join(ROOT, "tests", "golden", "hover_monodromy_tree.npz")


def load_fixture(tree=False):
    d = np.load(FIXTURE_TREE if tree else FIXTURE)
    return d["M"], d["orbit"], float(d["rho"])


def spectral_radius(M):
    return float(np.abs(np.linalg.eigvals(M)).max())


if __name__ == "__main__":
    import time
    t = time.time()
    use_tree = "--tree" in sys.argv
    settle = int(sys.argv[sys.argv.index("--settle") + 1]) if "--settle" in sys.argv else 30
    M, s, f0 = monodromy(tree=use_tree, settle=settle)
    ev = np.linalg.eigvals(M)
    print("residual of the orbit:", np.abs(f0 - s).max())
    print("spectral radius per hold period (0.1 s):", np.abs(ev).max(), " time %.0f s" % (time.time() - t))
    print(np.sort(np.abs(ev))[::-1][:10])
    if "--write" in sys.argv:
        np.savez(FIXTURE_TREE if use_tree else FIXTURE, M=M, orbit=s, rho=np.abs(ev).max(), seed0=4321, eps=1e-6, settle=settle)
        print("wrote", FIXTURE_TREE if use_tree else FIXTURE)
