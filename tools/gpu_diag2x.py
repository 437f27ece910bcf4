"""Diagnostic (GPU): per-tile comparison of the condensed Hessian M and the Cholesky factor L of the 2x-horizon kernel
against numpy, twice (run-to-run determinism).  python tools/gpu_diag2x.py [paper]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa
import vsmpc_ref as ref
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
L = importlib.import_module(PKG + ".layout"); S = importlib.import_module(PKG + ".synth"); solver = importlib.import_module(PKG + ".solver")
paper = len(sys.argv) > 1 and sys.argv[1] == "paper"
cfg, rcfg = (L.paper_config(), ref.paper_config()) if paper else (L.horizon2x_config(), ref.horizon2x_config())
m = solver.BatchedVSMPC(cfg, device=0, max_batch=64)
rec = S.make_batch(cfg, 2, workload="hover")[1]
nz = cfg.n_inputs; nxs = 26 * (cfg.n_iter + 1)
H, g, Ac, lo, hi = ref.assemble_dense(rcfg, rec)
sol = np.linalg.solve(Ac[:nxs, :nxs], np.column_stack([lo[:nxs], Ac[:nxs, nxs:]]))
Z = np.vstack([-sol[:, 1:], np.eye(nz)]); xp = np.concatenate([sol[:, 0], np.zeros(nz)])
Hr, gr = Z.T @ H @ Z, Z.T @ (H @ xp + g)
nu = 8 * cfg.control_horizon
perm = list(range(nu)) + list(range(nu + 4, nz)) + list(range(nu, nu + 4))
Hr, gr = Hr[np.ix_(perm, perm)], gr[perm]
Lr = np.linalg.cholesky(0.5 * (Hr + Hr.T))
runs = [m.debug_condensed(rec) for _ in range(3)]
NT = m.n_p // 16
for name, idx, refm in (("M", 0, np.tril(Hr)), ("L", 1, Lr)):
    A = [np.tril(r[idx][:nz, :nz]) for r in runs]
    print(name, "run-to-run max diff", np.abs(A[0] - A[1]).max(), np.abs(A[0] - A[2]).max(), " vs numpy", np.abs(A[0] - refm).max() / np.abs(refm).max())
    E = np.abs(A[0] - refm) / np.abs(refm).max()
    D = np.abs(A[0] - A[1])
    for i in range(NT):
        print("  row %2d err " % i + " ".join("%7.0e" % E[16*i:16*i+16, 16*j:16*j+16].max() if E[16*i:16*i+16, 16*j:16*j+16].size else "      -" for j in range(i + 1)))
    if D.max() > 0:
        for i in range(NT):
            print("  row %2d r2r " % i + " ".join("%7.0e" % D[16*i:16*i+16, 16*j:16*j+16].max() if D[16*i:16*i+16, 16*j:16*j+16].size else "      -" for j in range(i + 1)))
recs = S.make_batch(cfg, 64, workload="hover")
xs = [m.solve(recs)[0] for _ in range(3)]
print("solve run-to-run", np.abs(xs[0] - xs[1]).max(), np.abs(xs[0] - xs[2]).max())
xr = np.array([ref.solve_instance(rcfg, r)[0] for r in recs[:8]])
e = np.abs(xs[0][:8] - xr)
print("err states", e[:, :nxs].max(), "joints", e[:, nxs:nxs + nu].max(), "throttles", e[:, nxs + nu:].max())
# which trailing update is tile (10,7) missing?  pre-panel tile = L_10,7 L_77^T should equal M_10,7 - sum_k<7 L_10,k L_7,k^T
if not paper:
    Mg, Lg = runs[0]
    T = lambda A, i, j: A[16*i:16*i+16, 16*j:16*j+16]
    for (i, j) in ((10, 7), (9, 7), (10, 8)):
        pre_gpu = sum(T(Lg, i, k) @ np.tril(T(Lg, j, k)).T if k == j else T(Lg, i, k) @ T(Lg, j, k).T for k in range(j, j + 1))
        pre_gpu = T(Lg, i, j) @ np.tril(T(Lg, j, j)).T
        pre_ref = T(Mg, i, j) - sum(T(Lr, i, k) @ T(Lr, j, k).T for k in range(j))
        E = pre_gpu - pre_ref
        print("tile", (i, j), "pre-panel error", np.abs(E).max(), " contributions:", " ".join("%d:%.1e/%.1e" % (k, np.abs(T(Lr, i, k) @ T(Lr, j, k).T).max(), np.abs(E - T(Lr, i, k) @ T(Lr, j, k).T).max()) for k in range(j)))
    np.set_printoptions(linewidth=250, precision=1)
    i, j = 10, 7
    pre_gpu = T(Lg, i, j) @ np.tril(T(Lg, j, j)).T
    pre_ref = T(Mg, i, j) - sum(T(Lr, i, k) @ T(Lr, j, k).T for k in range(j))
    print(pre_gpu - pre_ref)
    print("L tile error"); print(T(Lg, i, j) - T(Lr, i, j))
