"""GPU box diagnostic for the 2x-horizon variant: stage-by-stage errors vs the oracle."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); solver = importlib.import_module(PKG + ".solver")
import vsmpc_ref as R
def rel(a, b): return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
cfg = pkg.horizon2x_config(); rcfg = R.horizon2x_config()
recs = synth.make_batch(cfg, 4, workload="hover")
m = solver.BatchedVSMPC(cfg, device=0, max_batch=8)
A, Bj, Bt, c, dt = m.linearize(recs)
Ar, Bjr, Btr, cr = R.linearize(rcfg, recs[1])
print("linearize", rel(A[1], Ar), rel(c[1], cr))
M, Lf = m.debug_condensed(recs[1])
H, g, Ac, lo, hi = R.assemble_dense(rcfg, recs[1])
nxs = 26 * 35
sol = np.linalg.solve(Ac[:nxs, :nxs], np.column_stack([lo[:nxs], Ac[:nxs, nxs:]]))
Z = np.vstack([-sol[:, 1:], np.eye(236)]); xp = np.concatenate([sol[:, 0], np.zeros(236)])
Hr, gr = Z.T @ H @ Z, Z.T @ (H @ xp + g)
perm = list(range(192)) + list(range(196, 236)) + list(range(192, 196))
Hr, gr = Hr[np.ix_(perm, perm)], gr[perm]
Mh = np.tril(M[:236, :236]); Mh = Mh + np.tril(Mh, -1).T
print("M err", rel(Mh, Hr), "grad err", rel(M[236, :236], gr))
E = np.abs(Mh - Hr) / np.abs(Hr).max()
bad = np.argwhere(E > 1e-9)
print("bad entries", len(bad), bad[:10].tolist())
tiles = sorted({(int(r) // 16, int(c) // 16) for r, c in bad if c <= r})
print("bad tiles (lower)", tiles[:40])
Lr = np.linalg.cholesky(0.5 * (Hr + Hr.T))
print("L err", rel(np.tril(Lf[:236, :236]), Lr), "nan in L", np.isnan(Lf).sum())
x, fm, st, it = m.solve(recs)
print("status", st, "iters", it)
for b in range(4):
    xr, yr, itr, _ = R.solve_instance(rcfg, recs[b])
    print(b, "err", rel(x[b], xr), "oracle iters", itr)
