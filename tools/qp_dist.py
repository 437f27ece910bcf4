import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
import __graft_entry__ as ge
ge.build()
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); solver = importlib.import_module(PKG + ".solver")
cfg = pkg.horizon2x_config()
X = synth.make_batch(cfg, 256, workload="hover")
m = solver.BatchedVSMPC(cfg, device=0, max_batch=256)
st = m.phase_cycles(X).astype(np.int64)
x, fm, status, iters = m.solve(X)
qp = np.diff(st[:, :10], axis=1)[:, 5]
v = x[:, cfg.off_throttle:cfg.off_throttle + 44]
vmin, vmax = v.min(), v.max()
nact = ((np.abs(v - vmin) < 1e-12) | (np.abs(v - vmax) < 1e-12)).sum(axis=1)
for it in sorted(set(iters)):
    sel = iters == it
    print("iters", it, "count", sel.sum(), "P4b median", int(np.median(qp[sel])), "min", qp[sel].min(), "max", qp[sel].max(), "active at solution median", np.median(nact[sel]))
