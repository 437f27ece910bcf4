import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); solver = importlib.import_module(PKG + ".solver")
cfg = pkg.paper_config()
for wl in ("takeoff", "montecarlo"):
    X = synth.make_batch(cfg, 256, workload=wl)
    m = solver.BatchedVSMPC(cfg, device=0, max_batch=256)
    st = m.phase_cycles(X).astype(np.int64)
    x, fm, status, iters = m.solve(X)
    d = np.diff(st[:, :10], axis=1)
    qp = d[:, 5]; tot = st[:, 9] - st[:, 0]
    print("==", wl, "mean total", int(tot.mean()), "median", int(np.median(tot)), "mean P4b", int(qp.mean()), "frac needing qp", float((iters > 1).mean()))
    for it in sorted(set(iters)):
        sel = iters == it
        print("  iters", it, "count", sel.sum(), "P4b median", int(np.median(qp[sel])), "max", qp[sel].max())
    m.close()
