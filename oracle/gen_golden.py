"""Generates tests/golden/*.npz from the numpy oracle (oracle/vsmpc_ref.py).

The reference cannot be built or imported in this image, so these vectors are
outputs of the ORACLE, not of the reference: parity stays "unpinned"; the fixtures freeze the
oracle's answers so that later edits to it (or to the workload generator) cannot drift silently,
and they give the GPU tests fixed inputs/outputs that travel with the repo.

    python oracle/gen_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vsmpc_ref as R  # noqa: E402

PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
synth = importlib.import_module(PKG + ".synth")
layout = importlib.import_module(PKG + ".layout")


def build(name, rcfg, pcfg, picks):
    inputs = []
    for workload, idx in picks:
        inputs.append(synth.make_batch(pcfg, 1, workload=workload, first_index=idx)[0])
    inputs = np.stack(inputs)
    out = {"inputs": inputs, "workloads": np.array([f"{w}:{i}" for w, i in picks])}
    A, Bj, Bt, c, g, lo, hi, x, y, fm, iters, cert = [], [], [], [], [], [], [], [], [], [], [], []
    Hdiag = None
    for rec in inputs:
        a, bj, bt, cc = R.linearize(rcfg, rec)
        xs, ys, it, (H, gg, Ac, l, u) = R.solve_instance(rcfg, rec)
        k = R.kkt_certificate(H, gg, Ac, l, u, xs, ys)
        assert k["stationarity_rel"] < 1e-12 and k["primal"] < 1e-10, k
        A.append(a); Bj.append(bj); Bt.append(bt); c.append(cc); g.append(gg); lo.append(l); hi.append(u)
        x.append(xs); y.append(ys); fm.append(R.first_move_vector(rcfg, xs)); iters.append(it)
        cert.append([k["stationarity_rel"], k["primal"], k["objective"]])
        Hdiag = np.diag(H).copy()
    out.update(A=np.stack(A), Bj=np.stack(Bj), Bt=np.stack(Bt), c=np.stack(c), g=np.stack(g), lo=np.stack(lo),
               hi=np.stack(hi), x=np.stack(x), y=np.stack(y), first_move=np.stack(fm), iters=np.array(iters),
               certificate=np.array(cert), H_diag=Hdiag, dt=R.dt_schedule(rcfg))
    path = os.path.join(ROOT, "tests", "golden", name)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB", "active-set iterations", iters)


if __name__ == "__main__":
    build("vsmpc_golden_paper.npz", R.paper_config(), layout.paper_config(),
          [("hover", 0), ("hover", 1), ("hover", 19), ("takeoff", 0), ("takeoff", 1), ("takeoff", 5),
           ("montecarlo", 4), ("montecarlo", 0)])
    build("vsmpc_golden_horizon2x.npz", R.horizon2x_config(), layout.horizon2x_config(),
          [("hover", 0), ("hover", 1), ("takeoff", 1), ("montecarlo", 2)])
