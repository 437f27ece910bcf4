"""Manual, long parity sweep (not collected by pytest): the HIP path against the numpy oracle on many seeded instances
of every workload, through the C-ABI.  Prints one JSON line; the committed result is profiles/r01_parity_sweep.json.

    python tests/parity_sweep.py [instances_per_workload]
"""
import importlib
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"


def _oracle_chunk(args):
    cfg_name, recs = args
    import vsmpc_ref as ref
    rcfg = ref.paper_config() if cfg_name == "paper" else ref.horizon2x_config()
    out = []
    for rec in recs:
        x, _, it, _ = ref.solve_instance(rcfg, rec)
        out.append((x, it))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"            # one BLAS thread per worker process (they are spawned with this environment)
    import multiprocessing as mp
    layout = importlib.import_module(PKG + ".layout")
    synth = importlib.import_module(PKG + ".synth")
    plan = [("paper", layout.paper_config(), n), ("horizon2x", layout.horizon2x_config(), max(16, n // 8))]
    t0 = time.time()
    # phase 1: oracle solutions in spawned worker processes, before anything in this process touches the GPU
    cases = []
    with mp.get_context("spawn").Pool(min(14, os.cpu_count() or 1)) as pool:
        for cfg_name, cfg, count in plan:
            for wl in ("hover", "takeoff", "montecarlo"):
                recs = synth.make_batch(cfg, count, workload=wl, seed0=20250)
                chunks = [(cfg_name, recs[i:i + 8]) for i in range(0, count, 8)]
                res = []
                for k, chunk in enumerate(pool.imap(_oracle_chunk, chunks)):
                    res.extend(chunk)
                    if k % 16 == 0:
                        print(f"  {cfg_name} {wl}: {len(res)}/{count} oracle solves, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
                cases.append((cfg_name, cfg, wl, recs, res))
                print(f"oracle done: {cfg_name} {wl} {count} instances, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    # phase 2: the HIP path through the C-ABI
    import torch  # noqa: F401  (HIP runtime first)
    import __graft_entry__ as ge
    ge.build()
    solver = importlib.import_module(PKG + ".solver")
    report = {"instances_per_workload": n, "seed0": 20250, "cases": []}
    for cfg_name, cfg, wl, recs, res in cases:
        mpc = solver.BatchedVSMPC(cfg, device=0, max_batch=len(recs))
        x, fm, st, it = mpc.solve(recs)
        mpc.close()
        err = max(float(np.abs(x[b] - xr).max() / max(1.0, np.abs(xr).max())) for b, (xr, _) in enumerate(res))
        it_ref = np.array([r[1] for r in res])
        report["cases"].append({"config": cfg_name, "workload": wl, "instances": len(recs),
                                "max_rel_err": err, "all_solved": bool((st == 1).all()),
                                "iterations_equal": bool((it == it_ref).all()),
                                "instances_with_active_set_iterations": int((it_ref > 1).sum()),
                                "max_iterations": int(it_ref.max())})
        print(report["cases"][-1], file=sys.stderr, flush=True)
    report["seconds"] = time.time() - t0
    report["worst_rel_err"] = max(c["max_rel_err"] for c in report["cases"])
    print(json.dumps(report))


if __name__ == "__main__":
    main()
