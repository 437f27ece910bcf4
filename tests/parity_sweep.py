"""Long parity sweep: the HIP path against the numpy oracle on many seeded instances of every workload, through the
C-ABI.  Two phases so that the oracle's worker processes never share a process tree with an initialised GPU:

    python tests/parity_sweep.py --oracle-only out.npz [instances_per_workload]     (CPU only)
    python tests/parity_sweep.py [instances_per_workload]                           (both phases, prints one JSON line)

`compare(npz)` is phase 2 (GPU); tests/test_gpu_full_configs.py::test_parity_sweep_collected runs both.
"""
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
SEED0 = 20250
WORKLOADS = ("hover", "takeoff", "montecarlo")


def _plan(n):
    layout = importlib.import_module(PKG + ".layout")
    return [("paper", layout.paper_config(), n), ("horizon2x", layout.horizon2x_config(), max(16, n // 8))]


def _oracle_chunk(args):
    cfg_name, recs = args
    import vsmpc_ref as ref
    rcfg = ref.paper_config() if cfg_name == "paper" else ref.horizon2x_config()
    out = []
    for rec in recs:
        x, _, it, _ = ref.solve_instance(rcfg, rec)
        out.append((x, it))
    return out


def oracle_phase(path, n):
    """Oracle solutions of every case into one npz (CPU only; worker processes with one BLAS thread each)."""
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    import multiprocessing as mp
    synth = importlib.import_module(PKG + ".synth")
    t0 = time.time()
    arrays = {"n": np.array(n)}
    with mp.get_context("spawn").Pool(min(14, os.cpu_count() or 1)) as pool:
        for cfg_name, cfg, count in _plan(n):
            for wl in WORKLOADS:
                recs = synth.make_batch(cfg, count, workload=wl, seed0=SEED0)
                chunks = [(cfg_name, recs[i:i + 8]) for i in range(0, count, 8)]
                res = []
                for k, chunk in enumerate(pool.imap(_oracle_chunk, chunks)):
                    res.extend(chunk)
                    if k % 16 == 0:
                        print(f"  {cfg_name} {wl}: {len(res)}/{count} oracle solves, {time.time() - t0:.0f} s",
                              file=sys.stderr, flush=True)
                arrays[f"{cfg_name}:{wl}:x"] = np.array([r[0] for r in res])
                arrays[f"{cfg_name}:{wl}:it"] = np.array([r[1] for r in res])
    np.savez(path, **arrays)
    return time.time() - t0


def compare(path):
    """HIP path through the C-ABI against the stored oracle solutions."""
    import torch  # noqa: F401  (HIP runtime first)
    import __graft_entry__ as ge
    ge.build()
    solver = importlib.import_module(PKG + ".solver")
    synth = importlib.import_module(PKG + ".synth")
    data = np.load(path)
    n = int(data["n"])
    report = {"instances_per_workload": n, "seed0": SEED0, "cases": []}
    for cfg_name, cfg, count in _plan(n):
        for wl in WORKLOADS:
            recs = synth.make_batch(cfg, count, workload=wl, seed0=SEED0)
            xr, it_ref = data[f"{cfg_name}:{wl}:x"], data[f"{cfg_name}:{wl}:it"]
            mpc = solver.BatchedVSMPC(cfg, device=0, max_batch=len(recs))
            x, fm, st, it = mpc.solve(recs)
            mpc.close()
            scale = np.maximum(1.0, np.abs(xr).max(axis=1))
            err = float((np.abs(x - xr).max(axis=1) / scale).max())
            report["cases"].append({"config": cfg_name, "workload": wl, "instances": len(recs),
                                    "max_rel_err": err, "all_solved": bool((st == 1).all()),
                                    "iterations_equal": bool((it == it_ref).all()),
                                    "instances_with_active_set_iterations": int((it_ref > 1).sum()),
                                    "max_iterations": int(it_ref.max())})
    report["worst_rel_err"] = max(c["max_rel_err"] for c in report["cases"])
    return report


def main():
    args = sys.argv[1:]
    if args and args[0] == "--oracle-only":
        n = int(args[2]) if len(args) > 2 else 768
        secs = oracle_phase(args[1], n)
        print(f"oracle phase: {secs:.0f} s", file=sys.stderr)
        return
    n = int(args[0]) if args else 512
    t0 = time.time()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "oracle.npz")
        oracle_phase(path, n)
        report = compare(path)
    report["seconds"] = time.time() - t0
    print(json.dumps(report))


if __name__ == "__main__":
    main()
