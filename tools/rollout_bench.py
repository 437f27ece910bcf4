#!/usr/bin/env python3
"""Closed-loop rollout throughput: `batch` resident loops advanced `ticks` MPC periods (3 kernels per tick).
Usage: python tools/rollout_bench.py [batch] [ticks] [workload] [jetnn|tree]
`jetnn` selects the jet plant option (LSTM thrust dynamics + EKF, weights of tests/golden/jet_lstm.npz), `tree` the
kinematic-tree plant (provider + kinematics kernels on the plant's joints every tick: six launches per tick)."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    workload = sys.argv[3] if len(sys.argv) > 3 else "hover"
    jetnn = len(sys.argv) > 4 and sys.argv[4] == "jetnn"
    tree = len(sys.argv) > 4 and sys.argv[4] == "tree"
    import torch  # noqa: F401  (HIP runtime first)
    import __graft_entry__ as ge
    ge.build()
    pkg = importlib.import_module(PKG)
    ro = importlib.import_module(PKG + ".rollout")
    cfg = pkg.paper_config()
    st, pa = ro.make_plant(cfg, batch, workload=workload)
    tr = None
    if tree:
        tr = importlib.import_module(PKG + ".robot_tree").default_tree()
        st, pa = ro.make_plant_tree(cfg, batch, tr, workload=workload)
    pos, vel, alpha, adt = ro.make_trajectory(cfg, "takeoff" if workload == "takeoff" else "hover", 60.0)
    r = ro.ClosedLoopRollout(cfg, batch, pos, vel, alpha, adt)
    if tree:
        r.set_tree(tr)
    jm = None
    if jetnn:
        import numpy as np
        g = np.load(os.path.join(ROOT, "tests", "golden", "jet_lstm.npz"))
        jm = importlib.import_module(PKG + ".jet_plant").JetModelTotal(g["w_ih"], g["w_hh"], g["b_ih"], g["b_hh"], g["fc_w"],
                                                                       g["fc_b"], g["norm"], device=0, max_series=64)
        r.set_jet_plant(jm)
    r.reset(st, pa)
    r.run(50, log=False)
    r.reset(st, pa)
    t0 = time.perf_counter()
    log = r.run(ticks, log=True)
    dt = time.perf_counter() - t0
    solved = float((log[:, :, 14] == 1).mean())
    print(json.dumps({"what": "closed-loop rollout", "batch": batch, "ticks": ticks, "workload": workload,
                      "us_per_tick": 1e6 * dt / ticks, "instance_ticks_per_s": batch * ticks / dt,
                      "realtime_factor": batch * ticks * cfg.period_mpc / dt, "solved_fraction": solved,
                      "mean_active_set_iters": float(log[:, :, 15].mean()), "jet_plant": "lstm+ekf" if jetnn else "polynomial", "plant": "kinematic tree" if tree else "parametric",
                      "final_altitude_error_mean_m": float(abs(log[-1, :, 2] - pa[:, 236] - (pos[min(len(pos) - 1, 0), 2])).mean())
                      if workload != "takeoff" else None,
                      "final_thrust_mean_N": float(log[-1, :, 6:10].mean())}))
    r.close()


if __name__ == "__main__":
    main()
