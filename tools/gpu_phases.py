"""GPU box: per-phase cycle shares of the solve kernel from the STAMPS diagnostic instantiation."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
import __graft_entry__ as ge
ge.build()
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); solver = importlib.import_module(PKG + ".solver")
names = ["P0 load+linearise", "P1 condense (propagate + MFMA SYRK)", "P2 augment", "P3 cholesky", "P4a throttle pass",
         "P4b box QP", "P5 back-subst", "P6 simulate", "output"]
import sys as _s
cfg = pkg.horizon2x_config() if (len(_s.argv) > 1 and _s.argv[1] == "h2x") else pkg.paper_config()
for wl, B in ((("hover", 256), ("takeoff", 256)) if len(_s.argv) > 1 else (("hover", 256), ("takeoff", 256), ("hover", 4096))):
    X = synth.make_batch(cfg, min(B, 256), workload=wl)
    if B > 256: X = np.tile(X, (B // 256, 1))
    m = solver.BatchedVSMPC(cfg, device=0, max_batch=B)
    st = m.phase_cycles(X).astype(np.int64)
    d = np.diff(st[:, :10], axis=1)
    tot = st[:, 9] - st[:, 0]
    print(f"== {wl} batch {B}: total cycles/instance median {np.median(tot):.0f} (s_memtime ticks), span of launch {(st[:,9].max()-st[:,0].min())}")
    for i, n in enumerate(names):
        print(f"  {n:38s} median {np.median(d[:, i]):9.0f}  max {d[:, i].max():9.0f}  share {100*np.median(d[:, i])/np.median(tot):5.1f}%")
    sub = ["P1 recursion / chain (wave 0)", "P1 barrier wait", "P1 MFMA / entries", "P1 set-up / contraction (structured form)", "start stamp (s_memrealtime)", "wall time, 10 ns ticks (s_memrealtime)"]
    if "VS_DIAG_P6" in os.environ.get("VSMPC_HIPCC_FLAGS", ""):
        sub[:4] = ["P6 input terms of all stages", "P6 set-up, joint expansion, first step", "P6 remaining steps", "P6 outputs to HBM"]
    elif "VS_DIAG_P3" in os.environ.get("VSMPC_HIPCC_FLAGS", ""):   # measurement build: P3 seen from wavefront 0
        sub[:4] = ["P3 panel streams (wave 0)", "P3 wait behind the stream", "P3 tile store + reload + trailing update", "P3 wait behind the update"]
    for i, n in enumerate(sub):
        print(f"    {n:36s} median {np.median(st[:, 10 + i]):9.0f}")
    start = st[:, 14] - st[:, 14].min(); end = start + st[:, 15]
    print(f"    start skew over the launch: median {np.median(start)/100:.1f} us, p90 {np.percentile(start,90)/100:.1f}, max {start.max()/100:.1f} us; last end - first start {end.max()/100:.1f} us")
    if B <= 256: print("    start offsets by workgroup id (us):", np.round(start[::16] / 100.0, 1))
    wall_us = np.median(st[:, 15]) / 100.0
    print(f"    shader clock while this instance ran: {np.median(tot) / wall_us / 1e3:.3f} GHz ({wall_us:.1f} us per instance)")
    m.close()
