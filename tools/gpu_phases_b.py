"""GPU box: phase cycles for several batch sizes (contention vs clock)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); solver = importlib.import_module(PKG + ".solver")
cfg = pkg.horizon2x_config() if (len(sys.argv) > 1 and sys.argv[1] == "h2x") else pkg.paper_config()
X = synth.make_batch(cfg, 256, workload="hover")
for B in (1, 8, 64, 128, 256, 512):
    Xb = np.tile(X, (max(1, B // 256), 1))[:B] if B > 256 else X[:B]
    m = solver.BatchedVSMPC(cfg, device=0, max_batch=B)
    st = m.phase_cycles(Xb).astype(np.int64)
    d = np.diff(st[:, :10], axis=1)
    print(f"batch {B:4d}: total {np.median(st[:,9]-st[:,0]):8.0f}  P0 {np.median(d[:,0]):6.0f} P1 {np.median(d[:,1]):6.0f} P3 {np.median(d[:,3]):6.0f} P5 {np.median(d[:,5]+d[:,6]):6.0f} P6 {np.median(d[:,7]):6.0f}")
    m.close()
