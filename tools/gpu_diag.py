"""Development diagnostic (GPU box): compares every stage of the HIP path with the oracle / algorithm model.
Writes gpurun_out/diag.log.  Not part of the product path."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); solver = importlib.import_module(PKG + ".solver")
import vsmpc_ref as R, algo_model as M

def rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))

def main():
    cfg = pkg.paper_config(); rcfg = R.paper_config()
    B = 16
    X = np.concatenate([synth.make_batch(cfg, B // 2, workload="hover"), synth.make_batch(cfg, B // 2, workload="takeoff")])
    mpc = solver.BatchedVSMPC(cfg, device=0, max_batch=B)
    print("kernel", mpc.kernel_name, "n_p", mpc.n_p)
    A, Bj, Bt, c, dt = mpc.linearize(X)
    wl = 0
    for b in range(B):
        Ar, Bjr, Btr, cr = R.linearize(rcfg, X[b])
        wl = max(wl, rel(A[b], Ar), rel(Bj[b], Bjr), rel(Bt[b], Btr), rel(c[b], cr))
    print("linearize max rel err", wl, "dt err", np.abs(dt - R.dt_schedule(rcfg)).max())
    # condensed matrices of instance 0 and 9
    for b in (0, 9):
        Mg, Lg = mpc.debug_condensed(X[b])
        # model: recompute M and L via algo_model internals
        xm, st, it = M.solve_model(rcfg, R, X[b])
        H, g, Ac, lo, hi = R.assemble_dense(rcfg, X[b])
        # condensed H from oracle via null-space (internal order)
        nxs = 26 * 18
        Ax, Az = Ac[:nxs, :nxs], Ac[:nxs, nxs:]
        sol = np.linalg.solve(Ax, np.column_stack([lo[:nxs], Az]))
        Xb, G = sol[:, 0], sol[:, 1:]
        Z = np.vstack([-G, np.eye(120)])
        xp = np.concatenate([Xb, np.zeros(120)])
        Hr = Z.T @ H @ Z; gr = Z.T @ (H @ xp + g)
        perm = list(range(96)) + list(range(100, 120)) + list(range(96, 100))
        Hr = Hr[np.ix_(perm, perm)]; gr = gr[perm]
        Mh = np.tril(Mg[:120, :120]); Mh = Mh + np.tril(Mh, -1).T
        print(f"inst {b}: condensed H rel err {rel(Mh, Hr):.3e}  gradient rel err {rel(Mg[120, :120], gr):.3e}")
        Lr = np.linalg.cholesky(Hr)
        print(f"inst {b}: L rel err {rel(np.tril(Lg[:120, :120]), Lr):.3e}  ghat rel err {rel(Lg[120, :120], np.linalg.solve(Lr, gr)):.3e}")
    x, fm, st, it = mpc.solve(X)
    print("status", st, "iters", it)
    worst = 0
    for b in range(B):
        xr, y, itr, _ = R.solve_instance(rcfg, X[b])
        e = rel(x[b], xr); worst = max(worst, e)
        fr = R.first_move_vector(rcfg, xr)
        print(f"inst {b}: rel err {e:.3e} first-move err {rel(fm[b], fr):.3e} oracle iters {itr} gpu iters {it[b]}")
    print("WORST", worst)
    # timing: device-resident loop
    for Bt_ in (256, 4096):
        Xb = synth.make_batch(cfg, 256, workload="hover"); Xb = np.tile(Xb, (Bt_ // 256, 1))
        m2 = solver.BatchedVSMPC(cfg, device=0, max_batch=Bt_)
        dev = torch.device("cuda:0")
        d_in = torch.from_numpy(Xb).to(dev); d_x = torch.empty((Bt_, cfg.n_var), dtype=torch.float64, device=dev)
        d_fm = torch.empty((Bt_, 24), dtype=torch.float64, device=dev); d_st = torch.empty(Bt_, dtype=torch.int32, device=dev); d_it = torch.empty(Bt_, dtype=torch.int32, device=dev)
        for _ in range(5): m2.solve_device(d_in, d_x, d_fm, d_st, d_it)
        torch.cuda.synchronize(); t = time.time(); K = 50
        s = torch.cuda.current_stream()
        m2.timing_begin(s)
        for _ in range(K): m2.solve_device(d_in, d_x, d_fm, d_st, d_it)
        ms = m2.timing_end(s, K)
        torch.cuda.synchronize(); wall = (time.time() - t) / K
        print(f"batch {Bt_}: {ms*1e3:.1f} us/launch (events), wall {wall*1e6:.1f} us -> {Bt_/(ms*1e-3):.3e} solves/s; status ok {(d_st.cpu().numpy()==1).all()}")

if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
