"""Executable numpy model of the algorithm the HIP kernel implements (csrc/vsmpc_kernels.hip).

Test helper only: it documents the kernel's phases in plain array code and lets the CPU test
suite check the *algorithm* (condensing recursion, augmented Cholesky, Schur-complement box QP,
back-substitution, forward simulation) against the oracle without a GPU.  It is NOT a product
path and nothing outside tests/ imports it.

Phases (same names as the kernel):
  P0 linearise      A, Bj, Bt, c from the input record
  P1 condense       S_{k+1} = (I + dt_k A) S_k + dt_k E_k, per column in registers; Y_k = sqrt(Q) S_k[W]
                    C = sum_k Y_k^T Y_k over 128 padded columns (MFMA f64 16x16x4 on the GPU)
                    column order: [U_0..U_{H-1} | v_1..v_{nvb-1} | v_0 | affine | pad]
  P2 augment        M = C + R ; row `NZ` of M holds the condensed gradient
  P3 cholesky       first NZ pivots of the padded matrix; row NZ becomes (L^-1 g)^T
  P4 box QP on v    S = L22 L22^T, s = L22 (L^-1 g)_v ; block principal pivoting, in the primal form (masked
                    system on S) or, `dual=True`, in the dual form on P = S_NN^-1 = X^T X the kernel uses for few
                    active bounds; both produce the same iterate for a given active set
  P5 back-subst     U = L11^-T (y_U - L21^T v)
  P6 simulate       X_{k+1} = X_k + dt_k (A X_k + Bj U + Bt v + c)
"""
from __future__ import annotations

import numpy as np

W_ROWS = list(range(0, 12)) + list(range(20, 26))  # the 18 weighted state rows (costsVSMPC.cpp:78-93)


NJC = 6   # condensed joint variables per block after the joint reduction (kernel v24): rank of [Lambda_lin; Lambda_ang]


def column_maps(cfg, njc=8):
    """Internal input ordering [U | pad | v_1..v_{nvb-1} | v_0] -> (kind, block, component).  njc = 8: the joint increments
    themselves; njc = 6: the reduced joint variables y (joint_reduction), padded with dummy columns ("pad") to a multiple
    of 16 so that the throttle rows start on a tile boundary."""
    H, nvb = cfg.control_horizon, cfg.n_vblocks
    cols = [("U", j, q) for j in range(H) for q in range(njc)]
    cols += [("pad", 0, 0)] * (joint_rows(cfg, njc) - njc * H)
    cols += [("v", b, i) for b in range(1, nvb) for i in range(4)]
    cols += [("v", 0, i) for i in range(4)]
    return cols


def joint_rows(cfg, njc=8):
    n = njc * cfg.control_horizon
    return n if njc == 8 else ((n + 15) // 16) * 16


def joint_reduction(Lam6, wj, b):
    """Every joint block enters the dynamics only through Lam6 = [Lambda_lin; Lambda_ang] (6 x 8, the same for all blocks:
    the model is LTI over the horizon, constraintsVSMPC.cpp:85-103), and its cost is 1/2 U^T W U + w_reg q_err^T U with
    W = diag(weightDeltaJoint + weightRegularizationJointPos) (costsVSMPC.cpp:375-381,564-591).  With ut = W^(1/2) U and
    the Householder QR  (Lam6 W^(-1/2))^T = [Q N] [R; 0]  the change of variables ut = Q y + N n gives
        Lam6 U = R^T y,    1/2 U^T W U = 1/2 (|y|^2 + |n|^2),    b^T ut = (Q^T b)^T y + (N^T b)^T n,  b = w_reg W^(-1/2) q_err
    so n = -N^T b in closed form (the same for every block) and the condensed problem keeps 6 unknowns y per block with
    input matrix R^T, unit weights and gradient Q^T b.  No inverse is formed: a rank-deficient Lam6 (zero thrust) just
    leaves zero columns in R^T.  Returns (Beff = R^T 6 x 6, gy (6), nstar (2), V (6 reflectors, 8 each), beta (6))."""
    isw = 1.0 / np.sqrt(wj)
    Acol = (Lam6 * isw[None, :]).T.copy()          # 8 x 6
    bt = b * isw
    V = np.zeros((6, 8))
    beta = np.zeros(6)
    for k in range(6):
        x = Acol[k:, k]
        s = float(x @ x)
        if s > 1e-300:
            nrm = np.sqrt(s)
            alpha = -np.copysign(nrm, x[0])
            v = x.copy()
            v[0] -= alpha
            beta[k] = 1.0 / (nrm * (nrm + abs(x[0])))
            V[k, k:] = v
            Acol[k:, k:] -= beta[k] * np.outer(v, v @ Acol[k:, k:])
            bt[k:] -= beta[k] * v * (v @ bt[k:])
    R = np.triu(Acol[:6, :6])
    return R.T.copy(), bt[:6].copy(), -bt[6:].copy(), V, beta


def joint_expand(y, nstar, V, beta, wj):
    """U = W^(-1/2) H_1 ... H_6 [y; n]"""
    u = np.concatenate([y, nstar])
    for k in range(5, -1, -1):
        u -= beta[k] * V[k] * (V[k] @ u)
    return u / np.sqrt(wj)


def reduced_condensed(cfg, oracle, inp):
    """What the device's condensed problem must be, derived from the oracle's REFERENCE-ORDERED dense QP (null-space
    elimination of the states, no condensing recursion) and the joint reduction: returns (M, g, L) in the kernel's order
    [y_0..y_{H-1} | dummies | v_1..v_{nvb-1} | v_0] -- M = T^T Hr T, g = T^T (Hr u0 + gr) with U = T [y; v] + u0."""
    N, H, nvb = cfg.n_iter, cfg.control_horizon, cfg.n_vblocks
    nxs, NUo, NV = 26 * (N + 1), 8 * H, 4 * nvb
    Hd, gd, Ac, lo, hi = oracle.assemble_dense(cfg, inp)
    sol = np.linalg.solve(Ac[:nxs, :nxs], np.column_stack([lo[:nxs], Ac[:nxs, nxs:]]))
    Z = np.vstack([-sol[:, 1:], np.eye(NUo + NV)])
    xp = np.concatenate([sol[:, 0], np.zeros(NUo + NV)])
    Hr, gr = Z.T @ Hd @ Z, Z.T @ (Hd @ xp + gd)
    perm = list(range(NUo)) + list(range(NUo + 4, NUo + NV)) + list(range(NUo, NUo + 4))   # [U | v1.. | v0]
    Hr, gr = Hr[np.ix_(perm, perm)], gr[perm]
    _, red = reduced_model(cfg, oracle, inp)
    E = np.column_stack([joint_expand(np.eye(NJC)[i], np.zeros(2), red["V"], red["beta"], red["wj"]) for i in range(NJC)])  # W^-1/2 Q
    u_n = joint_expand(np.zeros(NJC), red["nstar"], red["V"], red["beta"], red["wj"])                                      # W^-1/2 N n
    NU = joint_rows(cfg, NJC)
    NZ = NU + NV
    T = np.zeros((NUo + NV, NZ))
    u0 = np.zeros(NUo + NV)
    for i in range(H):
        T[8 * i:8 * i + 8, NJC * i:NJC * i + NJC] = E
        u0[8 * i:8 * i + 8] = u_n
    T[NUo:, NU:] = np.eye(NV)
    M = T.T @ Hr @ T
    g = T.T @ (Hr @ u0 + gr)
    for d in range(NJC * H, NU):
        M[d, d] = 1.0
    L = np.linalg.cholesky(0.5 * (M + M.T))
    return M, g, L


def reduced_model(cfg, oracle, inp):
    """(A, Bj_eff 26 x 6, Bt, c) and the reduction's by-products for the record `inp`"""
    A, Bj, Bt, c = oracle.linearize(cfg, inp)
    wj = np.asarray(cfg.w_delta_joint, float) + cfg.w_reg_joint_pos
    Lam6 = np.vstack([Bj[3:6, :], Bj[9:12, :]])
    Beff, gy, nstar, V, beta = joint_reduction(Lam6, wj, cfg.w_reg_joint_pos * inp[oracle.IN_QERR:oracle.IN_QERR + 8])
    Bje = np.zeros((26, NJC))
    Bje[3:6] = Beff[0:3]
    Bje[9:12] = Beff[3:6]
    return (A, Bje, Bt, c), dict(gy=gy, nstar=nstar, V=V, beta=beta, wj=wj)


def condense_syrk(cfg, oracle, inp, lin=None, njc=8):
    """P1 as the sensitivity recursion + SYRK (kernels up to v12): the padded C = sum_k Y_k^T Y_k, whose row / column
    NZ (the affine column) holds the condensed gradient of the tracking cost.  `lin` = (A, Bj, Bt, c) with njc joint
    columns per block replaces the record's own linearisation (the reduced model)."""
    N, nS, H = cfg.n_iter, cfg.n_iter_small, cfg.control_horizon
    NU, NV = joint_rows(cfg, njc), 4 * cfg.n_vblocks
    NZ = NU + NV
    NP = ((NZ + 1 + 15) // 16) * 16
    A, Bj, Bt, c = oracle.linearize(cfg, inp) if lin is None else lin
    dts = oracle.dt_schedule(cfg)
    qd = oracle.state_weight(cfg)
    sq = np.sqrt(qd[W_ROWS])
    cols = column_maps(cfg, njc)
    xref_win = inp[oracle.IN_XREF:oracle.IN_XREF + 12 * cfg.n_ref_cols].reshape(cfg.n_ref_cols, 12)
    S = np.zeros((26, NP))
    S[:, NZ] = inp[0:26]  # affine column starts at x0
    C = np.zeros((NP, NP))
    for k in range(N):  # stage k -> node k+1
        dt = dts[k]
        jb = oracle.joint_block_of_stage(cfg, k)
        tb = oracle.throttle_block_of_stage(cfg, k)
        E = np.zeros((26, NP))
        for ci, (kind, blk, comp) in enumerate(cols):
            if kind == "U" and blk == jb:
                E[:, ci] = Bj[:, comp]
            if kind == "v" and blk == tb:
                E[:, ci] = Bt[:, comp]
        E[:, NZ] = c
        S = S + dt * (A @ S + E)
        col = 0 if k < nS else k - nS
        xr = np.zeros(26)
        xr[0:12] = xref_win[col]
        Sa = S.copy()
        Sa[:, NZ] -= xr
        Y = sq[:, None] * Sa[W_ROWS, :]
        C += Y.T @ Y

    return C


def solve_model(cfg, oracle, inp, max_bpp_iter=60, dual=False, trace=None, condense=None, reduce=False, out=None):
    """`oracle` is the oracle module (for linearize/dt_schedule helpers that restate the reference);
    everything downstream of (A, Bj, Bt, c) is the kernel's own algorithm.  reduce=True: with the joint reduction in front
    (6 unknowns per joint block, kernel v24); `out` (a dict) receives the condensed matrices M and L."""
    N, nS, H = cfg.n_iter, cfg.n_iter_small, cfg.control_horizon
    nvb = cfg.n_vblocks
    njc = NJC if reduce else 8
    NU, NV = joint_rows(cfg, njc), 4 * nvb
    NZ = NU + NV
    NP = ((NZ + 1 + 15) // 16) * 16
    red = None
    if reduce:
        (A, Bj, Bt, c), red = reduced_model(cfg, oracle, inp)
    else:
        A, Bj, Bt, c = oracle.linearize(cfg, inp)
    dts = oracle.dt_schedule(cfg)
    qd = oracle.state_weight(cfg)
    sq = np.sqrt(qd[W_ROWS])
    cols = column_maps(cfg, njc)
    vmin, vmax = oracle.throttle_bounds(cfg)
    vprev = np.array([oracle.v_of_throttle(inp[oracle.IN_UPREV + i]) for i in range(4)])
    hold = inp[oracle.IN_HOLD] != 0.0
    xref_win = inp[oracle.IN_XREF:oracle.IN_XREF + 12 * cfg.n_ref_cols].reshape(cfg.n_ref_cols, 12)

    # ---- P1 condense
    lin = (A, Bj, Bt, c) if reduce else None
    C = condense_syrk(cfg, oracle, inp, lin, njc) if condense is None else condense(cfg, oracle, inp, lin, njc)

    # ---- P2 augment with R and the input-cost gradient
    M = C.copy()
    gz = np.zeros(NZ)
    wj = np.asarray(cfg.w_delta_joint) + cfg.w_reg_joint_pos
    for ci, (kind, blk, comp) in enumerate(cols):
        if kind == "U" and not reduce:
            M[ci, ci] += wj[comp]
            gz[ci] = cfg.w_reg_joint_pos * inp[oracle.IN_QERR + comp]
        elif kind == "U":
            M[ci, ci] += 1.0                 # |y|^2 / 2: unit weights in the reduced variables
            gz[ci] = red["gy"][comp]
        elif kind == "pad":
            M[ci, ci] += 1.0                 # dummy unknowns: decoupled, solution 0
    vidx = {(blk, comp): ci for ci, (kind, blk, comp) in enumerate(cols) if kind == "v"}
    for i in range(4):
        for b in range(nvb - 1):
            a_, b_ = vidx[(b, i)], vidx[(b + 1, i)]
            M[a_, a_] += cfg.w_throttle
            M[b_, b_] += cfg.w_throttle
            M[a_, b_] -= cfg.w_throttle
            M[b_, a_] -= cfg.w_throttle
        M[vidx[(0, i)], vidx[(0, i)]] += cfg.w_initial_throttle
        gz[vidx[(0, i)]] += -cfg.w_initial_throttle * vprev[i]
    M[NZ, :NZ] += gz
    M[:NZ, NZ] += gz

    # ---- P3 Cholesky of the leading NZ x NZ block; rows >= NZ are carried along (TRSM only)
    Lf = np.tril(M)
    for j in range(NZ):
        d = Lf[j, j]
        if not d > 0:
            return None, 3, 0
        inv = 1.0 / np.sqrt(d)
        Lf[j:, j] *= inv
        Lf[j, j] = d * inv
        for k2 in range(j + 1, NZ + 1):
            Lf[k2:, k2] -= Lf[k2:, j] * Lf[k2, j]
    if out is not None:
        out["M"], out["L"] = M, Lf
    Lm = Lf[:NZ, :NZ]
    ghat = Lf[NZ, :NZ]          # = L^-1 g
    y = -ghat

    # ---- P4 box QP on the throttle block via the Schur complement
    L22 = Lm[NU:, NU:]
    Sv = L22 @ L22.T
    sv = L22 @ ghat[NU:]
    lo = np.full(NV, vmin)
    hi = np.full(NV, vmax)
    fixed = np.zeros(NV, dtype=bool)
    if hold:
        lo[NV - 4:] = vprev
        hi[NV - 4:] = vprev
        fixed[NV - 4:] = True
    state = np.zeros(NV, dtype=int)
    state[fixed] = -1
    v = np.zeros(NV)
    best, patience, status, iters = NV + 1, 3, 2, 0
    # dual form (what the kernel runs for few active bounds): N = throttles not pinned by the hold, X = L_N^-1 with L_N
    # the leading block of L22 (= the factor of S_NN), P = S_NN^-1 = X^T X, v_u = the solve with only the pins enforced.
    # Fixing A at its bounds b_A:  mu = P_AA^-1 (v_u,A - b_A),  v_N = v_u,N - P[:,A] mu,  gradient_A = -mu.
    n_free = NV - 4 if hold else NV
    Xn = np.linalg.inv(L22[:n_free, :n_free])
    P = Xn.T @ Xn
    pinned = np.where(fixed, lo, 0.0)
    vu = np.linalg.solve(Sv[:n_free, :n_free], -(sv[:n_free] + Sv[:n_free, n_free:] @ pinned[n_free:]))
    for it in range(max_bpp_iter):
        iters = it + 1
        F = state == 0
        v = np.where(state == -1, lo, np.where(state == 1, hi, 0.0))
        if dual:
            A_ = np.nonzero((state != 0) & ~fixed)[0]
            vN = vu.copy()
            grad = np.zeros(NV)
            if A_.size:
                mu = np.linalg.solve(P[np.ix_(A_, A_)], vu[A_] - v[A_])
                vN = vu - P[:, A_] @ mu
                vN[A_] = v[A_]                      # bound variables sit exactly on their bound
                grad[A_] = -mu
            v = np.concatenate([vN, v[n_free:]])
        else:
            # masked system: identity on bound rows (what the single-wave kernel factorises)
            Sm = np.where(np.outer(F, F), Sv, 0.0) + np.diag((~F).astype(float))
            rhs = np.where(F, -(sv + Sv @ np.where(F, 0.0, v)), v)
            v = np.linalg.solve(Sm, rhs)
            grad = Sv @ v + sv
        if trace is not None:
            trace.append((state.copy(), v.copy(), np.where((state != 0) & ~fixed, grad, 0.0)))
        tolv = 1e-12 * (1.0 + np.abs(v))
        gtol = 1e-10 * (1.0 + np.abs(sv).max())
        vlo = F & (v < lo - tolv)
        vhi = F & (v > hi + tolv)
        rlo = (state == -1) & ~fixed & (grad < -gtol)
        rhi = (state == 1) & ~fixed & (grad > gtol)
        inf = vlo | vhi | rlo | rhi
        ninf = int(inf.sum())
        if ninf == 0:
            status = 1
            break
        if ninf < best:
            best, patience = ninf, 3
            pick = inf
        elif patience > 0:
            patience -= 1
            pick = inf
        else:
            pick = np.zeros(NV, dtype=bool)
            pick[np.nonzero(inf)[0].max()] = True
        state[pick & vlo] = -1
        state[pick & vhi] = 1
        state[pick & (rlo | rhi)] = 0

    # ---- P5 back-substitution for the joints
    L11, L21 = Lm[:NU, :NU], Lm[NU:, :NU]
    U = np.linalg.solve(L11.T, y[:NU] - L21.T @ v)
    z = np.concatenate([U, v])

    # ---- P6 forward simulation, output in the reference's variable order
    x = np.zeros(cfg.n_var)
    X = inp[0:26].copy()
    x[0:26] = X
    vref = np.zeros(NV)  # reference order v_0..v_{nvb-1}
    vref[0:4] = v[NV - 4:]
    vref[4:] = v[:NV - 4]
    for k in range(N):
        jb = oracle.joint_block_of_stage(cfg, k)
        tb = oracle.throttle_block_of_stage(cfg, k)
        X = X + dts[k] * (A @ X + Bj @ U[njc * jb:njc * jb + njc] + Bt @ vref[4 * tb:4 * tb + 4] + c)
        x[26 * (k + 1):26 * (k + 2)] = X
    if reduce:   # back to joint increments: U_i = W^(-1/2) (Q y_i + N n)
        U = np.concatenate([joint_expand(U[njc * i:njc * i + njc], red["nstar"], red["V"], red["beta"], red["wj"]) for i in range(H)])
    x[cfg.off_joints:cfg.off_joints + 8 * H] = U
    x[cfg.off_throttle:cfg.off_throttle + NV] = vref
    return x, status, iters
