"""Executable numpy model of the structure-exploiting condensing (phase P1 of csrc/vsmpc_kernels.hip, kernel v13+).

Test helper only (nothing outside tests/ imports it).  It documents, in plain array code, how the kernel forms the
condensed Hessian C = sum_k Y_k^T Y_k and the condensed gradient WITHOUT the sensitivity matrix:

The reference's model (systemDynamicsVSMPC.cpp:79-103,288-319,384-429) is a cascade
    throttles v -> jets (T, Tdot) -> momenta h -> CoM / RPY x -> error integrators e
whose linear half (p, h_lin, e_pos) and angular half (rpy, h_ang, e_rpy) do not talk to each other, and every input
enters a half only as a 3-vector forcing of its momentum rows:
    phi_i = Lambda_half U_{jb(i)} + A_mom,half (T_i - Tbar_i)              (stage i)
so, per half, with xi = (x, h, e) in R^9, Abar_m = I + dt_m K, K xi = (M1 h, Sk h, x):

    P_N = Q,  P_m = Q + Abar_m^T P_{m+1} Abar_m                             cost-to-go of the half (backward, 9x9)
    G(i, j) = E_h^T Phi(j+1, i+1)^T P_{j+1} E_h,  i <= j                   3x3: impulse on h at stage i seen from stage j
    H(i, j) = dt_i dt_j G(i, j),  H(j, i) = H(i, j)^T
    lambda_m = Q (xibar_m - ref_m) + Abar_m^T lambda_{m+1}                  adjoint of the nominal (zero-input) trajectory
    gamma_i = dt_i E_h^T lambda_{i+1}

and with the profile pi_c(i) in R^3 of condensed column c (joint (b, q): Lambda[:, q] while jb(i) == b; throttle (m, q):
A_mom[:, q] tau_i^{(m,q)}, tau = the jet's thrust sensitivity trajectory):

    C[r, c] = sum_half sum_{i, i'} pi_r(i)^T H(i, i') pi_c(i'),   g[c] = sum_half sum_i pi_c(i)^T gamma_i.

This is O(N^2) small 3x3 work instead of the O(N^3)-ish SYRK over the 18 N weighted sensitivity rows
(constraintsVSMPC.cpp:76-131 is what is condensed; costsVSMPC.cpp:166-200 the weights and the reference column map).
"""
from __future__ import annotations

import numpy as np

from algo_model import column_maps, joint_rows

HALVES = (
    # x rows, h rows, e rows
    (slice(0, 3), slice(3, 6), slice(20, 23)),
    (slice(6, 9), slice(9, 12), slice(23, 26)),
)


def jet_trajectories(cfg, oracle, A, Bt, c, inp):
    """tau[col][i]: thrust of the column's jet at stage i per unit of the throttle column (NV rows), and the nominal
    thrust trajectories Tbar[q][i] (v = 0).  Mirrors P1a of the kernel (explicit Euler, T_i is what stage i sees)."""
    N = cfg.n_iter
    dts = oracle.dt_schedule(cfg)
    cols = column_maps(cfg)
    NU = 8 * cfg.control_horizon    # (throttle columns are looked up in the un-reduced column map)
    NV = 4 * cfg.n_vblocks
    tau = np.zeros((NV, N))
    Tbar = np.zeros((4, N))
    for q in range(4):
        jon, ja, jb = A[12 + q, 16 + q], A[16 + q, 12 + q], A[16 + q, 16 + q]
        b12, b16 = Bt[12 + q, q], Bt[16 + q, q]
        T, Td = inp[12 + q], inp[16 + q]
        for k in range(N):
            Tbar[q, k] = T
            dT = jon * Td + c[12 + q]
            dTd = ja * T + jb * Td + c[16 + q]
            T, Td = T + dts[k] * dT, Td + dts[k] * dTd
        for ci in range(NV):
            kind, blk, comp = cols[NU + ci]
            if comp != q:
                continue
            T, Td = 0.0, 0.0
            for k in range(N):
                tau[ci, k] = T
                on = 1.0 if oracle.throttle_block_of_stage(cfg, k) == blk else 0.0
                dT = jon * Td + on * b12
                dTd = ja * T + jb * Td + on * b16
                T, Td = T + dts[k] * dT, Td + dts[k] * dTd
    return tau, Tbar


def half_operators(A, half):
    xs, hs, es = HALVES[half]
    M1 = A[xs, hs]
    Sk = A[hs, hs]
    K = np.zeros((9, 9))
    K[0:3, 3:6] = M1
    K[3:6, 3:6] = Sk
    K[6:9, 0:3] = A[es, xs]   # identity
    return K


def condense_structured(cfg, oracle, inp, lin=None, njc=8):
    """Returns the padded condensed matrix (NP x NP): C in [0:NZ, 0:NZ], the condensed gradient of the tracking cost in
    row / column NZ, the constant term at [NZ, NZ] -- i.e. what sum_k Y_k^T Y_k gives in algo_model.solve_model.
    `lin` / `njc`: the reduced model of algo_model.reduced_model (6 joint columns per block, input matrix R^T)."""
    N, nS, H = cfg.n_iter, cfg.n_iter_small, cfg.control_horizon
    NU, NV = joint_rows(cfg, njc), 4 * cfg.n_vblocks
    NZ = NU + NV
    NP = ((NZ + 1 + 15) // 16) * 16
    A, Bj, Bt, c = oracle.linearize(cfg, inp) if lin is None else lin
    dts = oracle.dt_schedule(cfg)
    qd = oracle.state_weight(cfg)
    cols = column_maps(cfg, njc)
    xref_win = inp[oracle.IN_XREF:oracle.IN_XREF + 12 * cfg.n_ref_cols].reshape(cfg.n_ref_cols, 12)
    tau, Tbar = jet_trajectories(cfg, oracle, A, Bt, c, inp)

    C = np.zeros((NP, NP))
    for half in range(2):
        xs, hs, es = HALVES[half]
        K = half_operators(A, half)
        Q = np.diag(np.concatenate([qd[xs], qd[hs], qd[es]]))
        Lam = Bj[hs, :]            # 3 x 8 (3 x 6 in the reduced model)
        Am = A[hs, 12:16]          # 3 x 4
        ch, ce = c[hs], c[es]
        Abar = [np.eye(9) + dts[m] * K for m in range(N)]
        # nominal trajectory (U = 0, v = 0) and its weighted residual
        xi = np.concatenate([inp[xs], inp[hs], inp[es]])
        res = np.zeros((N + 1, 9))
        for k in range(N):
            f = np.concatenate([np.zeros(3), Am @ Tbar[:, k] + ch, ce])
            xi = Abar[k] @ xi + dts[k] * f
            col = 0 if k < nS else k - nS
            ref = np.zeros(9)
            ref[0:3] = xref_win[col][xs if half == 0 else slice(6, 9)]
            ref[3:6] = xref_win[col][hs if half == 0 else slice(9, 12)]
            res[k + 1] = xi - ref
        # backward: cost-to-go P_m, adjoint lambda_m (m = N .. 1)
        P = [None] * (N + 2)
        lam = np.zeros((N + 2, 9))
        P[N] = Q.copy()
        lam[N] = Q @ res[N]
        for m in range(N - 1, 0, -1):
            P[m] = Q + Abar[m].T @ P[m + 1] @ Abar[m]
            lam[m] = Q @ res[m] + Abar[m].T @ lam[m + 1]
        gamma = np.array([dts[i] * lam[i + 1][3:6] for i in range(N)])
        # H(i, j) = dt_i dt_j E_h^T Phi(j+1, i+1)^T P_{j+1} E_h
        Hm = np.zeros((N, N, 3, 3))
        for j in range(N):
            Z = P[j + 1][:, 3:6].copy()
            for i in range(j, -1, -1):
                Hm[i, j] = dts[i] * dts[j] * Z[3:6, :]
                Hm[j, i] = Hm[i, j].T
                if i > 0:
                    Z = Abar[i].T @ Z
        # profiles pi_c(i)
        prof = np.zeros((NZ, N, 3))
        for ci, (kind, blk, comp) in enumerate(cols):
            for i in range(N):
                if kind == "U":
                    if oracle.joint_block_of_stage(cfg, i) == blk:
                        prof[ci, i] = Lam[:, comp]
                elif kind == "v":
                    prof[ci, i] = Am[:, comp] * tau[ci - NU, i]
        W = np.einsum("ijab,cjb->cia", Hm, prof)          # W_c(i) = sum_i' H(i, i') pi_c(i')
        C[:NZ, :NZ] += np.einsum("ria,cia->rc", prof, W)
        g = np.einsum("cia,ia->c", prof, gamma)
        C[NZ, :NZ] += g
        C[:NZ, NZ] += g
        C[NZ, NZ] += sum(res[k] @ Q @ res[k] for k in range(1, N + 1))
    return C
