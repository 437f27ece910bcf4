"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

numpy restatement of what the reference's Robot::setState caches (utils/src/Robot.cpp:198-335, getJacobian :505-514)
for a SIMPLIFIED kinematic tree: a floating base, the 8 controlled revolute joints (two 4-joint arm chains) and 4 jet
frames.  The reference computes these quantities with iDynTree 14.0.2 on the iRonCub URDF (ironcub-models 0.0.2); neither
is in this image, so nothing here can be checked against the reference's numbers: the restatement follows iDynTree's
published conventions (MIXED velocity representation: base linear velocity = velocity of the base origin in world axes,
angular velocity in world axes; free-floating Jacobians 6 x (6 + n), [linear; angular]) and is pinned by its own
consistency checks in tests/test_robot_tree_oracle.py (finite differences of the forward kinematics, kinetic energy
against the mass-matrix block, momentum against the CoM Jacobian).  The tree's numbers are synthetic ("iRonCub-like"),
committed in <package>/robot_tree.py.

Only tests/, smoke() and bench.py's cpu_baseline leg may import anything under oracle/.
"""
from __future__ import annotations

import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def rodrigues(axis, angle):
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    K = skew(a)
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


def sym6(v):
    xx, xy, xz, yy, yz, zz = v
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


def forward(tree: dict, state: dict) -> dict:
    """tree: the arrays of <package>/robot_tree.py (vsmpc_tree); state: p_base(3), R_base(3x3), v_base(3), w_base(3)
    (world), q(8), qd(8), thrust(4).  Returns what Robot::setState caches, n = 23 robot joints."""
    NB, NJ, n = len(tree["mass"]), len(tree["joint_axis"]), 23
    pb, Rb = np.asarray(state["p_base"], float), np.asarray(state["R_base"], float).reshape(3, 3)
    vb, wb = np.asarray(state["v_base"], float), np.asarray(state["w_base"], float)
    q, qd = np.asarray(state["q"], float), np.asarray(state["qd"], float)
    R = [None] * NB
    p = [None] * NB
    R[0], p[0] = Rb, pb
    axis_w, org_w = [None] * NJ, [None] * NJ
    up = [[] for _ in range(NB)]                    # joints between the base and body b
    for j in range(NJ):                             # joint j moves body j + 1; parents precede children
        b, par = j + 1, tree["parent"][j + 1]
        org_w[j] = p[par] + R[par] @ np.asarray(tree["joint_origin"][j], float)
        axis_w[j] = R[par] @ (np.asarray(tree["joint_axis"][j], float) / np.linalg.norm(tree["joint_axis"][j]))
        R[b] = R[par] @ rodrigues(tree["joint_axis"][j], q[j])
        p[b] = org_w[j]
        up[b] = up[par] + [j]
    mass = np.asarray(tree["mass"], float)
    m = mass.sum()
    cw = [p[b] + R[b] @ np.asarray(tree["com"][b], float) for b in range(NB)]
    Iw = [R[b] @ sym6(tree["inertia"][b]) @ R[b].T for b in range(NB)]
    com = sum(mass[b] * cw[b] for b in range(NB)) / m
    # body velocities
    w = [wb + sum((qd[j] * axis_w[j] for j in up[b]), np.zeros(3)) for b in range(NB)]
    vc = [vb + np.cross(wb, cw[b] - pb) + sum((qd[j] * np.cross(axis_w[j], cw[b] - org_w[j]) for j in up[b]), np.zeros(3))
          for b in range(NB)]
    h_lin = sum(mass[b] * vc[b] for b in range(NB))
    h_ang = sum(Iw[b] @ w[b] + mass[b] * np.cross(cw[b] - com, vc[b]) for b in range(NB))
    momentum = np.concatenate([h_lin, h_ang])                                   # centroidal, world axes
    momentum_body = np.concatenate([Rb.T @ h_lin, Rb.T @ h_ang])               # Robot.cpp:324-326
    # base block of the free-floating mass matrix (MIXED): [[m I, -m S(c)], [m S(c), I_O]], c = com - p_base
    c = com - pb
    IO = sum(Iw[b] + mass[b] * (np.dot(cw[b] - pb, cw[b] - pb) * np.eye(3) - np.outer(cw[b] - pb, cw[b] - pb)) for b in range(NB))
    Mb = np.zeros((6, 6))
    Mb[0:3, 0:3] = m * np.eye(3)
    Mb[0:3, 3:6] = -m * skew(c)
    Mb[3:6, 0:3] = m * skew(c)
    Mb[3:6, 3:6] = IO
    col = tree["robot_joint"]

    def frame_jacobian(point, chain):
        J = np.zeros((6, 6 + n))
        J[0:3, 0:3] = np.eye(3)
        J[0:3, 3:6] = -skew(point - pb)
        J[3:6, 3:6] = np.eye(3)
        for j in chain:
            J[0:3, 6 + col[j]] = np.cross(axis_w[j], point - org_w[j])
            J[3:6, 6 + col[j]] = axis_w[j]
        return J

    Jcom = np.zeros((3, 6 + n))
    for b in range(NB):
        Jcom += mass[b] / m * frame_jacobian(cw[b], up[b])[0:3]
    jets = []
    Amom = np.zeros((6, 4))
    axes, arms, Jjet, Jrel = [], [], [], []
    for i in range(4):
        b = tree["jet_body"][i]
        pj = p[b] + R[b] @ np.asarray(tree["jet_origin"][i], float)
        a = R[b] @ (np.asarray(tree["jet_axis"][i], float) / np.linalg.norm(tree["jet_axis"][i]))   # Robot.cpp:256
        r = pj - com                                                                                # :258 (zero CoM offset)
        J = frame_jacobian(pj, up[b])
        Jr = np.vstack([Rb.T @ J[0:3, 6:], Rb.T @ J[3:6, 6:]])                  # getRelativeJacobian(base, jet), base axes
        axes.append(a); arms.append(r); Jjet.append(J); Jrel.append(Jr)
        Amom[0:3, i] = a                                                        # :261
        Amom[3:6, i] = np.cross(r, a)                                           # :263-264
        jets.append(pj)
    Amom_body = np.vstack([Rb.T @ Amom[0:3], Rb.T @ Amom[3:6]])                 # :327-328
    return dict(mass=m, com=com, momentum=momentum, momentum_body=momentum_body, Mb=Mb, Jcom=Jcom, Jjet=Jjet, Jrel=Jrel,
                axes=np.array(axes), arms=np.array(arms), Amom=Amom, Amom_body=Amom_body, jet_pos=np.array(jets),
                body_com=np.array(cw), body_w=np.array(w), body_vc=np.array(vc), body_Iw=np.array(Iw), R=R, p=p)


def kin_record(out: dict, state: dict, layout) -> np.ndarray:
    """the VSMPC_KIN_* record vsmpc_kinematics_batch reads (include/vsmpc.h), from forward()'s result"""
    k = np.zeros(layout.KIN_SIZE)
    Rb = np.asarray(state["R_base"], float).reshape(3, 3)
    k[layout.KIN_WRB:layout.KIN_WRB + 9] = Rb.reshape(-1)
    k[layout.KIN_THRUST:layout.KIN_THRUST + 4] = state["thrust"]
    k[layout.KIN_AXES:layout.KIN_AXES + 12] = out["axes"].reshape(-1)
    k[layout.KIN_ARMS:layout.KIN_ARMS + 12] = out["arms"].reshape(-1)
    k[layout.KIN_JREL:layout.KIN_JREL + 276] = np.stack([j[3:6, :] for j in out["Jrel"]]).reshape(-1)
    k[layout.KIN_JFRAME:layout.KIN_JFRAME + 276] = np.stack([j[0:3, 6:29] for j in out["Jjet"]]).reshape(-1)
    k[layout.KIN_JCOM:layout.KIN_JCOM + 69] = out["Jcom"][:, 6:29].reshape(-1)
    k[layout.KIN_MB:layout.KIN_MB + 36] = out["Mb"].reshape(-1)
    k[layout.KIN_R:layout.KIN_R + 3] = out["com"] - np.asarray(state["p_base"], float)
    return k
