"""The simplified kinematic tree of the batched kinematics provider (include/vsmpc.h `vsmpc_tree`).

The reference's Robot wraps iDynTree on the iRonCub URDF (utils/src/Robot.cpp:198-335; ironcub-models 0.0.2, not in this
image).  What is committed here is a SYNTHETIC "iRonCub-like" tree as plain arrays -- a floating base (torso, head and
legs lumped), the 8 controlled joints as two 4-joint arm chains (shoulder pitch / roll / yaw, elbow: the order of
`controlledJoints`, src/config/vs_mcp_config.xml:17) at robot joint indices 3..10, and 4 jet frames (two on the
forearms, two on the back of the base).  Masses, offsets and inertias are plausible round numbers, NOT the URDF's.
"""
from __future__ import annotations

import ctypes

import numpy as np

NB, NJ, NJETS = 9, 8, 4
RS_P, RS_R, RS_V, RS_W, RS_Q, RS_QD, RS_T, RS_SIZE = 0, 3, 12, 15, 18, 26, 34, 38       # VSMPC_RS_* (state record)
RO_COM, RO_MOM, RO_MOMB, RO_MASS, RO_AMOM, RO_AMOMB, RO_RPY, RO_SIZE = 0, 3, 9, 15, 16, 40, 64, 67   # VSMPC_RO_* (outputs)


class CTree(ctypes.Structure):
    """ctypes image of `vsmpc_tree` (include/vsmpc.h)."""
    _fields_ = [
        ("parent", ctypes.c_int * NB),
        ("robot_joint", ctypes.c_int * NJ),
        ("joint_axis", ctypes.c_double * (3 * NJ)),
        ("joint_origin", ctypes.c_double * (3 * NJ)),
        ("mass", ctypes.c_double * NB),
        ("com", ctypes.c_double * (3 * NB)),
        ("inertia", ctypes.c_double * (6 * NB)),
        ("jet_body", ctypes.c_int * NJETS),
        ("jet_origin", ctypes.c_double * (3 * NJETS)),
        ("jet_axis", ctypes.c_double * (3 * NJETS)),
        ("gravity", ctypes.c_double * 3),
    ]


def default_tree() -> dict:
    def arm(side):     # side = +1 left, -1 right
        return dict(
            axis=[[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 1.0, 0.0]],      # pitch, roll, yaw, elbow
            origin=[[0.0, 0.11 * side, 0.25], [0.0, 0.03 * side, 0.0], [0.0, 0.0, -0.05], [0.015, 0.0, -0.15]],
            mass=[0.9, 0.7, 1.4, 1.9],                                                      # forearm carries the arm jet
            com=[[0.0, 0.01 * side, 0.0], [0.0, 0.0, -0.02], [0.0, 0.0, -0.08], [0.0, 0.0, -0.09]],
            inertia=[[1.2e-3, 0, 0, 1.0e-3, 0, 1.1e-3], [9e-4, 0, 0, 9e-4, 0, 6e-4], [6.5e-3, 0, 1e-4, 6.4e-3, 0, 1.1e-3],
                     [9.5e-3, 0, 0, 9.3e-3, 2e-4, 1.6e-3]])
    L, R = arm(+1.0), arm(-1.0)
    return dict(
        parent=[-1, 0, 1, 2, 3, 0, 5, 6, 7],
        robot_joint=[3, 4, 5, 6, 7, 8, 9, 10],
        joint_axis=L["axis"] + R["axis"],
        joint_origin=L["origin"] + R["origin"],
        mass=[60.5] + L["mass"] + R["mass"],
        com=[[-0.01, 0.0, -0.12]] + L["com"] + R["com"],
        inertia=[[6.8, 0.0, 0.05, 6.1, 0.0, 1.4]] + L["inertia"] + R["inertia"],
        jet_body=[4, 8, 0, 0],
        jet_origin=[[0.0, 0.02, -0.16], [0.0, -0.02, -0.16], [-0.13, 0.09, 0.16], [-0.13, -0.09, 0.16]],
        jet_axis=[[0.0, 0.17, 0.985], [0.0, -0.17, 0.985], [-0.1, 0.0, 0.995], [-0.1, 0.0, 0.995]],   # thrust force directions
        gravity=[0.0, 0.0, -9.81],
    )


def to_c(tree: dict) -> CTree:
    c = CTree()
    c.parent = (ctypes.c_int * NB)(*tree["parent"])
    c.robot_joint = (ctypes.c_int * NJ)(*tree["robot_joint"])
    for name, n in (("joint_axis", 3 * NJ), ("joint_origin", 3 * NJ), ("com", 3 * NB), ("inertia", 6 * NB),
                    ("jet_origin", 3 * NJETS), ("jet_axis", 3 * NJETS)):
        setattr(c, name, (ctypes.c_double * n)(*np.asarray(tree[name], float).reshape(-1)))
    c.mass = (ctypes.c_double * NB)(*tree["mass"])
    c.jet_body = (ctypes.c_int * NJETS)(*tree["jet_body"])
    c.gravity = (ctypes.c_double * 3)(*tree["gravity"])
    return c


def pack_state(p_base, R_base, v_base, w_base, q, qd, thrust) -> np.ndarray:
    s = np.zeros(RS_SIZE)
    s[RS_P:RS_P + 3] = p_base
    s[RS_R:RS_R + 9] = np.asarray(R_base, float).reshape(-1)
    s[RS_V:RS_V + 3] = v_base
    s[RS_W:RS_W + 3] = w_base
    s[RS_Q:RS_Q + 8] = q
    s[RS_QD:RS_QD + 8] = qd
    s[RS_T:RS_T + 4] = thrust
    return s
