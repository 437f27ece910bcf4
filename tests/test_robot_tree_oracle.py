"""oracle/robot_tree_ref.py (the kinematics-provider oracle, parity unpinned: no iDynTree here) checked for internal
consistency: Jacobians against finite differences of the forward kinematics, the mass-matrix base block against the
kinetic energy, the centroidal momentum against the CoM Jacobian and against its own time derivative structure."""
import importlib
import os
import sys

import numpy as np

from conftest import PKG, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import robot_tree_ref as rt  # noqa: E402


def random_state(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return dict(p_base=rng.normal(0, 0.5, 3), R_base=q * np.sign(np.linalg.det(q)), v_base=rng.normal(0, 0.3, 3),
                w_base=rng.normal(0, 0.5, 3), q=rng.normal(0, 0.5, 8), qd=rng.normal(0, 0.8, 8),
                thrust=rng.uniform(100, 220, 4))


def perturbed(state, nu, eps):
    """state after moving along the generalised velocity nu = (v_base, w_base, qd[8]) for time eps"""
    s = dict(state)
    s["p_base"] = state["p_base"] + eps * nu[0:3]
    s["R_base"] = rt.rodrigues(nu[3:6], eps * np.linalg.norm(nu[3:6])) @ state["R_base"] if np.linalg.norm(nu[3:6]) > 0 else state["R_base"]
    s["q"] = state["q"] + eps * nu[6:14]
    return s


def test_jacobians_match_finite_differences():
    tree = importlib.import_module(PKG + ".robot_tree").default_tree()
    rng = np.random.default_rng(5)
    for _ in range(4):
        st = random_state(rng)
        out = rt.forward(tree, st)
        nu = np.concatenate([st["v_base"], st["w_base"], st["qd"]])
        full = np.zeros(6 + 23)
        full[0:6] = nu[0:6]
        full[6 + np.array(tree["robot_joint"])] = nu[6:14]
        eps = 1e-6
        a, b = rt.forward(tree, perturbed(st, nu, eps)), rt.forward(tree, perturbed(st, nu, -eps))
        assert np.abs((a["com"] - b["com"]) / (2 * eps) - out["Jcom"] @ full).max() < 1e-7
        for i in range(4):
            assert np.abs((a["jet_pos"][i] - b["jet_pos"][i]) / (2 * eps) - out["Jjet"][i][0:3] @ full).max() < 1e-7
            # angular rows: the jet axis rotates with omega = J_ang nu
            wj = out["Jjet"][i][3:6] @ full
            assert np.abs((a["axes"][i] - b["axes"][i]) / (2 * eps) - np.cross(wj, out["axes"][i])).max() < 1e-7
            # relative Jacobian = joint columns seen from the base, in base axes
            R = st["R_base"]
            assert np.abs(out["Jrel"][i] - np.vstack([R.T @ out["Jjet"][i][0:3, 6:], R.T @ out["Jjet"][i][3:6, 6:]])).max() == 0
        # momentum against the CoM Jacobian, and the body-coordinate version
        assert np.abs(out["momentum"][0:3] - out["mass"] * (out["Jcom"] @ full)).max() < 1e-10
        assert np.abs(out["momentum_body"][3:6] - st["R_base"].T @ out["momentum"][3:6]).max() < 1e-14


def test_mass_matrix_block_and_momentum_with_locked_joints():
    tree = importlib.import_module(PKG + ".robot_tree").default_tree()
    rng = np.random.default_rng(6)
    st = random_state(rng)
    st["qd"] = np.zeros(8)                                   # joints locked: the robot is one rigid body
    out = rt.forward(tree, st)
    nub = np.concatenate([st["v_base"], st["w_base"]])
    kinetic = sum(0.5 * m * v @ v + 0.5 * w @ I @ w for m, v, w, I in
                  zip(tree["mass"], out["body_vc"], out["body_w"], out["body_Iw"]))
    assert abs(0.5 * nub @ out["Mb"] @ nub - kinetic) < 1e-10 * max(1.0, kinetic)
    assert abs(out["Mb"][0, 0] - sum(tree["mass"])) < 1e-12 and np.abs(out["Mb"] - out["Mb"].T).max() < 1e-12
    # centroidal momentum = X^T Mb nu with X the base -> CoM transform (locked joints), i.e. the locked inertia of a3/a4
    c = out["com"] - st["p_base"]
    h_lin = out["mass"] * (st["v_base"] + np.cross(st["w_base"], c))
    IG = out["Mb"][3:6, 3:6] - out["mass"] * (c @ c * np.eye(3) - np.outer(c, c))
    assert np.abs(out["momentum"][0:3] - h_lin).max() < 1e-10
    assert np.abs(out["momentum"][3:6] - IG @ st["w_base"]).max() < 1e-10
    # A_mom: force along the axis at the jet position -> wrench about the CoM
    for i in range(4):
        assert np.abs(out["Amom"][3:6, i] - np.cross(out["jet_pos"][i] - out["com"], out["axes"][i])).max() < 1e-14
