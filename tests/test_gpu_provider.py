"""Row N2: the batched kinematics provider (vsmpc_provider_batch, csrc/vsmpc_provider.hip) on the committed simplified
tree (robot_tree.py) against oracle/robot_tree_ref.py -- parity unpinned: the reference computes these quantities with
iDynTree on the iRonCub URDF, neither of which is in this image -- and the provider -> kinematics -> solve chain."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT, relerr

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import robot_tree_ref as rt  # noqa: E402

pytestmark = pytest.mark.gpu


def random_states(rng, n):
    out = []
    for _ in range(n):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        out.append(dict(p_base=rng.normal(0, 0.5, 3), R_base=q * np.sign(np.linalg.det(q)), v_base=rng.normal(0, 0.3, 3),
                        w_base=rng.normal(0, 0.5, 3), q=rng.normal(0, 0.6, 8), qd=rng.normal(0, 0.8, 8),
                        thrust=rng.uniform(100, 220, 4)))
    return out


def test_provider_matches_oracle(solver_mod, layout, ref):
    RT = importlib.import_module(PKG + ".robot_tree")
    tree = RT.default_tree()
    rng = np.random.default_rng(21)
    B = 33
    sts = random_states(rng, B)
    packed = np.stack([RT.pack_state(s["p_base"], s["R_base"], s["v_base"], s["w_base"], s["q"], s["qd"], s["thrust"]) for s in sts])
    cfg = layout.paper_config()
    m = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=64)
    try:
        kin, robot = m.provider(tree, packed)
        for b in range(B):
            o = rt.forward(tree, sts[b])
            k = rt.kin_record(o, sts[b], layout)
            assert relerr(kin[b], k) < 1e-13, int(np.abs(kin[b] - k).argmax())
            assert relerr(robot[b, RT.RO_COM:RT.RO_COM + 3], o["com"]) < 1e-14
            assert relerr(robot[b, RT.RO_MOM:RT.RO_MOM + 6], o["momentum"]) < 1e-13
            assert relerr(robot[b, RT.RO_MOMB:RT.RO_MOMB + 6], o["momentum_body"]) < 1e-13
            assert abs(robot[b, RT.RO_MASS] - o["mass"]) < 1e-12
            assert relerr(robot[b, RT.RO_AMOM:RT.RO_AMOM + 24], o["Amom"].reshape(-1)) < 1e-14
            assert relerr(robot[b, RT.RO_AMOMB:RT.RO_AMOMB + 24], o["Amom_body"].reshape(-1)) < 1e-14
        # unmodelled joints have zero Jacobian columns
        J = kin[:, layout.KIN_JFRAME:layout.KIN_JFRAME + 276].reshape(B, 4, 3, 23)
        assert np.abs(J[..., [0, 1, 2] + list(range(11, 23))]).max() == 0 and np.abs(J[..., 3:11]).max() > 0
        # the chest jets sit on the base: no joint moves them; an arm jet is moved by its own arm only
        assert np.abs(J[:, 2:4]).max() == 0 and np.abs(J[:, 0, :, 7:11]).max() == 0 and np.abs(J[:, 1, :, 3:7]).max() == 0
    finally:
        m.close()


def test_provider_feeds_the_solve_without_the_host(solver_mod, synth, layout, ref):
    """provider -> kinematics terms -> solve: the Robot-derived fields of hover records are replaced by the provider's (on
    the device), and the result equals the oracle's solve of the records assembled from the oracle's Robot quantities."""
    RT = importlib.import_module(PKG + ".robot_tree")
    tree = RT.default_tree()
    cfg, rcfg = layout.paper_config(), ref.paper_config()
    rng = np.random.default_rng(22)
    B = 6
    sts = random_states(rng, B)
    for s in sts:                      # a hovering attitude: small tilt, the jets roughly carry the weight
        s["R_base"] = rt.rodrigues(rng.normal(size=3), 0.05)
        s["w_base"] = rng.normal(0, 0.05, 3); s["v_base"] = rng.normal(0, 0.05, 3); s["qd"] = rng.normal(0, 0.05, 8)
        s["q"] = rng.normal(0, 0.15, 8)
        s["thrust"] = sum(tree["mass"]) * 9.81 / 4 * (1 + rng.normal(0, 0.03, 4))
    packed = np.stack([RT.pack_state(s["p_base"], s["R_base"], s["v_base"], s["w_base"], s["q"], s["qd"], s["thrust"]) for s in sts])
    recs = synth.make_batch(cfg, B, workload="hover")
    expect = recs.copy()
    m = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=8)
    try:
        kin, robot = m.provider(tree, packed, recs)
        for b in range(B):
            o = rt.forward(tree, sts[b])
            e = expect[b]
            R = sts[b]["R_base"]
            rpy = np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], np.hypot(R[2, 1], R[2, 2])), np.arctan2(R[1, 0], R[0, 0])])
            e[layout.IN_X0 + 0:layout.IN_X0 + 3] = o["com"]
            e[layout.IN_X0 + 3:layout.IN_X0 + 6] = o["momentum_body"][0:3]
            e[layout.IN_X0 + 6:layout.IN_X0 + 9] = rpy
            e[layout.IN_X0 + 9:layout.IN_X0 + 12] = o["momentum_body"][3:6]
            e[layout.IN_X0 + 12:layout.IN_X0 + 16] = sts[b]["thrust"]
            e[layout.IN_T0:layout.IN_T0 + 4] = sts[b]["thrust"]
            e[layout.IN_MASS] = float(np.float32(o["mass"]))
            e[layout.IN_WRB:layout.IN_WRB + 9] = R.reshape(-1)
            e[layout.IN_OMEGA:layout.IN_OMEGA + 3] = R.T @ sts[b]["w_base"]
            e[layout.IN_GRAV:layout.IN_GRAV + 3] = tree["gravity"]
            e[layout.IN_AMOM:layout.IN_AMOM + 24] = o["Amom_body"].reshape(-1)
            e[layout.IN_RPY:layout.IN_RPY + 3] = rpy
            l1, l2, ig = ref.kinematics_terms(rt.kin_record(o, sts[b], layout))
            e[layout.IN_LLIN:layout.IN_LLIN + 24] = l1.reshape(-1)
            e[layout.IN_LANG:layout.IN_LANG + 24] = l2.reshape(-1)
            e[layout.IN_INERTIA:layout.IN_INERTIA + 9] = ig.reshape(-1)
            assert relerr(recs[b], e) < 1e-12, int(np.abs(recs[b] - e).argmax())
        x, fm, st, it = m.solve(recs)
        assert (st == layout.STATUS_SOLVED).all()
        for b in (0, B - 1):
            xr, _, _, _ = ref.solve_instance(rcfg, expect[b])
            assert relerr(x[b], xr) < 1e-8
    finally:
        m.close()


def test_provider_refuses_constant_lambda(solver_mod, synth, layout):
    """jointsLambdaOption 'constant' re-reads the Jacobian slots of the kinematics record as configure-time quantities the
    provider does not deliver: patching records through the provider with that option set is refused, not silently wrong."""
    RT = importlib.import_module(PKG + ".robot_tree")
    tree = RT.default_tree()
    cfg = layout.paper_config()
    rng = np.random.default_rng(5)
    sts = random_states(rng, 2)
    packed = np.stack([RT.pack_state(s["p_base"], s["R_base"], s["v_base"], s["w_base"], s["q"], s["qd"], s["thrust"]) for s in sts])
    recs = synth.make_batch(cfg, 2, workload="hover")
    m = solver_mod.BatchedVSMPC(cfg, device=0, max_batch=4)
    try:
        m.set_kinematics_options(None, True)
        before = recs.copy()
        with pytest.raises(Exception, match="unsupported"):
            m.provider(tree, packed, recs)
        np.testing.assert_array_equal(recs, before)
        m.provider(tree, packed)                       # the Robot-level outputs alone do not depend on the option
        m.set_kinematics_options(None, False)
        m.provider(tree, packed, recs)
    finally:
        m.close()
