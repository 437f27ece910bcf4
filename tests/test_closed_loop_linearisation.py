"""The committed linearisation of the closed loop (tests/golden/hover_monodromy.npz, the bound of the GPU hover property
test) against a fresh derivation: the stored point still behaves like the hover trajectory it was taken from, and two
columns of the monodromy matrix recomputed by central differences equal the stored ones."""
import numpy as np

import closed_loop_linearisation as cl


def test_fixture_matches_a_fresh_derivation():
    M, orbit, rho = cl.load_fixture()
    cfg, rcfg, ref, rm, layout, _, p, traj = cl.setup()
    assert M.shape == (cl.N_LIN, cl.N_LIN) and orbit.shape == (layout.PLANT_STATE,)
    f0 = cl.period_map(cfg, rcfg, ref, rm, orbit, p, traj)
    # 3 s of settling leave the slow lateral mode still moving (rho > 1): the point is a point of the hover trajectory, not
    # a fixed point; positions and attitude drift by centimetres / hundredths of a radian per period
    res = np.abs(f0 - orbit)
    assert res[layout.PS_P:layout.PS_P + 3].max() < 0.05 and res[layout.PS_RPY:layout.PS_RPY + 3].max() < 0.05
    scale = cl.fd_scale(layout)
    for i in (layout.PS_RPY, layout.PS_HLIN + 1):               # roll, lateral momentum: the slow lateral mode lives here
        col = cl.column(cfg, rcfg, ref, rm, orbit, p, traj, i, 1e-6 * scale[i])
        np.testing.assert_allclose(col, M[:, i], rtol=1e-5, atol=1e-7)
    assert abs(cl.spectral_radius(M) - rho) < 1e-12 and rho < 1.008


def test_the_marginal_modes_are_the_ones_the_cost_does_not_see():
    """What rho > 1 is made of (tools/hover_modes.py, profiles/r03_hover_modes.txt): every multiplier
    outside the unit circle belongs to one of two families the reference's cost barely weighs --
      * a complex pair in the lateral momentum h_lin,y against the differential thrust RATE of the two arm jets
        (|lambda| 1.0062 per 0.1 s hold period, period 2.65 s: e-folding time 16 s), and
      * real multipliers within 0.2 % of 1 in which thrust moves from one jet to another at constant total force (the cost
        penalises throttle increments only, costsVSMPC.cpp:383-409, so the split between the jets is a free integrator);
    everything else contracts.  The same numbers come out with the plant's gravity frozen in the body frame, with a 100x
    posture weight and after 12 s instead of 3 s of settling; without the 20-tick hold the thrust-split family gets WORSE
    (1.028), with dA_mom/dq = 0 the lateral pair drops to 1.0005: the modes belong to the formulation, not to the loop code."""
    import importlib
    import os
    import sys
    M, _, rho = cl.load_fixture()
    layout = importlib.import_module("paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd.layout")
    w, V = np.linalg.eig(M)
    order = np.argsort(-np.abs(w))
    lead = order[0]
    assert abs(abs(w[lead]) - rho) < 1e-12 and abs(w[lead].imag) > 0.1          # the complex pair leads
    v = np.abs(V[:, lead]) / np.abs(V[:, lead]).max()
    hly = layout.PS_HLIN + 1
    assert v[hly] == 1.0                                                         # lateral momentum dominates it ...
    td = v[layout.PS_TD:layout.PS_TD + 4]
    assert td[0] > 0.5 and td[1] > 0.5 and td[2] < 0.4 and td[3] < 0.4           # ... with the arm jets' thrust rates
    assert v[layout.PS_Q:layout.PS_Q + 8].max() < 0.2                            # and no joint wind-up
    period_s = 2 * np.pi / abs(np.angle(w[lead])) * 0.1
    assert 2.0 < period_s < 3.5
    # every other multiplier outside the unit circle is real and a thrust-split mode
    for k in order[2:]:
        if abs(w[k]) <= 1.0:
            break
        vk = np.abs(V[:, k]) / np.abs(V[:, k]).max()
        assert abs(w[k].imag) < 1e-9 and abs(w[k]) < 1.002
        assert vk[layout.PS_T:layout.PS_T + 4].max() == 1.0 or vk[layout.PS_TDES:layout.PS_TDES + 4].max() == 1.0
    # the modes the cost does weigh contract: altitude (h_lin,z: a pair at 0.979 and a real one at 0.919 per period, i.e.
    # time constants 4.7 s and 1.2 s) and everything the angular momentum leads
    hlz, hang = layout.PS_HLIN + 2, slice(layout.PS_HANG, layout.PS_HANG + 3)
    lead_of = [int(np.argmax(np.abs(V[:, k]))) for k in range(len(w))]
    vertical = [abs(w[k]) for k in range(len(w)) if lead_of[k] == hlz]
    assert len(vertical) >= 3 and max(vertical) < 0.985
    assert all(abs(w[k]) < 0.6 for k in range(len(w)) if hang.start <= lead_of[k] < hang.stop)
    # and the longitudinal twin of the lateral pair sits just inside the circle: which side of it the pair falls on is a
    # matter of the synthetic jet geometry, not of the loop
    hlx_pairs = [abs(w[k]) for k in range(len(w)) if lead_of[k] == layout.PS_HLIN and abs(w[k].imag) > 0.1 and abs(w[k]) > 0.9]
    assert hlx_pairs and max(hlx_pairs) < 1.0


def test_tree_plant_fixture_matches_a_fresh_derivation():
    """The same linearisation on the KINEMATIC-TREE plant (vsmpc_rollout_set_tree; tests/closed_loop_linearisation.py --tree
    --settle 100): the plant whose Lambda is its own kinematics.  Two columns recomputed from scratch, and the structure of
    the spectrum: the lateral-momentum / arm-jet thrust-rate pair still leads, at 1.0022 per hold period instead of 1.0062."""
    import importlib
    import rollout_model as rm
    M, orbit, rho = cl.load_fixture(tree=True)
    layout = importlib.import_module("paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd.layout")
    cfg, rcfg, ref, _, _, _, _, traj = cl.setup()
    tr, st, pa = cl.tree_plant()
    rm.set_tree(tr)
    try:
        p = pa[0].copy()
        scale = cl.fd_scale(layout)
        for i in (layout.PS_HLIN + 1, layout.PS_Q + 2):
            col = cl.column(cfg, rcfg, ref, rm, orbit, p, traj, i, 1e-6 * scale[i])
            np.testing.assert_allclose(col, M[:, i], rtol=1e-5, atol=1e-7)
    finally:
        rm.set_tree(None)
    assert abs(cl.spectral_radius(M) - rho) < 1e-12 and 1.0 < rho < 1.003
    w, V = np.linalg.eig(M)
    lead = int(np.argmax(np.abs(w)))
    v = np.abs(V[:, lead]) / np.abs(V[:, lead]).max()
    assert abs(w[lead].imag) > 0.1 and v[layout.PS_HLIN + 1] == 1.0                      # lateral momentum ...
    assert v[layout.PS_TD] > 0.4 and v[layout.PS_TD + 1] > 0.4                           # ... against the arm jets' thrust rates
    assert 2.0 < 2 * np.pi / abs(np.angle(w[lead])) * 0.1 < 3.5                          # period in seconds
