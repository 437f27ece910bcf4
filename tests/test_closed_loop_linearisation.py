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
    assert abs(cl.spectral_radius(M) - rho) < 1e-12 and 1.0 < rho < 1.02
