"""CPU tests of the closed-loop rollout's host side: layouts agree with the header, the synthetic plant hovers at
equal thrust, and the numpy model of the rollout kernels closes the loop with the oracle (tick state machine)."""
import importlib
import os
import re

import numpy as np

import rollout_model as rm
from conftest import PKG, ROOT


def _rollout():
    return importlib.import_module(PKG + ".rollout")


def test_plant_layout_matches_header(layout):
    text = open(os.path.join(ROOT, "include", "vsmpc.h")).read()
    defs = dict(re.findall(r"#define\s+VSMPC_(PS_[A-Z0-9_]+|PP_[A-Z0-9_]+|PLANT_STATE|PLANT_PARAMS|ROLLOUT_LOG)\s+(\d+)", text))
    assert len(defs) >= 25
    for key, val in defs.items():
        assert getattr(layout, key) == int(val), key


def test_synthetic_plant_is_well_posed_and_sliceable(layout):
    ro, cfg = _rollout(), layout.paper_config()
    st, pa = ro.make_plant(cfg, 6, workload="hover")
    st2, pa2 = ro.make_plant(cfg, 3, workload="hover", first_index=3)
    np.testing.assert_array_equal(st[3:], st2)
    np.testing.assert_array_equal(pa[3:], pa2)
    for b in range(6):
        m = pa[b, layout.PP_MASS]
        A = pa[b, layout.PP_AMOM0:layout.PP_AMOM0 + 24].reshape(6, 4)
        w = A @ np.full(4, m * 9.81 / 4.0)                 # wrench of equal hover thrusts, body frame
        assert abs(w[2] - m * 9.81) < 0.01 * m * 9.81       # supports the weight
        assert np.abs(w[:2]).max() < 0.03 * m * 9.81 and np.abs(w[3:]).max() < 6.0   # nearly balanced
        DJ = pa[b, layout.PP_DJ:layout.PP_DJ + 192].reshape(8, 6, 4)
        assert np.all(DJ[:4, :, 1:] == 0) and np.all(DJ[4:, :, [0, 2, 3]] == 0)       # arm joints move their own jet only
        lam = np.einsum("jrc,c->rj", DJ, st[b, layout.PS_T:layout.PS_T + 4])
        assert np.linalg.matrix_rank(lam[3:6]) == 3         # attitude authority through the joints
    stt, pat = ro.make_plant(cfg, 4, workload="takeoff")
    stm, pam = ro.make_plant(cfg, 4, workload="montecarlo")
    assert (pat[:, layout.PP_TICK0] >= 0).all() and (pam[:, layout.PP_DIST_T1] > pam[:, layout.PP_DIST_T0]).all()


def test_trajectory_shapes(layout):
    ro, cfg = _rollout(), layout.paper_config()
    pos, vel, alpha, adt = ro.make_trajectory(cfg, "takeoff", 40.0)
    assert pos.shape == vel.shape == (401, 3) and alpha.shape == (401,) and adt == 0.1
    assert alpha[0] == 0.08 and alpha[-1] == 1.0 and abs(pos[-1, 2] - 2.5) < 1e-12
    assert rm.interp_clamped(alpha, -1.0) == alpha[0] and rm.interp_clamped(alpha, 1e9) == alpha[-1]


def test_model_closed_loop_with_oracle_holds_throttle(layout, ref):
    """25 ticks of one hover instance with the numpy oracle in the loop: every solve is optimal, the throttle command
    changes only on the free tick of the 20-tick hold (constraintsVSMPC.cpp:351-372), the attitude error shrinks."""
    ro, cfg, rcfg = _rollout(), layout.paper_config(), ref.paper_config()
    st, pa = ro.make_plant(cfg, 1, workload="hover", seed0=77)
    pos, vel, alpha, adt = ro.make_trajectory(cfg, "hover", 5.0)
    s, p = st[0].copy(), pa[0]
    tick0 = int(p[layout.PP_TICK0])
    changes = []
    for tick in range(25):
        rec = rm.build_record(cfg, s, p, tick, pos, vel, alpha, adt)
        assert rec[layout.IN_HOLD] == (0.0 if (tick0 + tick) % cfg.ratio == cfg.ratio - 1 else 1.0)
        x, y, _, qp = ref.solve_instance(rcfg, rec)
        cert = ref.kkt_certificate(*qp, x, y)
        assert cert["stationarity_rel"] < 1e-9 and cert["primal"] < 1e-9
        u_before = s[layout.PS_U:layout.PS_U + 4].copy()
        s = rm.advance(cfg, s, p, tick, ref.first_move_vector(rcfg, x), 1, alpha, adt)
        if np.abs(s[layout.PS_U:layout.PS_U + 4] - u_before).max() > 1e-9:
            changes.append((tick0 + tick) % cfg.ratio)
    assert changes and all(c == cfg.ratio - 1 for c in changes)
    assert np.isfinite(s).all() and np.abs(s[layout.PS_RPY:layout.PS_RPY + 3]).max() < 0.2
