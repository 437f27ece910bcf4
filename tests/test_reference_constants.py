"""Everything the reference itself HOLDS for the path — its XML configuration, its jet coefficients (written twice in
the reference: JetModel.cpp:13-26 and jet_kalman_filter.py:6-22) and its trajectory files — against every place this
repo restates a number.  The fixtures are produced from the reference's data files by tools/gen_reference_constants.py
(build container only) and committed; nothing here reads /root/reference."""
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def consts():
    return json.load(open(os.path.join(GOLD, "reference_constants.json")))


@pytest.fixture(scope="module")
def traj():
    return np.load(os.path.join(GOLD, "reference_trajectories.npz"))


def _check_config(cfg, xml):
    assert cfg.n_iter == xml["nIter"] and cfg.n_iter_small == xml["nIterSmall"]
    assert cfg.control_horizon == xml["controlHorizon"]
    assert bool(cfg.use_jet_dynamic) == xml["useJetDynamic"]
    assert cfg.period_mpc == xml["periodMPC"] and cfg.period_small == xml["periodMPCSmallSteps"]
    assert cfg.period_large == xml["periodMPCLargeSteps"]
    for mine, key in (("w_com_pos", "weightCoMPos"), ("w_com_pos_err", "weightCoMPosError"), ("w_lin_mom", "weightLinMom"),
                      ("w_rpy", "weightRPY"), ("w_rpy_err", "weightRPYError"), ("w_ang_mom", "weightAngMom"),
                      ("w_delta_joint", "weightDeltaJoint")):
        assert list(getattr(cfg, mine)) == xml[key], key
    assert cfg.w_throttle == xml["weightThrottle"] and cfg.w_initial_throttle == xml["weightInitialThrottle"]
    assert cfg.w_reg_joint_pos == xml["weightRegularizationJointPos"]
    assert cfg.throttle_min == xml["throttleMin"] and cfg.throttle_max == xml["throttleMax"]


def test_paper_config_is_the_xml(consts, layout, ref):
    xml = consts["VS_MPC_CONFIG"]
    _check_config(layout.paper_config(), xml)        # product-side configuration (-> vsmpc_config)
    _check_config(ref.paper_config(), xml)           # oracle-side configuration
    assert xml["jointsLambdaOption"] == "unfiltered" and xml["useEstimatedThrust"] is True
    assert len(xml["controlledJoints"]) == layout.N_JOINTS
    # sizes that follow (variableSamplingMPC.cpp:42-45)
    assert layout.paper_config().n_var == 588 and layout.paper_config().n_con == 512


def test_jet_coefficients_everywhere(consts, pkg):
    cpp, ekf = consts["jet"]["JetModel.cpp"], consts["jet"]["jet_kalman_filter.py"]
    assert cpp["u2TCoeff"] == ekf["coeffs"] and len(cpp["u2TCoeff"]) == 13            # the reference agrees with itself
    assert cpp["u2Tnormalization"] == [ekf["mean_thrust"], ekf["std_thrust"], ekf["mean_throttle"], ekf["std_throttle"]]
    import importlib
    jm = importlib.import_module(pkg.__name__ + ".jet_model").JetModel()
    assert list(jm.u2TCoeff) == cpp["u2TCoeff"] and list(jm.u2Tnormalization) == cpp["u2Tnormalization"]
    # device constants: struct Jet in csrc/vsmpc_device.hpp
    hpp = open(os.path.join(ROOT, pkg.__name__, "csrc", "vsmpc_device.hpp")).read()
    body = hpp[hpp.index("struct Jet"):]
    dev = [float(re.search(r"\bc%d\s*=\s*([-+0-9.eE]+)" % i, body).group(1)) for i in range(13)]
    assert dev == cpp["u2TCoeff"]
    norm = [float(re.search(r"\b%s\s*=\s*([-+0-9.eE]+)" % k, body).group(1)) for k in ("muT", "sgT", "muU", "sgU")]
    assert norm == cpp["u2Tnormalization"]
    # oracle (numpy and C)
    import vsmpc_ref as ref
    assert [float(v) for v in ref.JET_COEFF] == cpp["u2TCoeff"]
    assert list(ref.JET_NORM) == cpp["u2Tnormalization"]
    csrc = open(os.path.join(ROOT, "oracle", "vsmpc_oracle.c")).read()
    num = r"[-+]?[0-9.]+(?:[eE][-+]?[0-9]+)?"
    jc = [float(v) for v in re.findall(num, re.search(r"JC\[13\]\s*=\s*\{([^}]*)\}", csrc).group(1))]
    jn = [float(v) for v in re.findall(num, re.search(r"JN\[4\]\s*=\s*\{([^}]*)\}", csrc).group(1))]
    assert jc == cpp["u2TCoeff"] and jn == cpp["u2Tnormalization"]


def test_trajectory_files(consts, traj, layout):
    """src/trajectories/*.mat."""
    a = traj["alphaGravity"]
    assert a.shape == (351,) and float(traj["alphaGravity_fps"][0]) == 10.0
    assert (a[:21] == 0.08).all() and a[21] > 0.08            # 0.08 for samples 0..20
    assert (a[200:] == 1.0).all() and a[199] < 1.0            # 1.0 from sample 200 (t = 20 s)
    assert (np.diff(a) >= 0).all()
    for k in ("positionCoM", "velocityCoM", "RPY", "RPYDot"):
        assert traj[k].shape == (1481, 3)
    assert float(traj["trajectory_fps"][0]) == 10.0
    assert (traj["RPY"] == 0).all() and (traj["RPYDot"] == 0).all()
    p, v = traj["positionCoM"], traj["velocityCoM"]
    assert (p[:200] == 0).all() and np.abs(p[201]).max() > 0   # rest until t = 20 s
    np.testing.assert_allclose(p[-1], [0.0, 0.0, 2.5], atol=1e-12)
    assert np.abs(v).max() < 0.95 and p[:, 2].max() <= 3.4 + 1e-9 and p[:, 2].min() >= 0.0
    # the velocity track is the derivative of the position track at the file's rate
    np.testing.assert_allclose(np.gradient(p, 0.1, axis=0)[5:-5], v[5:-5], atol=0.05)
    # sampling period of the file == periodMPCLargeSteps: one window column per sample (costsVSMPC.cpp:68)
    assert 1.0 / float(traj["trajectory_fps"][0]) == layout.paper_config().period_large


def test_synthetic_trajectories_have_the_reference_shape(traj, pkg, layout):
    """rollout.make_trajectory / synth profiles are re-synthesised stand-ins: same rates, same end points."""
    import importlib
    rollout = importlib.import_module(pkg.__name__ + ".rollout")
    synth = importlib.import_module(pkg.__name__ + ".synth")
    pos, vel, alpha, alpha_dt = rollout.make_trajectory(layout.paper_config(), "takeoff", horizon_s=35.0)
    assert alpha_dt == 1.0 / float(traj["alphaGravity_fps"][0])
    assert pos.shape[1] == 3 and vel.shape == pos.shape
    a = traj["alphaGravity"]
    assert alpha[0] == a[0] and alpha[-1] == a[-1]
    assert abs(synth.alpha_gravity_profile(2.0) - a[20]) < 1e-12 and synth.alpha_gravity_profile(20.0) == a[200]
    # reference ramp vs the re-synthesised minimum-jerk blend: same end points, close in between
    t = np.arange(351) * 0.1
    assert np.abs(np.array([synth.alpha_gravity_profile(x) for x in t]) - a).max() < 0.08
    # the reference's own trajectories load through the same interface
    lp, lv, la, ldt = rollout.load_reference_trajectories(os.path.join(GOLD, "reference_trajectories.npz"))
    assert lp.shape == (1481, 3) and lv.shape == (1481, 3) and la.shape == (351,) and ldt == 0.1
