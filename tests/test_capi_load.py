"""The C-ABI library builds for gfx950, loads, and exports every symbol include/vsmpc.h declares.
No compute call is made here (CPU-only container)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"


def header_functions():
    """every function any header under include/ declares (vsmpc.h: the MPC path, vsmpc_jet.h: the jet plant side)"""
    names = set()
    for header in ("vsmpc.h", "vsmpc_jet.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(vsmpc_[a-z_]+)\s*\(", text))
    return sorted(names)


def test_build_and_exports(solver_mod, pkg):
    import importlib
    _lib = importlib.import_module(pkg.__name__ + "._lib")
    assert os.path.exists(_lib.LIB_PATH)
    cdll = _lib.load()          # imports torch first so that the library binds to torch's HIP runtime
    declared = header_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(cdll, name), f"{name} declared in include/vsmpc.h but not exported"
    assert sorted(_lib.EXPORTS) == declared


def test_header_offsets_match_layout(layout, ref):
    text = open(os.path.join(ROOT, "include", "vsmpc.h")).read()
    defs = dict(re.findall(r"#define\s+VSMPC_(IN_[A-Z0-9]+|FM_[A-Z0-9]+)\s+(\d+)", text))
    for key, val in defs.items():
        assert getattr(layout, key) == int(val), key
        if key.startswith("IN_"):
            assert getattr(ref, key) == int(val), key       # oracle and boundary agree on the record
    assert layout.paper_config().n_in == 294 and layout.horizon2x_config().n_in == 414


def test_config_struct_image(layout):
    c = layout.paper_config().to_c()
    # 4 ints + 3 doubles + 6*3 + 8 + 5 doubles = 16 + 34*8
    assert ctypes.sizeof(c) == 16 + 34 * 8
    assert c.n_iter == 17 and c.n_iter_small == 7 and c.control_horizon == 12 and c.use_jet_dynamic == 1
    assert list(c.w_com_pos_err) == [25000.0, 25000.0, 50000.0] and c.w_throttle == 80000.0


def test_strerror_and_arg_validation_without_gpu(solver_mod, pkg, layout):
    import importlib
    _lib = importlib.import_module(pkg.__name__ + "._lib")
    lib = _lib.load()
    assert lib.vsmpc_strerror(0) == b"ok"
    assert b"unsupported" in lib.vsmpc_strerror(-2)
    # argument checks that need no device: NULL handles / pointers, negative sizes
    assert lib.vsmpc_solve_batch_device(None, None, 4, None, None, None, None, None) == -1
    assert lib.vsmpc_solve_batch(None, None, 4, None, None, None, None, None) == -1
    assert lib.vsmpc_kinematics_batch(None, None, 1, None, None) == -1
    assert lib.vsmpc_rollout_run(None, 5, None, None) == -1
    assert lib.vsmpc_jet_nn_step(None, None, None, 3, 0.001, None, None, None, None) == -1
    hj = ctypes.c_void_p()
    assert lib.vsmpc_jet_create(None, None, None, None, None, None, None, 80, 0, 16, ctypes.byref(hj)) == -1
    h = ctypes.c_void_p()
    bad = layout.MPCConfig(n_iter=1).to_c()
    assert lib.vsmpc_create(ctypes.byref(bad), 0, 4, ctypes.byref(h)) == -1          # invalid argument
    odd = layout.MPCConfig(n_iter=20, n_iter_small=5, control_horizon=9).to_c()
    assert lib.vsmpc_create(ctypes.byref(odd), 0, 4, ctypes.byref(h)) == -2          # no kernel instantiation
    assert lib.vsmpc_create(None, 0, 4, ctypes.byref(h)) == -1
    assert lib.vsmpc_num_variables(None) == -1
    lib.vsmpc_destroy(None)                                                            # must be a no-op
    # rollout entry points validate their arguments before touching the device
    r = ctypes.c_void_p()
    z = np.zeros(3)
    zp = z.ctypes.data_as(ctypes.c_void_p)
    assert lib.vsmpc_rollout_create(None, 4, zp, zp, 1, zp, 1, 0.1, ctypes.byref(r)) == -1
    assert lib.vsmpc_rollout_reset(None, zp, zp) == -1
    assert lib.vsmpc_rollout_run(None, 1, None, None) == -1
    assert lib.vsmpc_rollout_get_state(None, zp) == -1 and lib.vsmpc_rollout_get_records(None, zp) == -1
    lib.vsmpc_rollout_destroy(None)                                                    # must be a no-op


def test_missing_library_fails_loudly(pkg, monkeypatch):
    import importlib
    _lib = importlib.import_module(pkg.__name__ + "._lib")
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libvsmpc.so")
    try:
        _lib.load()
    except _lib.VsmpcError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")


def test_product_path_does_not_touch_oracle():
    pk = os.path.join(ROOT, "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "vsmpc_ref" not in txt and "liboracle" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def test_build_adds_a_horizon_from_the_environment(tmp_path, solver_mod):
    """Row a14: the horizon table is a build-time input (the reference sizes itself from its XML at run time,
    variableSamplingMPC.cpp:24-45).  VSMPC_HORIZONS adds an instantiation: build.horizons() parses it, and a library built
    with an extra horizon (tools/quick_build.sh: a copy of the sources, the tracked table is left alone) carries the solve
    kernel for it.  Compile-only: vsmpc_create needs a device."""
    import importlib
    import subprocess
    build = importlib.import_module(PKG + ".build")
    os.environ["VSMPC_HORIZONS"] = "17,7,12;13,5,8"
    try:
        assert build.horizons() == ((17, 7, 12), (13, 5, 8))
    finally:
        del os.environ["VSMPC_HORIZONS"]
    assert build.horizons() == build.DEFAULT_HORIZONS
    out = str(tmp_path / "quick")
    res = subprocess.run([os.path.join(ROOT, "tools", "quick_build.sh"), "13,5,8"], capture_output=True, text=True,
                         env=dict(os.environ, QUICK_OUT=out))
    assert res.returncode == 0, res.stdout + res.stderr
    lib = os.path.join(out, "libvsmpc.so")
    syms = subprocess.run(["nm", "-C", lib], capture_output=True, text=True, check=True).stdout
    assert "solve_kernel<vsmpc::Dims<13, 5, 8>" in syms and "Dims<17, 7, 12>" not in syms
