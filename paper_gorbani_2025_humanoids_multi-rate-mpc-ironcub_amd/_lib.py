"""ctypes binding of include/vsmpc.h.  Loading fails loudly when libvsmpc.so is missing: there is
no CPU fallback anywhere in the product path."""
from __future__ import annotations

import ctypes
import os

from .layout import CConfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvsmpc.so")

# every symbol include/vsmpc.h declares
EXPORTS = (
    "vsmpc_create", "vsmpc_destroy", "vsmpc_num_variables", "vsmpc_num_constraints", "vsmpc_input_doubles",
    "vsmpc_max_batch", "vsmpc_solve_batch", "vsmpc_solve_batch_device", "vsmpc_linearize_batch",
    "vsmpc_assemble_dense", "vsmpc_condensed_dim", "vsmpc_debug_condensed", "vsmpc_timing_begin",
    "vsmpc_timing_end", "vsmpc_strerror", "vsmpc_kernel_name", "vsmpc_debug_phase_cycles", "vsmpc_kinematics_batch",
    "vsmpc_rollout_create", "vsmpc_rollout_destroy", "vsmpc_rollout_reset", "vsmpc_rollout_run",
    "vsmpc_rollout_get_state", "vsmpc_rollout_get_records", "vsmpc_alloc_host", "vsmpc_free_host",
    "vsmpc_set_kernel_form", "vsmpc_set_kinematics_options", "vsmpc_provider_batch", "vsmpc_rollout_set_attitude_tracks",
    "vsmpc_tick", "vsmpc_rollout_set_tree",
    # include/vsmpc_jet.h
    "vsmpc_jet_create", "vsmpc_jet_destroy", "vsmpc_jet_nn_step", "vsmpc_jet_nn_sequence", "vsmpc_jet_ekf_update",
    "vsmpc_jet_plant_run", "vsmpc_jet_plant_run_device", "vsmpc_rollout_set_jet_plant",
)

_lib = None


class VsmpcError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VsmpcError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "The batched MPC path has no CPU fallback.")
    try:  # share torch's HIP runtime (same SONAME) when torch is in the process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C-ABI itself
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, ip, dp, c_int = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.c_int
    lib.vsmpc_create.argtypes = [ctypes.POINTER(CConfig), c_int, c_int, ctypes.POINTER(vp)]
    lib.vsmpc_create.restype = c_int
    lib.vsmpc_destroy.argtypes = [vp]
    lib.vsmpc_destroy.restype = None
    for name in ("vsmpc_num_variables", "vsmpc_num_constraints", "vsmpc_input_doubles", "vsmpc_max_batch",
                 "vsmpc_condensed_dim"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = c_int
    lib.vsmpc_solve_batch.argtypes = [vp, dp, c_int, dp, dp, vp, vp, vp]
    lib.vsmpc_solve_batch.restype = c_int
    lib.vsmpc_solve_batch_device.argtypes = [vp, dp, c_int, dp, dp, vp, vp, vp]
    lib.vsmpc_solve_batch_device.restype = c_int
    lib.vsmpc_linearize_batch.argtypes = [vp, dp, c_int, dp, dp, dp, dp, dp]
    lib.vsmpc_linearize_batch.restype = c_int
    lib.vsmpc_assemble_dense.argtypes = [vp, dp, dp, dp, dp, dp, dp]
    lib.vsmpc_assemble_dense.restype = c_int
    lib.vsmpc_debug_condensed.argtypes = [vp, dp, dp, dp]
    lib.vsmpc_debug_condensed.restype = c_int
    lib.vsmpc_kinematics_batch.argtypes = [vp, dp, c_int, dp, dp]
    lib.vsmpc_kinematics_batch.restype = c_int
    lib.vsmpc_debug_phase_cycles.argtypes = [vp, dp, c_int, vp]
    lib.vsmpc_debug_phase_cycles.restype = c_int
    lib.vsmpc_timing_begin.argtypes = [vp, vp]
    lib.vsmpc_timing_begin.restype = c_int
    lib.vsmpc_timing_end.argtypes = [vp, vp, c_int, ctypes.POINTER(ctypes.c_float)]
    lib.vsmpc_timing_end.restype = c_int
    lib.vsmpc_rollout_create.argtypes = [vp, c_int, dp, dp, c_int, dp, c_int, ctypes.c_double, ctypes.POINTER(vp)]
    lib.vsmpc_rollout_create.restype = c_int
    lib.vsmpc_rollout_destroy.argtypes = [vp]
    lib.vsmpc_rollout_destroy.restype = None
    lib.vsmpc_rollout_reset.argtypes = [vp, dp, dp]
    lib.vsmpc_rollout_reset.restype = c_int
    lib.vsmpc_rollout_run.argtypes = [vp, c_int, dp, vp]
    lib.vsmpc_rollout_run.restype = c_int
    lib.vsmpc_rollout_get_state.argtypes = [vp, dp]
    lib.vsmpc_rollout_get_state.restype = c_int
    lib.vsmpc_rollout_get_records.argtypes = [vp, dp]
    lib.vsmpc_rollout_get_records.restype = c_int
    lib.vsmpc_alloc_host.argtypes = [ctypes.c_size_t]
    lib.vsmpc_alloc_host.restype = vp
    lib.vsmpc_free_host.argtypes = [vp]
    lib.vsmpc_free_host.restype = None
    lib.vsmpc_set_kernel_form.argtypes = [vp, c_int]
    lib.vsmpc_set_kernel_form.restype = c_int
    lib.vsmpc_set_kinematics_options.argtypes = [vp, ip, c_int]
    lib.vsmpc_set_kinematics_options.restype = c_int
    lib.vsmpc_provider_batch.argtypes = [vp, vp, dp, c_int, dp, dp, dp]
    lib.vsmpc_provider_batch.restype = c_int
    lib.vsmpc_rollout_set_attitude_tracks.argtypes = [vp, dp, dp]
    lib.vsmpc_rollout_set_attitude_tracks.restype = c_int
    lib.vsmpc_rollout_set_tree.argtypes = [vp, vp]
    lib.vsmpc_rollout_set_tree.restype = c_int
    lib.vsmpc_tick.argtypes = [vp, dp, dp, c_int, dp, dp, vp, vp, vp]
    lib.vsmpc_tick.restype = c_int
    lib.vsmpc_rollout_set_jet_plant.argtypes = [vp, vp, dp, dp]
    lib.vsmpc_rollout_set_jet_plant.restype = c_int
    fp = ctypes.c_void_p
    lib.vsmpc_jet_create.argtypes = [fp, fp, fp, fp, fp, fp, dp, c_int, c_int, c_int, ctypes.POINTER(vp)]
    lib.vsmpc_jet_create.restype = c_int
    lib.vsmpc_jet_destroy.argtypes = [vp]
    lib.vsmpc_jet_destroy.restype = None
    lib.vsmpc_jet_nn_step.argtypes = [vp, fp, fp, c_int, ctypes.c_float, fp, fp, fp, fp]
    lib.vsmpc_jet_nn_step.restype = c_int
    lib.vsmpc_jet_nn_sequence.argtypes = [vp, fp, c_int, c_int, ctypes.c_float, fp, fp, fp, fp]
    lib.vsmpc_jet_nn_sequence.restype = c_int
    lib.vsmpc_jet_ekf_update.argtypes = [vp, dp, dp, dp, dp, c_int, ctypes.c_double, dp, dp]
    lib.vsmpc_jet_ekf_update.restype = c_int
    lib.vsmpc_jet_plant_run.argtypes = [vp, fp, dp, dp, fp, c_int, c_int, c_int, ctypes.c_double, dp, dp, dp]
    lib.vsmpc_jet_plant_run.restype = c_int
    lib.vsmpc_jet_plant_run_device.argtypes = [vp, fp, dp, dp, fp, c_int, c_int, c_int, ctypes.c_double, dp, dp, dp, vp]
    lib.vsmpc_jet_plant_run_device.restype = c_int
    lib.vsmpc_strerror.argtypes = [c_int]
    lib.vsmpc_strerror.restype = ctypes.c_char_p
    lib.vsmpc_kernel_name.argtypes = [vp]
    lib.vsmpc_kernel_name.restype = ctypes.c_char_p
    _lib = lib
    return lib


def check(code: int, what: str = "vsmpc"):
    if code != 0:
        msg = load().vsmpc_strerror(code).decode()
        raise VsmpcError(f"{what} failed ({code}): {msg}")
