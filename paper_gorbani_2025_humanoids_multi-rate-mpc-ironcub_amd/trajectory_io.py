"""Trajectory files of the reference's TrajectoryManager (utils/src/TrajectoryManager.cpp:67-140): a MAT file whose
variables are the tracks (dims[0] = track dimension, dims[1] = samples, column-major) plus a scalar `fps`.

`load_mat_trajectory(path)` returns {variable: array [samples, dim], "fps": int} -- the mapping the front ends
(reference_api.VariableSamplingMPC, bindingsMPC) accept for the POSITION_TRAJECTORY / TRAJECTORY_MANAGER groups -- and is
what `momentum_based_mpc.bindingsMPC` installs as the loader for groups that hold a `trajectoryFile` name.  Decoding is
delegated to whichever reader the environment has: scipy.io.loadmat for MAT <= 7.2, h5py / hdf5storage for MAT 7.3 (the
shipped files).  No reader for a file's format -> a RuntimeError naming it; no silent default."""
from __future__ import annotations

import os

import numpy as np


def _resolve(path: str) -> str:
    """yarp's ResourceFinder::findFileByName (TrajectoryManager.cpp:71-73) searches its data directories; here: the path as
    given, relative to the working directory, then relative to $VSMPC_TRAJECTORY_ROOT."""
    path = path.strip().strip('"')
    cands = [path]
    root = os.environ.get("VSMPC_TRAJECTORY_ROOT")
    if root:
        cands.append(os.path.join(root, path))
    for c in cands:
        if os.path.exists(c):
            return c
    raise FileNotFoundError(f"trajectory file '{path}' not found (searched {cands}; set VSMPC_TRAJECTORY_ROOT)")


def _tracks(raw: dict) -> dict:
    out = {}
    fps = None
    for name, value in raw.items():
        if name.startswith("__") or name.startswith("#"):
            continue
        a = np.asarray(value, dtype=float)
        if name == "fps":
            fps = int(a.reshape(-1)[0])
            continue
        a = np.atleast_2d(a)
        # MAT layout: dim x samples (TrajectoryManager.cpp:104-112 pushes one COLUMN per sample)
        out[name] = np.ascontiguousarray(a.T)
    if fps is None:
        raise RuntimeError("trajectory file has no `fps` variable (TrajectoryManager.cpp:84-91)")
    out["fps"] = fps
    return out


def load_mat_trajectory(path: str) -> dict:
    full = _resolve(path)
    with open(full, "rb") as f:
        head = f.read(128)
    if b"MATLAB 7.3" in head or head[:8] == b"\x89HDF\r\n\x1a\n" or b"HDF" in head[:16]:
        try:
            import h5py
        except ImportError as e:  # pragma: no cover - depends on the environment
            raise RuntimeError(f"'{full}' is a MAT 7.3 (HDF5) file and h5py is not installed; pass arrays or install a loader "
                               "with set_trajectory_loader") from e
        with h5py.File(full, "r") as h:  # pragma: no cover - h5py is absent in the build image
            # h5py presents MATLAB's column-major dim x samples arrays transposed (samples x dim): undo for _tracks
            return _tracks({k: np.asarray(v).T for k, v in h.items() if hasattr(v, "shape")})
    import scipy.io
    return _tracks(scipy.io.loadmat(full))
