#!/usr/bin/env python3
"""Pins every number the reference itself HOLDS for the MPC path into committed fixtures (data, not code):

    tests/golden/reference_constants.json      VS_MPC_CONFIG keys of src/config/vs_mcp_config.xml:7-43,
                                               the jet coefficients / normalisers as written in
                                               src/flight-controller/utils/src/JetModel.cpp:13-26 and, independently,
                                               src/mujoco_lib/jet_kalman_filter.py:6-22, EKF covariances
                                               (ironcub_mujoco_simulator.py:54-56), trajectory metadata
    tests/golden/reference_trajectories.npz    src/trajectories/alphaGravity.mat (1x351) and
                                               minimumJerkTrajectory.mat (positionCoM, velocityCoM, RPY, RPYDot 3x1481,
                                               fps), read with h5dump (MAT-7.3 = HDF5; no h5py in this image)

Runs in the BUILD container only (needs /root/reference and /opt/conda/bin/h5dump); the outputs travel with the repo.
The reference's files are read as DATA / text: nothing of the reference is imported or executed here.

    python tools/gen_reference_constants.py
"""
import json
import os
import re
import subprocess
import sys
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
H5DUMP = "/opt/conda/bin/h5dump"
FLOAT = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def parse_value(text):
    """YARP parameter syntax: scalars, "(a b c)" vectors, quoted strings, true/false."""
    t = text.strip()
    if t in ("true", "false"):
        return t == "true"
    if t.startswith("(") and t.endswith(")"):
        inner = t[1:-1].strip()
        if '"' in inner:
            return re.findall(r'"([^"]*)"', inner)
        return [float(v) for v in inner.replace(",", " ").split()]
    if t.startswith('"') and t.endswith('"'):
        return t[1:-1]
    try:
        return int(t)
    except ValueError:
        return float(t)


def xml_config():
    path = os.path.join(REF, "src/config/vs_mcp_config.xml")
    group = ET.parse(path).getroot().find(".//group[@name='VS_MPC_CONFIG']")
    out = {p.get("name"): parse_value(p.text) for p in group.findall("param")}
    for sub in group.findall("group"):
        out[sub.get("name")] = {p.get("name"): parse_value(p.text) for p in sub.findall("param")}
    return out


def floats_between(text, start_pat, end_pat):
    m = re.search(start_pat + r"(.*?)" + end_pat, text, re.S)
    if not m:
        raise SystemExit(f"pattern {start_pat!r} not found")
    return [float(v) for v in re.findall(FLOAT, m.group(1))]


def jet_constants():
    cpp = open(os.path.join(REF, "src/flight-controller/utils/src/JetModel.cpp")).read()
    py = open(os.path.join(REF, "src/mujoco_lib/jet_kalman_filter.py")).read()
    out = {
        "JetModel.cpp": {
            "u2TCoeff": floats_between(cpp, r"m_u2TCoeff\s*=\s*\{", r"\}"),
            "u2Tnormalization": floats_between(cpp, r"m_u2Tnormalization\s*=\s*\{", r"\}"),   # muT sgT muU sgU
        },
        "jet_kalman_filter.py": {
            "coeffs": floats_between(py, r"self\.coeffs\s*=\s*\[", r"\]"),
            "mean_thrust": float(re.search(r"self\.mean_thrust\s*=\s*(" + FLOAT + ")", py).group(1)),
            "std_thrust": float(re.search(r"self\.std_thrust\s*=\s*(" + FLOAT + ")", py).group(1)),
            "mean_throttle": float(re.search(r"self\.mean_throttle\s*=\s*(" + FLOAT + ")", py).group(1)),
            "std_throttle": float(re.search(r"self\.std_throttle\s*=\s*(" + FLOAT + ")", py).group(1)),
        },
    }
    sim = open(os.path.join(REF, "src/mujoco_lib/ironcub_mujoco_simulator.py")).read()
    out["ekf"] = {k: float(re.search(k + r"\s*=\s*np\.eye\(2\)\s*\*\s*(" + FLOAT + ")", sim).group(1)) for k in ("P", "Q", "R")}
    out["ekf"]["timestep"] = float(re.search(r"self\.model\.opt\.timestep\s*=\s*(" + FLOAT + ")", sim).group(1))
    return out


def h5_dataset(path, name):
    """One dataset of a MAT-7.3 file as a numpy array (HDF5 dataspace order), through h5dump's text output."""
    txt = subprocess.run([H5DUMP, "-m", "%.17g", "-w", "0", "-d", "/" + name, path], check=True, capture_output=True,
                         text=True).stdout
    dims = [int(v) for v in re.search(r"DATASPACE\s+SIMPLE\s*\{\s*\(([^)]*)\)", txt).group(1).split(",")]
    body = txt[txt.index("DATA {") + 6:]
    vals = [float(v) for v in re.findall(r"(?:\(\d+(?:,\d+)*\):)?\s*(" + FLOAT + r")\s*,?", re.sub(r"\(\d+(?:,\d+)*\):", " ", body))]
    n = int(np.prod(dims))
    return np.array(vals[:n]).reshape(dims)


def trajectories():
    tdir = os.path.join(REF, "src/trajectories")
    out = {}
    a = os.path.join(tdir, "alphaGravity.mat")
    out["alphaGravity"] = h5_dataset(a, "alphaGravity").reshape(-1)             # HDF5 (351,1) = MATLAB 1x351
    out["alphaGravity_fps"] = h5_dataset(a, "fps").reshape(-1)
    m = os.path.join(tdir, "minimumJerkTrajectory.mat")
    for name in ("positionCoM", "velocityCoM", "RPY", "RPYDot"):
        out[name] = h5_dataset(m, name)                                          # HDF5 (1481,3) = MATLAB 3x1481
    out["trajectory_fps"] = h5_dataset(m, "fps").reshape(-1)
    return out


def main():
    if not os.path.isdir(REF) or not os.path.exists(H5DUMP):
        raise SystemExit("needs /root/reference and /opt/conda/bin/h5dump (build container only)")
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    traj = trajectories()
    consts = {
        "generated_by": "tools/gen_reference_constants.py",
        "sources": ["src/config/vs_mcp_config.xml", "src/flight-controller/utils/src/JetModel.cpp",
                    "src/mujoco_lib/jet_kalman_filter.py", "src/mujoco_lib/ironcub_mujoco_simulator.py",
                    "src/trajectories/alphaGravity.mat", "src/trajectories/minimumJerkTrajectory.mat"],
        "VS_MPC_CONFIG": xml_config(),
        "jet": jet_constants(),
        "trajectories": {k: {"shape": list(v.shape), "min": float(v.min()), "max": float(v.max()),
                             "first": float(v.reshape(-1)[0]), "last": float(v.reshape(-1)[-1])} for k, v in traj.items()},
    }
    json.dump(consts, open(os.path.join(gold, "reference_constants.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(gold, "reference_trajectories.npz"), **traj)
    print(json.dumps(consts["trajectories"], indent=1))
    print("wrote", os.path.join(gold, "reference_constants.json"), os.path.join(gold, "reference_trajectories.npz"))


if __name__ == "__main__":
    sys.exit(main())
