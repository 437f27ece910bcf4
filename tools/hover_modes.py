"""Which synthetic quantity makes the hover loop's lateral mode marginal?  

Monodromy matrix of the oracle-in-the-loop model over one 20-tick hold period (tests/closed_loop_linearisation.py) for
variants of the synthetic plant / loop, each in its own process:
    base        the committed plant (rollout.make_plant, hover, seed 4321)
    lambda0     dA_mom/dq = 0: the joints do not redirect the thrust (Lambda = 0)
    nohold      the 20-tick throttle hold never pins v0 (hold flag 0 on every tick)
    noroll_g    the plant's gravity term frozen in the body frame at the attitude of the tick start (what the MPC's LTI
                model assumes over its horizon: no roll -> lateral force coupling inside a tick) -- isolates the mismatch
    stiff_post  joint-posture weight x 100 (20 -> 2000): less joint wind-up
    settle      the same plant, linearised after 120 periods instead of 30
Prints the leading eigenvalues and the dominant components of the slowest mode.  CPU only (oracle); ~3 min per variant.
    python tools/hover_modes.py [variant ...]
"""
import importlib
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

NAMES = (['px', 'py', 'pz', 'hlx', 'hly', 'hlz', 'roll', 'pitch', 'yaw', 'hax', 'hay', 'haz'] + ['T%d' % i for i in range(4)]
         + ['Td%d' % i for i in range(4)] + ['q%d' % i for i in range(8)] + ['u%d' % i for i in range(4)]
         + ['Tdes%d' % i for i in range(4)] + ['Tddes%d' % i for i in range(4)])


def run(variant):
    import closed_loop_linearisation as cl
    import rollout_model as rm
    cfg, rcfg, ref, _, layout, s0, p, traj = cl.setup()
    settle = 30
    if variant == "lambda0":
        p[layout.PP_DJ:layout.PP_DJ + 192] = 0.0
    if variant == "stiff_post":
        cfg.w_reg_joint_pos = 2000.0
        rcfg.w_reg_joint_pos = 2000.0
    if variant == "settle":
        settle = 120
    if variant == "nohold":
        orig = rm.build_record

        def build(cfg_, model, s, q):
            rec = orig(cfg_, model, s, q)
            rec[layout.IN_HOLD] = 0.0
            return rec
        rm.build_record = build
    if variant == "noroll_g":
        orig_adv = rm.advance

        def advance(cfg_, s, q, tick, fm, status, traj_alpha, alpha_dt, substeps=5, jet=None):
            # integrate with the rotation used for the gravity term frozen at the tick start: emulate by sub-stepping once
            # per tick with the attitude's effect on gravity removed -> run the normal advance, then correct h_lin
            R0 = rm.rot(s[layout.PS_RPY:layout.PS_RPY + 3])
            out = orig_adv(cfg_, s, q, tick, fm, status, traj_alpha, alpha_dt, substeps, jet)
            R1 = rm.rot(out[layout.PS_RPY:layout.PS_RPY + 3])
            m = q[layout.PP_MASS]
            g = np.array([0.0, 0.0, -9.81])
            # trapezoid of the difference between the frozen and the rotating gravity term over the tick
            out[layout.PS_HLIN:layout.PS_HLIN + 3] += cfg_.period_mpc * m * 0.5 * ((R0.T @ g - R0.T @ g) + (R0.T @ g - R1.T @ g))
            return out
        rm.advance = advance
    t = time.time()
    s = s0.copy()
    for _ in range(settle):
        s = cl.period_map(cfg, rcfg, ref, rm, s, p, traj)
    f0 = cl.period_map(cfg, rcfg, ref, rm, s, p, traj)
    M = np.zeros((cl.N_LIN, cl.N_LIN))
    scale = cl.fd_scale(layout)
    for i in range(cl.N_LIN):
        M[:, i] = cl.column(cfg, rcfg, ref, rm, s, p, traj, i, 1e-6 * scale[i])
    w, V = np.linalg.eig(M)
    order = np.argsort(-np.abs(w))
    lines = [f"== {variant}: orbit residual {np.abs(f0 - s)[:12].max():.2e}, {time.time() - t:.0f} s"]
    for k in order[:5]:
        v = V[:, k] / np.abs(V[:, k]).max()
        top = np.argsort(-np.abs(v))[:6]
        lines.append(f"   |lambda| {abs(w[k]):.5f}  arg {np.angle(w[k]):+.3f}   " + ", ".join(f"{NAMES[i]}:{v[i].real:+.2f}" for i in top))
    return "\n".join(lines), variant, M


if __name__ == "__main__":
    variants = sys.argv[1:] or ["base", "lambda0", "nohold", "noroll_g", "stiff_post", "settle"]
    with ProcessPoolExecutor(max_workers=min(6, len(variants))) as ex:
        for text, name, M in ex.map(run, variants):
            print(text, flush=True)
            np.save(os.path.join(ROOT, "gpurun_out", f"hover_modes_{name}.npy"), M)
