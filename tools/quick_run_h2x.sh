#!/bin/bash
# GPU box: measure exp/quick/libvsmpc.so built for the 2x horizon (tools/quick_build.sh 34,14,24): form agreement,
# phase cycles, bench at batch 4096.      tools/quick_run_h2x.sh <tag>
PKG=paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd
TAG=${1:-q}
mkdir -p gpurun_out/quick
cp exp/quick/libvsmpc.so $PKG/libvsmpc.so || exit 1
timeout -k 10 300 python - > gpurun_out/quick/${TAG}_h2x_agree.txt 2>&1 <<'PY' || { tail -20 gpurun_out/quick/${TAG}_h2x_agree.txt; exit 1; }
import importlib, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
solver = importlib.import_module(PKG + ".solver"); layout = importlib.import_module(PKG + ".layout"); synth = importlib.import_module(PKG + ".synth")
cfg = layout.horizon2x_config()
recs = np.concatenate([synth.make_batch(cfg, 24, workload="takeoff"), synth.make_batch(cfg, 24, workload="montecarlo"), synth.make_batch(cfg, 16, workload="hover")])
m = solver.BatchedVSMPC(cfg, device=0, max_batch=64)
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
m.set_kernel_form(solver.KERNEL_FORM_STRUCTURED); a = m.solve(recs); Ma, La = m.debug_condensed(recs[5])[:2]
m.set_kernel_form(solver.KERNEL_FORM_SYRK); b = m.solve(recs); Mb, Lb = m.debug_condensed(recs[5])[:2]
nz = 236
dM = np.abs(np.tril(Ma[:nz + 1, :nz]) - np.tril(Mb[:nz + 1, :nz]))
print("M rel", rel(np.tril(Ma[:nz + 1, :nz]), np.tril(Mb[:nz + 1, :nz])), "L rel", rel(np.tril(La[:nz + 1, :nz]), np.tril(Lb[:nz + 1, :nz])))
bad = np.argwhere(dM > 1e-9 * np.abs(Mb).max())
print("bad entries", len(bad), bad[:20].tolist())
if len(bad):
    tiles = sorted(set((int(r) // 16, int(c) // 16) for r, c in bad)); print("bad tiles", tiles)
print("status equal", (a[2] == b[2]).all(), "iters equal", (a[3] == b[3]).all(), "x rel", rel(a[0], b[0]), "fm rel", rel(a[1], b[1]))
PY
cat gpurun_out/quick/${TAG}_h2x_agree.txt
timeout -k 10 300 python tools/gpu_phases.py h2x > gpurun_out/quick/${TAG}_h2x_phases.txt 2>&1 || { tail -5 gpurun_out/quick/${TAG}_h2x_phases.txt; exit 1; }
grep -v "start\|shader" gpurun_out/quick/${TAG}_h2x_phases.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency --no-extra --config horizon2x --batch 4096 --workload hover --steps 12 --warmup 2 > gpurun_out/quick/${TAG}_h2x_b4096.json 2>/dev/null || exit 1
python - <<PY
import json
d=json.load(open("gpurun_out/quick/${TAG}_h2x_b4096.json")); r=d["roofline"]
print("h2x b4096 us/launch", round(r["kernel_us_per_launch"],2), "frac", round(r["frac"],4), "parity", d.get("parity_max_rel_err_vs_oracle"))
PY
