#!/bin/bash
# GPU box: measure exp/quick/libvsmpc.so (paper horizon only): phase cycles + bench at batch 256 and 4096.
#   tools/quick_run.sh <tag>
PKG=paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd
TAG=${1:-q}
mkdir -p gpurun_out/quick
cp exp/quick/libvsmpc.so $PKG/libvsmpc.so || exit 1
timeout -k 10 200 python tools/gpu_phases.py > gpurun_out/quick/${TAG}_phases.txt 2>&1 || { tail -5 gpurun_out/quick/${TAG}_phases.txt; exit 1; }
grep -v "start\|shader" gpurun_out/quick/${TAG}_phases.txt
timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency --no-extra > gpurun_out/quick/${TAG}_b256.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency --no-extra --batch 4096 --workload montecarlo --steps 40 --warmup 4 > gpurun_out/quick/${TAG}_b4096.json 2>/dev/null || exit 1
python - <<PY
import json
for n in ("b256","b4096"):
    d=json.load(open("gpurun_out/quick/${TAG}_%s.json"%n)); r=d["roofline"]
    print(n, "us/launch", round(r["kernel_us_per_launch"],2), "frac", round(r["frac"],4), "parity", d.get("parity_max_rel_err_vs_oracle"))
PY
