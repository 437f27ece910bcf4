#!/bin/bash
# On the GPU box: 2x-horizon phase cycles for every experiment library exp/NAME (arguments).
PKG=paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd
mkdir -p gpurun_out/exp
for name in "$@"; do
  cp exp/$name/libvsmpc.so $PKG/libvsmpc.so || exit 1
  timeout -k 10 200 python tools/gpu_phases.py h2x > gpurun_out/exp/${name}_h2x.txt 2>&1 || exit 1
  echo "== $name"; sed -n 2,16p gpurun_out/exp/${name}_h2x.txt | grep -E "total|P1|P3|P4b|P5"
done
