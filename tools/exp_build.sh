#!/bin/bash
# Experiment builds of the library: tools/exp_build.sh NAME "-DVS_X=1 ..." [NAME2 "flags2" ...] -> exp/NAME/libvsmpc.so
# (built in parallel; on the GPU box: cp exp/NAME/libvsmpc.so <package>/libvsmpc.so before a measurement)
set -e
cd "$(dirname "$0")/.."
PKG=paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p exp/$name
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wl,-rpath,/opt/rocm/lib $flags \
      -o exp/$name/libvsmpc.so $PKG/csrc/vsmpc_kernels.hip $PKG/csrc/vsmpc_rollout.hip $PKG/csrc/vsmpc_capi.hip $PKG/csrc/vsmpc_jet.hip $PKG/csrc/vsmpc_provider.hip \
      > exp/$name/build.log 2>&1 && echo "built $name" || echo "FAILED $name" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
