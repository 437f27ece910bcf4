#!/bin/bash
# Experiment build with ONE horizon (fast: ~40 s instead of 2.5 min): tools/quick_build.sh [N,NS,HC] [extra hipcc flags]
# Writes $Q/libvsmpc.so (or $QUICK_OUT/libvsmpc.so); the tracked csrc/vsmpc_horizons.def is left alone (a copy of
# the sources is compiled).
set -e
cd "$(dirname "$0")/.."
PKG=paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd
H=${1:-17,7,12}; shift || true
Q=${QUICK_OUT:-exp/quick}
rm -rf $Q/src && mkdir -p $Q/src/$PKG/csrc $Q/src/include
cp $PKG/csrc/*.hip $PKG/csrc/*.hpp $PKG/csrc/*.inc $Q/src/$PKG/csrc/
cp include/*.h $Q/src/include/
echo "X($(echo $H | sed 's/,/, /g'))" > $Q/src/$PKG/csrc/vsmpc_horizons.def
S=$Q/src/$PKG/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wl,-rpath,/opt/rocm/lib "$@" \
    -o $Q/libvsmpc.so $S/vsmpc_kernels.hip $S/vsmpc_rollout.hip $S/vsmpc_capi.hip $S/vsmpc_jet.hip $S/vsmpc_provider.hip
echo "built $Q/libvsmpc.so for horizon $H"
