#!/bin/bash
# Condenses gpurun_out/prof_<tag>/ (tools/prof_r03.sh) into profiles/: per-config kernel stats + summary + HBM traffic,
# MFMA counters, and the keyed databases profiles/hbm_traffic.json / mfma_counters.json bench.py reads.
#   tools/prof_r03_summarize.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-r03}
O=gpurun_out/prof_$TAG
for spec in "c1_hover256 paper:hover:256" "c2_takeoff4096 paper:takeoff:4096" "c3_montecarlo4096 paper:montecarlo:4096" "c4_h2x4096 horizon2x:hover:4096"; do
  set -- $spec
  python tools/summarize_profile.py ${TAG}_$1 $O/$1_stats $O/$1_fetch $O/$1_write --key $2 > /dev/null || echo "summary failed for $1"
  python tools/summarize_mfma.py ${TAG}_$1 gpurun_out/mfma_${TAG}_$1 --key $2 > /dev/null || echo "mfma summary failed for $1"
  cp $O/$1_bench_under_rocprof.json profiles/${TAG}_$1_bench_under_rocprof.json 2>/dev/null
done
cp $O/phases.txt profiles/${TAG}_phase_cycles.txt 2>/dev/null
cp $O/phases_h2x.txt profiles/${TAG}_phase_cycles_h2x.txt 2>/dev/null
ls profiles | grep "^${TAG}_" | head -40
