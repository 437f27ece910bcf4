#!/bin/bash
# Full profile set of one kernel version (run on the GPU box through gpurun):
#   tools/prof_r02.sh <tag>
# kernel-trace stats at the headline config (batch 256) and at batch 4096 (configs[3] slice), separate --pmc passes for
# FETCH_SIZE / WRITE_SIZE, MFMA counters at both batch sizes, phase cycles.  Condensed by tools/summarize_profile.py and
# tools/summarize_mfma.py into profiles/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r02}
O=gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
B="python3 bench.py --no-cpu-baseline --no-latency --no-extra"
step() { "$@"; local rc=$?; echo "rc=$rc: $*" >> "$O/steps.txt"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$O/stats256" -o s --output-format csv -- $B > "$O/bench256_under_rocprof.json" 2> "$O/stats256.err"
step timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$O/stats4096" -o s --output-format csv -- $B --batch 4096 --workload montecarlo --steps 40 --warmup 4 > "$O/bench4096_under_rocprof.json" 2> "$O/stats4096.err"
step timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/fetch256" -o f --output-format csv -- $B --steps 20 --warmup 5 > /dev/null 2> "$O/fetch256.err"
step timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/write256" -o w --output-format csv -- $B --steps 20 --warmup 5 > /dev/null 2> "$O/write256.err"
step timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/fetch4096" -o f --output-format csv -- $B --batch 4096 --workload montecarlo --steps 10 --warmup 2 > /dev/null 2> "$O/fetch4096.err"
step timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/write4096" -o w --output-format csv -- $B --batch 4096 --workload montecarlo --steps 10 --warmup 2 > /dev/null 2> "$O/write4096.err"
step bash tools/prof_mfma.sh ${TAG}_b256
step bash tools/prof_mfma.sh ${TAG}_b4096 --batch 4096 --workload montecarlo
step python3 tools/gpu_phases.py > "$O/phases.txt" 2>/dev/null
step python3 tools/gpu_phases.py h2x > "$O/phases_h2x.txt" 2>/dev/null
cat "$O/steps.txt"
exit 0
