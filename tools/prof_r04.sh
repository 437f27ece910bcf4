#!/bin/bash
# Counter evidence for ALL FOUR benchmark configurations of one kernel version (run on the GPU box through gpurun):
#   tools/prof_r04.sh <tag>            e.g. tools/prof_r04.sh r04_v24
# Round 4: + the FP64 vector-instruction counters (executed FLOPs next to the algorithmic ones) and the LDS conflict
# counters, as passes d and e of tools/prof_mfma.sh; + the LDS access-pattern microbenchmark under the same counters.
# Per configuration (BASELINE configs[1..4]): rocprofv3 --kernel-trace --stats, separate --pmc FETCH_SIZE / WRITE_SIZE
# passes, the three MFMA counter passes of tools/prof_mfma.sh -- the program directly after `--`, counters never combined
# with other trace domains.  tools/prof_r04_summarize.sh condenses gpurun_out/prof_<tag>/ into profiles/ afterwards (here).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r04}
O=gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
B="python3 bench.py --no-cpu-baseline --no-latency --no-extra"
step() { "$@"; local rc=$?; echo "rc=$rc: $*" >> "$O/steps.txt"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then cat "$O/steps.txt"; exit $rc; fi; }
run_cfg() {   # name, steps, warmup, bench args...
  local name=$1 steps=$2 warm=$3; shift 3
  step timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$O/${name}_stats" -o s --output-format csv -- $B --steps $steps --warmup $warm "$@" > "$O/${name}_bench_under_rocprof.json" 2> "$O/${name}_stats.err"
  step timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/${name}_fetch" -o f --output-format csv -- $B --steps 10 --warmup 2 "$@" > /dev/null 2> "$O/${name}_fetch.err"
  step timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/${name}_write" -o w --output-format csv -- $B --steps 10 --warmup 2 "$@" > /dev/null 2> "$O/${name}_write.err"
  step bash tools/prof_mfma.sh ${TAG}_${name} "$@"
  echo "done $name" >> "$O/progress.txt"
}
run_cfg c1_hover256 200 20
run_cfg c2_takeoff4096 40 4 --batch 4096 --workload takeoff
run_cfg c3_montecarlo4096 40 4 --batch 4096 --workload montecarlo
run_cfg c4_h2x4096 12 2 --config horizon2x --batch 4096 --workload hover
hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_conflict.hip -o "$O/lds_conflict" 2> /dev/null
step timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d "$O/lds_micro" -o m --output-format csv -- "$O/lds_conflict" > "$O/lds_micro.out" 2> "$O/lds_micro.err"
step python3 tools/gpu_phases.py > "$O/phases.txt" 2>/dev/null
step python3 tools/gpu_phases.py h2x > "$O/phases_h2x.txt" 2>/dev/null
cat "$O/steps.txt"
exit 0
