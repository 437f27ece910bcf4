#!/bin/bash
# Condenses gpurun_out/prof_<tag>/ (tools/prof_r04.sh) into profiles/: per-config kernel stats + summary + HBM traffic,
# MFMA counters, and the keyed databases profiles/hbm_traffic.json / mfma_counters.json bench.py reads.
#   tools/prof_r04_summarize.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-r04}
O=gpurun_out/prof_$TAG
for spec in "c1_hover256 paper:hover:256" "c2_takeoff4096 paper:takeoff:4096" "c3_montecarlo4096 paper:montecarlo:4096" "c4_h2x4096 horizon2x:hover:4096"; do
  set -- $spec
  python tools/summarize_profile.py ${TAG}_$1 $O/$1_stats $O/$1_fetch $O/$1_write --key $2 > /dev/null || echo "summary failed for $1"
  python tools/summarize_mfma.py ${TAG}_$1 gpurun_out/mfma_${TAG}_$1 --key $2 > /dev/null || echo "mfma summary failed for $1"
  cp $O/$1_bench_under_rocprof.json profiles/${TAG}_$1_bench_under_rocprof.json 2>/dev/null
done
cp $O/phases.txt profiles/${TAG}_phase_cycles.txt 2>/dev/null
cp $O/phases_h2x.txt profiles/${TAG}_phase_cycles_h2x.txt 2>/dev/null
python - "$TAG" "$O" <<'PY'
# LDS access-pattern microbenchmark (tools/microbench/lds_conflict.hip) -> profiles/<tag>_lds_patterns.md
import collections, csv, os, statistics, sys
tag, o = sys.argv[1], sys.argv[2]
f = None
for root, _, files in os.walk(os.path.join(o, "lds_micro")):
    for fn in files:
        if fn.endswith("counter_collection.csv"):
            f = os.path.join(root, fn)
if f:
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = ["b32_linear", "b64_linear", "b64_tile17  (4 ks + g) * 17 + j", "b64_tile16  (4 ks + g) * 16 + j", "b64_crow17  g * 17 + j + 68 r",
             "b64_bcast   wave-uniform address", "b128_linear"]
    lines = ["# " + tag + ": what SQ_LDS_BANK_CONFLICT counts (tools/microbench/lds_conflict.hip, 256 wavefronts x 4096 reads per lane)", "",
             "| pattern | LDS instructions | SQ_LDS_IDX_ACTIVE cycles | SQ_LDS_BANK_CONFLICT cycles | conflict cycles per instruction |", "|---|---|---|---|---|"]
    for k in sorted(vals):
        i = int(k.split("<")[1].split(">")[0])
        m = {c: statistics.median(v) for c, v in vals[k].items()}
        lines.append(f"| {names[i]} | {m['SQ_INSTS_LDS']:.0f} | {m['SQ_LDS_IDX_ACTIVE']:.0f} | {m['SQ_LDS_BANK_CONFLICT']:.0f} | {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_INSTS_LDS']:.2f} |")
    open(os.path.join("profiles", tag + "_lds_patterns.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
PY
ls profiles | grep "^${TAG}_" | head -40
