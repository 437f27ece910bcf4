"""Condenses rocprofv3 outputs under gpurun_out/ into profiles/ (tracked).

    python tools/summarize_profile.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>] [--key paper:hover:256]

Writes profiles/<tag>_kernel_stats.csv (the rocprofv3 --kernel-trace --stats summary, trimmed to the
top kernels), profiles/<tag>_summary.md and, when PMC passes are given, updates
profiles/hbm_traffic.json with the HBM bytes per launch of the solve kernel:
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024      (rocprofv3 reports KiB; MI355X_MICROARCH.md "HBM":
on gfx950 FETCH_SIZE reports half the bytes of a 16 B/lane coalesced stream, WRITE_SIZE is exact for
16 B/lane stores; the kernel's global loads and stores are 16 B per lane).  The uncorrected sum is kept too.
A large share of the fetches are instruction fetches, for which the x2 is an upper bound.
"""
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    f = sorted(glob.glob(pattern, recursive=True))
    if not f:
        raise SystemExit(f"no file matches {pattern}")
    return f[0]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    key = "paper:hover:256"
    for i, a in enumerate(sys.argv):
        if a == "--key":
            key = sys.argv[i + 1]
            args = [x for x in args if x != key]
    tag, stats_dir = args[0], args[1]
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(one(os.path.join(stats_dir, "**", "*_kernel_stats.csv")))))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows[:8]:
            r = dict(r)
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    solve = [r for r in rows if "solve_kernel" in r["Name"]][0]
    trace = list(csv.DictReader(open(one(os.path.join(stats_dir, "**", "*_kernel_trace.csv")))))
    t0 = [r for r in trace if "solve_kernel" in r["Kernel_Name"]][0]
    lines = [f"# {tag}: rocprofv3 --kernel-trace --stats", "",
             f"kernel: `{solve['Name'][:120]}`", "",
             f"- calls: {solve['Calls']}", f"- average duration: {float(solve['AverageNs'])/1e3:.2f} us "
             f"(min {float(solve['MinNs'])/1e3:.2f}, max {float(solve['MaxNs'])/1e3:.2f}, stddev {float(solve['StdDev'])/1e3:.2f})",
             f"- share of GPU time: {solve['Percentage']} %",
             f"- grid {t0['Grid_Size_X']} threads, workgroup {t0['Workgroup_Size_X']}, VGPR {t0['VGPR_Count']} + AGPR {t0['Accum_VGPR_Count']}, SGPR {t0['SGPR_Count']}"]
    if len(args) >= 4:
        vals = {}
        for name, d in (("FETCH_SIZE", args[2]), ("WRITE_SIZE", args[3])):
            rr = [r for r in csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv"))))
                  if "solve_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
            vals[name] = statistics.median(float(r["Counter_Value"]) for r in rr)
        total = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        raw = (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        lines += ["", "## HBM traffic (separate --pmc passes)", "",
                  f"- FETCH_SIZE median {vals['FETCH_SIZE']:.0f} KiB / launch", f"- WRITE_SIZE median {vals['WRITE_SIZE']:.0f} KiB / launch",
                  f"- bytes per launch = (2*FETCH+WRITE)*1024 = {total:.0f}  (uncorrected (FETCH+WRITE)*1024 = {raw:.0f})"]
        tpath = os.path.join(out_dir, "hbm_traffic.json")
        db = json.load(open(tpath)) if os.path.exists(tpath) else {}
        db[key] = {"bytes_per_launch": total, "bytes_per_launch_uncorrected": raw, "fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"], "tag": tag}
        json.dump(db, open(tpath, "w"), indent=1, sort_keys=True)
    open(os.path.join(out_dir, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
