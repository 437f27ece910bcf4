#!/usr/bin/env python3
"""Condenses the --pmc passes of tools/prof_mfma.sh into profiles/<tag>_mfma_counters.{json,md} and
profiles/mfma_counters.json (keyed like hbm_traffic.json; bench.py copies `mfma_busy_frac` into its roofline object).

    python tools/summarize_mfma.py <tag> gpurun_out/mfma_<tag> [--key paper:hover:256]

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (64 per
v_mfma_f64_16x16x4_f64); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_BUSY_CYCLES is summed over
the 32 shader engines; SQ_INSTS_VALU_MFMA_MOPS_F64 counts operations / 512.
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024 SIMDs): the share of all SIMD-cycles of the
launch in which the matrix pipe was executing."""
import collections
import csv
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SE, N_SIMD = 32, 1024


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    key = "paper:hover:256"
    if "--key" in sys.argv:
        key = sys.argv[sys.argv.index("--key") + 1]
        args = [a for a in args if a != key]
    tag, d = args[0], args[1]
    med, meta = {}, {}
    for p in sorted(os.listdir(d)):
        f = os.path.join(d, p, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {"kernel": r["Kernel_Name"][:90], "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]),
                        "lds_block_size": int(r["LDS_Block_Size"]), "vgpr": int(r["VGPR_Count"]),
                        "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"])}
        for k, v in vals.items():
            med[k] = statistics.median(v)
    waves = med.get("SQ_WAVES", meta["grid"] / 64)
    inst = meta["grid"] / meta["workgroup"]
    out = {"tag": tag, **meta, "instances_per_launch": inst, "counters_median_per_launch": med}
    busy_cyc = med["SQ_BUSY_CYCLES"] / N_SE
    out["kernel_busy_cycles"] = busy_cyc
    out["mfma_busy_frac"] = med["SQ_VALU_MFMA_BUSY_CYCLES"] / (busy_cyc * N_SIMD)
    out["mfma_instructions_per_instance"] = med["SQ_INSTS_MFMA"] / inst
    out["mfma_flops_per_instance"] = med["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / inst
    out["wave_cycles_mean"] = 4 * med["SQ_WAVE_CYCLES"] / waves
    if "SQ_WAIT_ANY" in med:
        wc = med["SQ_WAVE_CYCLES"]
        out["wave_time_shares"] = {"parked_waitcnt_or_barrier": med["SQ_WAIT_ANY"] / wc,
                                   "issue_stall": med["SQ_WAIT_INST_ANY"] / wc,
                                   "issuing": med["SQ_ACTIVE_INST_ANY"] / wc,
                                   "valu_active": med["SQ_ACTIVE_INST_VALU"] / wc}
        out["valu_instructions_per_wave"] = med["SQ_INSTS_VALU"] / waves
    pdir = os.path.join(ROOT, "profiles")
    json.dump(out, open(os.path.join(pdir, f"{tag}_mfma_counters.json"), "w"), indent=1, sort_keys=True)
    dbp = os.path.join(pdir, "mfma_counters.json")
    db = json.load(open(dbp)) if os.path.exists(dbp) else {}
    db[key] = {"tag": tag, "mfma_busy_frac": out["mfma_busy_frac"], "mfma_flops_per_instance": out["mfma_flops_per_instance"],
               "mfma_instructions_per_instance": out["mfma_instructions_per_instance"]}
    json.dump(db, open(dbp, "w"), indent=1, sort_keys=True)
    lines = [f"# {tag}: matrix-core utilisation by counter (rocprofv3 --pmc, three separate passes)", "",
             f"kernel `{meta['kernel']}`, {int(inst)} instances per launch, workgroup {meta['workgroup']}, "
             f"VGPR {meta['vgpr']} + AGPR {meta['agpr']}", "",
             f"- SQ_INSTS_MFMA {med['SQ_INSTS_MFMA']:.0f} = {out['mfma_instructions_per_instance']:.0f} per instance; "
             f"SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 = {out['mfma_flops_per_instance']/1e6:.2f} MFLOP executed per instance "
             f"(F_alg = 3.086 MFLOP at the paper horizon)",
             f"- SQ_VALU_MFMA_BUSY_CYCLES {med['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} (= 64 per instruction); kernel busy "
             f"{busy_cyc:.0f} cycles (SQ_BUSY_CYCLES / 32 SEs)",
             f"- **mfma_busy_frac = {out['mfma_busy_frac']:.3f}** of all SIMD-cycles of the launch",
             f"- mean wave lifetime {out['wave_cycles_mean']:.0f} cycles"]
    if "wave_time_shares" in out:
        s = out["wave_time_shares"]
        lines += [f"- wave time: parked at s_waitcnt / s_barrier {100*s['parked_waitcnt_or_barrier']:.0f} %, issue-stalled "
                  f"{100*s['issue_stall']:.0f} %, issuing {100*s['issuing']:.0f} % (VALU active {100*s['valu_active']:.0f} %); "
                  f"{out['valu_instructions_per_wave']:.0f} VALU instructions per wave"]
    if "SQ_VALU_MFMA_COEXEC_CYCLES" in med:
        lines += [f"- SQ_VALU_MFMA_COEXEC_CYCLES {med['SQ_VALU_MFMA_COEXEC_CYCLES']:.0f}; SQ_LDS_BANK_CONFLICT "
                  f"{med.get('SQ_LDS_BANK_CONFLICT', 0):.0f} of SQ_ACTIVE_INST_LDS {med.get('SQ_ACTIVE_INST_LDS', 0):.0f}"]
    open(os.path.join(pdir, f"{tag}_mfma_counters.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
