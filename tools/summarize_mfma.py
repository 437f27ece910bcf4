#!/usr/bin/env python3
"""Condenses the --pmc passes of tools/prof_mfma.sh into profiles/<tag>_mfma_counters.{json,md} and
profiles/mfma_counters.json (keyed like hbm_traffic.json; bench.py copies `mfma_busy_frac` into its roofline object).

    python tools/summarize_mfma.py <tag> gpurun_out/mfma_<tag> [--key paper:hover:256]

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (64 per
v_mfma_f64_16x16x4_f64); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_BUSY_CYCLES is summed over
the 32 shader engines; SQ_INSTS_VALU_MFMA_MOPS_F64 counts operations / 512.
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024 SIMDs): the share of all SIMD-cycles of the
launch in which the matrix pipe was executing.
Round 4 (passes d, e of tools/prof_mfma.sh): the FP64 vector instructions by kind -- executed FLOPs = 512 x MOPS_F64 (matrix
cores) + 64 lanes x (ADD + MUL + 2 FMA + TRANS) (vector unit; an upper bound: masked-off lanes are counted) -- and the LDS
conflict counters (SQ_LDS_BANK_CONFLICT and SQ_LDS_IDX_ACTIVE count cycles, SQ_ACTIVE_INST_LDS quad-cycles: round 3 compared
the first with the last and read "89 %"; the share of LDS cycles lost to bank conflicts is conflict / idx_active).
`bound`: "mfma" if the matrix pipe is busy more than half of the SIMD-cycles, "hbm" never here (arithmetic intensity 437
FLOP/B), else "issue": the wavefronts are parked at s_waitcnt / s_barrier or issuing dependent instructions one at a time."""
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
    if "SQ_INSTS_VALU_FMA_F64" in med:
        vi = {k: med.get("SQ_INSTS_VALU_" + k + "_F64", 0.0) / inst for k in ("ADD", "MUL", "FMA", "TRANS")}
        out["valu_f64_instructions_per_instance"] = vi
        out["valu_f64_flops_per_instance"] = 64.0 * (vi["ADD"] + vi["MUL"] + 2.0 * vi["FMA"] + vi["TRANS"])
        out["executed_flops_per_instance"] = out["mfma_flops_per_instance"] + out["valu_f64_flops_per_instance"]
        out["instructions_per_instance"] = {"valu": med.get("SQ_INSTS_VALU", 0.0) / inst, "salu": med.get("SQ_INSTS_SALU", 0.0) / inst,
                                            "lds": med.get("SQ_INSTS_LDS", 0.0) / inst, "smem": med.get("SQ_INSTS_SMEM", 0.0) / inst,
                                            "mfma": med["SQ_INSTS_MFMA"] / inst, "valu_int32": med.get("SQ_INSTS_VALU_INT32", 0.0) / inst}
    if "SQ_LDS_IDX_ACTIVE" in med:
        out["lds_bank_conflict_share_of_lds_cycles"] = med["SQ_LDS_BANK_CONFLICT"] / med["SQ_LDS_IDX_ACTIVE"]
        out["lds_bank_conflict_cycles_per_instance"] = med["SQ_LDS_BANK_CONFLICT"] / inst
    out["bound"] = "mfma" if out["mfma_busy_frac"] > 0.5 else "issue"
    pdir = os.path.join(ROOT, "profiles")
    json.dump(out, open(os.path.join(pdir, f"{tag}_mfma_counters.json"), "w"), indent=1, sort_keys=True)
    dbp = os.path.join(pdir, "mfma_counters.json")
    db = json.load(open(dbp)) if os.path.exists(dbp) else {}
    db[key] = {"tag": tag, "mfma_busy_frac": out["mfma_busy_frac"], "mfma_flops_per_instance": out["mfma_flops_per_instance"],
               "mfma_instructions_per_instance": out["mfma_instructions_per_instance"],
               "executed_flops_per_instance": out.get("executed_flops_per_instance"),
               "valu_f64_flops_per_instance": out.get("valu_f64_flops_per_instance"), "bound": out["bound"],
               "wave_time_shares": out.get("wave_time_shares")}
    json.dump(db, open(dbp, "w"), indent=1, sort_keys=True)
    lines = [f"# {tag}: matrix-core utilisation by counter (rocprofv3 --pmc, three separate passes)", "",
             f"kernel `{meta['kernel']}`, {int(inst)} instances per launch, workgroup {meta['workgroup']}, "
             f"VGPR {meta['vgpr']} + AGPR {meta['agpr']}", "",
             f"- SQ_INSTS_MFMA {med['SQ_INSTS_MFMA']:.0f} = {out['mfma_instructions_per_instance']:.0f} per instance; "
             f"SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 = {out['mfma_flops_per_instance']/1e6:.2f} MFLOP executed per instance "
             f"(F_alg = {'6.17 MFLOP at the 2x horizon' if 'Dims<34' in meta['kernel'] else '3.086 MFLOP at the paper horizon'})",
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
                  f"{med.get('SQ_LDS_BANK_CONFLICT', 0):.0f} cycles, SQ_ACTIVE_INST_LDS {med.get('SQ_ACTIVE_INST_LDS', 0):.0f} quad-cycles"]
    if "executed_flops_per_instance" in out:
        vi, ii = out["valu_f64_instructions_per_instance"], out["instructions_per_instance"]
        lines += [f"- FP64 vector instructions per instance (wave-level): ADD {vi['ADD']:.0f}, MUL {vi['MUL']:.0f}, FMA {vi['FMA']:.0f}, "
                  f"TRANS {vi['TRANS']:.0f} -> {out['valu_f64_flops_per_instance']/1e6:.2f} MFLOP (64 lanes each, masked lanes included); "
                  f"**executed {out['executed_flops_per_instance']/1e6:.2f} MFLOP per instance** (matrix cores + vector unit)",
                  f"- instructions per instance: VALU {ii['valu']:.0f} (INT32 {ii['valu_int32']:.0f}), SALU {ii['salu']:.0f}, LDS {ii['lds']:.0f}, "
                  f"SMEM {ii['smem']:.0f}, MFMA {ii['mfma']:.0f}"]
    if "lds_bank_conflict_share_of_lds_cycles" in out:
        lines += [f"- LDS: {out['lds_bank_conflict_cycles_per_instance']:.0f} bank-conflict cycles per instance = "
                  f"{100*out['lds_bank_conflict_share_of_lds_cycles']:.0f} % of the LDS index-active cycles (SQ_LDS_IDX_ACTIVE "
                  f"{med['SQ_LDS_IDX_ACTIVE']:.0f})"]
    lines += [f"- **bound: {out['bound']}** (matrix pipe busy {100*out['mfma_busy_frac']:.0f} % of the SIMD-cycles)"]
    open(os.path.join(pdir, f"{tag}_mfma_counters.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
