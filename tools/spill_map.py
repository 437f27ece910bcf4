"""Where a solve kernel spills: compiles one translation unit of vsmpc_kernels.hip with --save-temps and lists every
scratch store / reload of the chosen instantiation with the nearest basic-block label in front of it (inlined function
names survive in the labels).     python tools/spill_map.py 34,14,24 [form=1] [plds=0]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd", "csrc", "vsmpc_kernels.hip")


def main():
    hz = sys.argv[1] if len(sys.argv) > 1 else "34,14,24"
    form = sys.argv[2] if len(sys.argv) > 2 else "1"
    plds = sys.argv[3] if len(sys.argv) > 3 else "0"
    tmp = os.environ.get("SPILL_TMP") or tempfile.mkdtemp(prefix="spill_")
    asm = os.path.join(tmp, "vsmpc_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(asm) or os.environ.get("SPILL_REBUILD"):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                        f"-DVS_TU_HORIZON={hz}", "-DVS_TU_STAMPS=0", "--save-temps=obj", SRC, "-o", os.path.join(tmp, "k.o")]
                       + os.environ.get("VSMPC_HIPCC_FLAGS", "").split(), cwd=tmp, check=True, capture_output=True)
    S = open(asm).read().split("\n")
    n, ns, hc = hz.split(",")
    pat = rf"^_ZN5vsmpc12solve_kernelINS_4DimsILi{n}ELi{ns}ELi{hc}EEELb0ELi{form}ELb{plds}E.*:"
    start = [i for i, l in enumerate(S) if re.match(pat, l)][0]
    end = [i for i in range(start, len(S)) if S[i].startswith(".Lfunc_end")][0]
    body = S[start:end]
    label = "entry"
    by_label_st, by_label_ld = collections.OrderedDict(), collections.OrderedDict()
    ninstr = 0
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\S+):\s*;?\s*(.*)", l)
        if m:
            label = f"{m.group(1)} {m.group(2)[:90]} @{100 * i // len(body)}%"
        if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
            ninstr += 1
        if "scratch_store" in l:
            by_label_st[label] = by_label_st.get(label, 0) + 1
        if "scratch_load" in l:
            by_label_ld[label] = by_label_ld.get(label, 0) + 1
    print("instructions", ninstr, "| spill stores", sum(by_label_st.values()), "reloads", sum(by_label_ld.values()), "| asm in", tmp)
    print("-- stores")
    for k, v in by_label_st.items():
        print(f"{v:4d}  {k}")
    print("-- reloads")
    for k, v in by_label_ld.items():
        print(f"{v:4d}  {k}")


if __name__ == "__main__":
    main()
