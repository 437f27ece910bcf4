"""Register / scratch / LDS figures of every kernel in a HIP object (the .o files under <pkg>/build/, or the library),
read from the AMDGPU code-object metadata (llvm-readelf --notes of the unbundled gfx950 code object).

    python tools/kernel_resources.py [path ...]        default: all objects of the current build

Used by tests/test_kernel_resources.py (no spilled register in a production instantiation) and for the profiles."""
from __future__ import annotations

import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd")
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def code_object(path: str, out: str) -> bool:
    """the gfx950 code object of a HIP host object file or shared library: section .hip_fatbin, unbundled"""
    fat = out + ".fatbin"
    res = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", path, out + ".host"],
                         capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(fat):
        return False
    res = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={fat}",
                          f"--output={out}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
    return res.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0


def kernels_of(path: str) -> dict:
    with tempfile.TemporaryDirectory() as tmp:
        co = os.path.join(tmp, "co")
        if not code_object(path, co):
            return {}
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out = {}
    for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        block = ".agpr_count:" + block
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        rec = {}
        for f in FIELDS:
            m = re.search(rf"\.{f}:\s+(\d+)", block)
            if m:
                rec[f] = int(m.group(1))
        sym = name.group(1)
        try:
            dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip() or sym
        except FileNotFoundError:
            dem = sym
        out[dem] = rec
    return out


def all_kernels(paths=None) -> dict:
    paths = paths or sorted(glob.glob(os.path.join(PKG, "build", "*.o")))
    out = {}
    for p in paths:
        for k, v in kernels_of(p).items():
            out[k] = dict(v, object=os.path.basename(p))
    return out


if __name__ == "__main__":
    ks = all_kernels(sys.argv[1:] or None)
    w = max((len(k) for k in ks), default=10)
    for k in sorted(ks):
        v = ks[k]
        short = re.sub(r"vsmpc::", "", k)
        print(f"{short[:110]:110s} vgpr {v.get('vgpr_count', 0):3d} agpr {v.get('agpr_count', 0):3d} sgpr {v.get('sgpr_count', 0):3d} "
              f"spill v {v.get('vgpr_spill_count', 0):4d} s {v.get('sgpr_spill_count', 0):4d} scratch {v.get('private_segment_fixed_size', 0):5d} B "
              f"lds {v.get('group_segment_fixed_size', 0):6d}")
