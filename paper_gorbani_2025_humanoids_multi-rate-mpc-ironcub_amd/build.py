"""Builds the HIP library in-tree: csrc/*.hip -> build/*.o -> libvsmpc.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libvsmpc.so")
SOURCES = ["vsmpc_kernels.hip", "vsmpc_rollout.hip", "vsmpc_capi.hip", "vsmpc_jet.hip", "vsmpc_provider.hip"]
HEADERS = ["vsmpc_device.hpp", "vsmpc_launch.hpp", "vsmpc_panel_asm.inc", "vsmpc_jet_device.hpp", "vsmpc_horizons.def", os.path.join("..", "..", "include", "vsmpc.h"),
           os.path.join("..", "..", "include", "vsmpc_jet.h")]


# Horizons (nIter, nIterSmall, controlHorizon) that get a kernel instantiation.  The kernels are straight-line code per
# horizon (variableSamplingMPC.cpp:24-45 sizes the reference from its XML at run time; here the table is fixed at build
# time): BASELINE.json's two configurations, plus one odd size that exercises the general paths (controlHorizon odd:
# joint rows share a tile with throttle rows; three throttle tile rows).  VSMPC_HORIZONS="17,7,12;25,10,18" overrides.
DEFAULT_HORIZONS = ((17, 7, 12), (34, 14, 24), (21, 9, 15))
HORIZONS_DEF = os.path.join(CSRC, "vsmpc_horizons.def")


def horizons():
    env = os.environ.get("VSMPC_HORIZONS", "").strip()
    if not env:
        return DEFAULT_HORIZONS
    out = []
    for item in env.split(";"):
        n, ns, hc = (int(v) for v in item.split(","))
        out.append((n, ns, hc))
    return tuple(out)


def write_horizons_def() -> bool:
    """(Re)writes csrc/vsmpc_horizons.def when the requested table differs; returns True if it changed."""
    text = ("// Horizons (nIter, nIterSmall, controlHorizon) with a kernel instantiation: one X(...) line each.\n"
            "// Written by build.py (DEFAULT_HORIZONS or VSMPC_HORIZONS); the first two are BASELINE.json's configurations.\n"
            + "".join(f"X({n}, {ns}, {hc})\n" for n, ns, hc in horizons()))
    old = open(HORIZONS_DEF).read() if os.path.exists(HORIZONS_DEF) else None
    if old != text:
        with open(HORIZONS_DEF, "w") as f:
            f.write(text)
        return True
    return False


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the vsmpc HIP library cannot be built")


OBJ_DIR = os.path.join(HERE, "build")


SCHED_MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def _units():
    """(object name, source, extra flags): vsmpc_kernels.hip is compiled once for its common part and twice per horizon
    (production / diagnostic instantiations, see the note on translation units in the file), everything else once."""
    units = [("kernels_common", "vsmpc_kernels.hip", ["-DVS_TU_COMMON"])]
    for n, ns, hc in horizons():
        # Machine scheduler strategy per horizon (measured on MI355X, default against -amdgpu-sched-strategy=max-ilp): the
        # short-horizon kernels (<= 256 registers, two workgroups per CU) gain 2-3 % with max-ilp -- 36.6 -> 35.7 us per 256-launch,
        # 327 -> 317 us per 4096 -- the long-horizon kernel (512 registers, tiles in AGPRs) loses 2.6 % (1,593 -> 1,634 us).
        sched = SCHED_MAX_ILP if n <= 24 else []
        for st in (0, 1):
            units.append((f"kernels_{n}_{ns}_{hc}_{'diag' if st else 'prod'}", "vsmpc_kernels.hip",
                          [f"-DVS_TU_HORIZON={n},{ns},{hc}", f"-DVS_TU_STAMPS={st}"] + sched))
    for src in SOURCES:
        if src != "vsmpc_kernels.hip":
            units.append((os.path.splitext(src)[0], src, []))
    return units


def _deps(src):
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]


def _extra_flags():
    """VSMPC_HIPCC_FLAGS: measurement builds (tools/exp_build.sh), e.g. -DVS_DIAG_SPLIT"""
    return os.environ.get("VSMPC_HIPCC_FLAGS", "").split()


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    """Objects under build/ (git-ignored), one hipcc process each, `jobs` at a time (default: the CPUs of the machine, at
    most 8); an object is rebuilt when its source, a header or this file is newer, or when its flags changed."""
    from concurrent.futures import ThreadPoolExecutor
    changed = write_horizons_def()
    if not force and not changed and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + _extra_flags()
    if verbose:
        base.insert(1, "-Rpass-analysis=kernel-resource-usage")
    todo, objs = [], []
    for name, src, flags in _units():
        obj = os.path.join(OBJ_DIR, name + ".o")
        stamp = obj + ".flags"
        cmd = base + flags + [os.path.join(CSRC, src), "-o", obj]
        objs.append(obj)
        flagtext = " ".join(cmd)
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == flagtext
                 and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps(src)))
        if not fresh:
            todo.append((cmd, stamp, flagtext))

    def run(item):
        cmd, stamp, flagtext = item
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode == 0:
            with open(stamp, "w") as f:
                f.write(flagtext)
        return cmd, res

    jobs = jobs or int(os.environ.get("VSMPC_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        results = list(pool.map(run, todo))
    for cmd, res in results:
        if res.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + res.stdout + res.stderr)
        if verbose:
            print(res.stderr)
    keep = {os.path.basename(o) for o in objs}
    for f in os.listdir(OBJ_DIR):           # objects of horizons that left the table
        if f.endswith(".o") and f not in keep:
            os.remove(os.path.join(OBJ_DIR, f))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-o", LIB_PATH] + objs
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


def build_bindings(force: bool = False) -> str | None:
    """pybind11 shim `bindingsMPC` (reference Python surface, MPCPyBindings.cpp:12-91); host-only g++ build that links
    libvsmpc.so.  Returns the module path, or None when pybind11 is not available."""
    try:
        import pybind11
    except ImportError:
        return None
    import sysconfig
    src = os.path.join(CSRC, "bindings_mpc.cpp")
    out = os.path.join(HERE, "bindingsMPC" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    deps = [src, os.path.join(HERE, "..", "include", "VariableSamplingMPC.hpp"), os.path.join(HERE, "..", "include", "vsmpc.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    build_library()
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", f"-I{pybind11.get_include()}",
           f"-I{sysconfig.get_paths()['include']}", src, "-o", out, f"-L{HERE}", "-lvsmpc",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("bindingsMPC build failed:\n" + res.stdout + res.stderr)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
