"""The generated assembly panel stream (csrc/vsmpc_panel_asm.inc <- tools/gen_panel_asm.py) against the C++ stream it replaces:
tools/microbench/panel_probe.hip factors the same panel with both and compares L and 1 / L_jj bit for bit."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"


@pytest.mark.gpu
def test_dpp_panel_stream_is_bit_identical_to_the_cpp_stream(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "panel_probe")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, PKG, "csrc"), "-o", exe,
                    os.path.join(ROOT, "tools", "microbench", "panel_probe.hip")], check=True, capture_output=True, text=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "panel_probe.txt"), "w") as f:
            f.write(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "bit-identical to the C++ stream" in res.stdout, res.stdout
