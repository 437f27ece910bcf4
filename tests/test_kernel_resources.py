"""Register allocation of the shipped solve kernels, read from the code-object metadata of the current build
(tools/kernel_resources.py: llvm-readelf --notes on the unbundled gfx950 code object).  A spilled vector register is a
scratch (= global memory) round trip on the serial instruction stream of a latency-bound workgroup, for EVERY instance:
round 3's 2x-horizon kernel shipped with 107 of them (432 B per lane, 12.6x the algorithmic HBM traffic).  The production
instantiations of BASELINE.json's two horizons -- structured condensing, the form every benchmark line runs -- must have
none, and no scratch segment at all.  (`sgpr_spill_count` counts scalar values parked in lanes of a vector register with
v_writelane / v_readlane: no memory traffic; bounded here so that it cannot creep up unnoticed.)"""
import importlib
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"


@pytest.fixture(scope="module")
def kernels(solver_mod):
    ks = kr.all_kernels()
    assert ks, "no code objects under <pkg>/build: build first (python __graft_entry__.py)"
    return ks


def production(kernels, n, ns, hc):
    """solve_kernel<Dims<n, ns, hc>, STAMPS = false, FORM = 1, PLDS = *>"""
    pat = re.compile(rf"solve_kernel<.*Dims<{n}, {ns}, {hc}>\s*,\s*false,\s*1,\s*(true|false)>")
    return {k: v for k, v in kernels.items() if pat.search(k)}


@pytest.mark.parametrize("horizon, sgpr_spill_bound", [((17, 7, 12), 260), ((34, 14, 24), 500)])
def test_production_solve_kernels_do_not_spill(kernels, horizon, sgpr_spill_bound):
    prod = production(kernels, *horizon)
    assert len(prod) >= 1, list(kernels)
    for name, r in prod.items():
        assert r["vgpr_spill_count"] == 0, (name, r)
        assert r["private_segment_fixed_size"] == 0, (name, r)          # no scratch segment at all
        assert r["sgpr_spill_count"] <= sgpr_spill_bound, (name, r)
        assert r["max_flat_workgroup_size"] == 256


def test_two_workgroups_per_cu_at_the_paper_horizon(kernels):
    """<= 256 registers per lane (unified file of 512 per SIMD lane) is what lets two workgroups share a CU"""
    for name, r in production(kernels, 17, 7, 12).items():
        assert r["vgpr_count"] + r.get("agpr_count", 0) <= 256, (name, r)
