"""csrc/vsmpc_panel_asm.inc is generated (tools/gen_panel_asm.py): the committed file must be what the generator writes, the
hazards gfx940 does not interlock must be padded, and every stream must fit the operand limit of an asm statement."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
INC = os.path.join(ROOT, PKG, "csrc", "vsmpc_panel_asm.inc")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_panel_asm", os.path.join(ROOT, "tools", "gen_panel_asm.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_streams_are_what_the_generator_writes(tmp_path):
    g = _gen()
    g.OUT = str(tmp_path / "out.inc")
    g.main()
    assert open(g.OUT).read() == open(INC).read(), "run python tools/gen_panel_asm.py"


def test_wait_states_of_the_hazards_the_hardware_does_not_interlock():
    """VALU write -> DPP read of that register: 2 wait states; transcendental result -> use: 1 (tools/gen_panel_asm.py)."""
    txt = open(INC).read()
    for body in re.findall(r"asm volatile\(\n(.*?)\n        :", txt, flags=re.S):
        lines = [l.strip().strip('"').replace("\\n\\t", "") for l in body.split("\n")]
        written = {}   # register -> (index, is_trans)
        slot = 0
        for l in lines:
            if l.startswith("s_nop"):
                slot += int(l.split()[1]) + 1
                continue
            regs = re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", l)
            ops = []
            for a, b, c in regs:
                ops.append(list(range(int(a), int(b) + 1)) if a else [int(c)])
            if l.startswith("v_") and ops:
                dst, srcs = ops[0], ops[1:]
                if "row_newbcast" in l and srcs:       # the DPP operand is src0
                    for r in srcs[0]:
                        if r in written:
                            assert slot - written[r][0] - 1 >= 2, l
                for sr in srcs:
                    for r in sr:
                        if r in written and written[r][1]:
                            assert slot - written[r][0] - 1 >= 1, l
                for r in dst:
                    written[r] = (slot, l.startswith("v_rsq"))
            elif l.startswith("ds_read") and ops:
                for r in ops[0]:
                    written.pop(r, None)
            slot += 1


def test_operand_counts_fit_an_asm_statement():
    txt = open(INC).read()
    for m in re.finditer(r"\n        : (.*?)\n        : (.*?)\n        : ", txt):
        n = m.group(1).count("(") + m.group(2).count("(")
        assert n <= 30, n
