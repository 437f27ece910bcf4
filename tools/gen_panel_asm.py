#!/usr/bin/env python3
"""Generates csrc/vsmpc_panel_asm.inc: the 16-pivot panel streams of P3 as hand-scheduled blocks of gfx950 assembly.

Why.  A panel stream is ~550 vector instructions in ONE wavefront, and a lone wavefront issues one FP64 vector instruction
every ~5.5 cycles whatever the dependencies look like (tools/microbench/lat_probe.hip, f64_issue.hip): the stream costs
its instruction count.  In the C++ form (panel_factor<1, 16> of vsmpc_kernels.hip) two thirds of the instructions are
v_readlane pairs: every entry l_cj of the pivot column is moved to a scalar register pair so that the FMA that updates
column c can read it.  gfx90a+ has a cheaper broadcast for FP64: DPP `row_newbcast:c` on v_fmac_f64 -- lane c of every row
of 16 lanes feeds all 16 lanes of that row, inside the FMA.  That needs the broadcast source in every 16-lane row, so:

Layout ("dpp" stream).  Lane L = 16 r + c carries
    a[0..15]   panel row L of the 64 rows below the diagonal tile this wavefront owns     (its own LDS row)
    g[0..15]   row c of the DIAGONAL tile                                                (the same in all four lane rows)
The diagonal tile is factored redundantly in g (four copies, bit-identical), and per pivot j and column c > j
    g[c] += (-l_cj) * g_j       a[c] += (-l_cj) * a_j        with l_cj = lane c of the scaled pivot column g_j, by DPP.
Two instructions per column instead of three, no scalar round trips, and 64 panel rows per stream instead of 48 (the C++
SPLIT form keeps the diagonal tile in lanes 0..15).  448 instructions + 48 LDS accesses against 568 + 32.
The product (-l_cj) * l_j and the accumulation order are those of the C++ form: results are bit-identical.

Also generated: the "readlane" stream (the C++ algorithm, only re-ordered), kept for the microbenchmark's comparison.

Scheduling.  Dependencies (RAW / WAR / WAW) are derived from each instruction's register reads and writes in program order;
a list scheduler with a small in-order issue model (issue cost and result latency per class) orders them, longest remaining
path first; a last pass pads the wait states gfx940 does NOT interlock (LLVM's GCNHazardRecognizer does not see inside
inline assembly): VALU write -> DPP read of that VGPR 2, transcendental result -> VALU use 1, VALU write -> v_readlane
read 1, VALU-written SGPR -> VALU read 2.
The tile lives in fixed physical registers because inline-assembly operands cannot name the halves of a 64-bit register;
the output operands are bound to exactly those registers ("={v[100:101]}"), so no copies are made.

    python tools/gen_panel_asm.py            # rewrites the .inc (committed; build.py does not run this)
"""
from __future__ import annotations

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd", "csrc", "vsmpc_panel_asm.inc")

NP = 16
# (issue cycles, cycles until the result can be consumed without a stall) per class: lone wavefront
COST = {"bar": (16, 16), "fma": (6, 10), "mul": (6, 10), "min": (6, 10), "dpp": (6, 12), "movdpp": (6, 12), "rl": (5, 34), "rsq": (10, 28),
        "cmp": (4, 8), "cnd": (4, 6), "mov": (4, 6), "smov": (2, 4), "dsr": (5, 70), "dsw": (5, 5), "wait": (2, 2)}


def vp(base):
    return f"v[{base}:{base + 1}]"


def sp(base):
    return f"s[{base}:{base + 1}]"


def regs(prefix, base, n=2):
    return [f"{prefix}{base + i}" for i in range(n)]


class Node:
    def __init__(self, kind, text, rd, wr, dpp_src=()):
        self.kind, self.text, self.rd, self.wr, self.dpp_src = kind, text, list(rd), list(wr), list(dpp_src)
        self.deps, self.succ, self.prio, self.done_at = set(), [], 0, 0


class Stream:
    def __init__(self):
        self.nodes = []
        self.edges = []          # (first, then): scheduling constraints beyond the register dependencies

    def add(self, kind, text, rd, wr, dpp_src=()):
        self.nodes.append(Node(kind, text, rd, wr, dpp_src))
        return self.nodes[-1]

    def link(self):
        last_w, readers = {}, {}
        for n in self.nodes:
            for r in n.rd:
                if r in last_w:
                    n.deps.add(last_w[r])                    # RAW
            for w in n.wr:
                for q in readers.get(w, []):
                    n.deps.add(q)                             # WAR
                if w in last_w:
                    n.deps.add(last_w[w])                     # WAW
            for r in n.rd:
                readers.setdefault(r, []).append(n)
            for w in n.wr:
                last_w[w] = n
                readers[w] = []
        for a_, b_ in self.edges:
            b_.deps.add(a_)
        for n in self.nodes:
            n.deps.discard(n)
            for d in n.deps:
                d.succ.append(n)

    def schedule(self):
        self.link()
        done = set()

        def prio(n):      # longest latency-weighted path to the end (the extra edges can point backwards in list order)
            if id(n) not in done:
                done.add(id(n))
                n.prio = COST[n.kind][1] + max((prio(q) for q in n.succ), default=0)
            return n.prio
        import sys
        sys.setrecursionlimit(10000)
        for n in self.nodes:
            prio(n)
        pending = {id(n): len(n.deps) for n in self.nodes}
        ready_at = {id(n): 0 for n in self.nodes}
        avail = [n for n in self.nodes if not n.deps]
        order, time = [], 0
        while avail:
            ok = [n for n in avail if ready_at[id(n)] <= time]
            pick = max(ok, key=lambda n: n.prio) if ok else min(avail, key=lambda n: (ready_at[id(n)], -n.prio))
            time = max(time, ready_at[id(pick)])
            avail.remove(pick)
            order.append(pick)
            issue, lat = COST[pick.kind]
            pick.done_at = time + lat
            time += issue
            for s in pick.succ:
                pending[id(s)] -= 1
                # RAW wants the result, WAR / WAW only the issue order: the latency counts for true dependencies only
                ready_at[id(s)] = max(ready_at[id(s)], pick.done_at if set(pick.wr) & set(s.rd) else 0)
                if pending[id(s)] == 0:
                    avail.append(s)
        assert len(order) == len(self.nodes)
        return order, time

    def emit(self):
        order, cycles = self.schedule()
        lines, writer, nops = [], {}, 0   # writer: register -> (index in `lines` of the instruction that wrote it, its kind)
        for n in order:
            need = 0
            for r in n.rd:
                if r not in writer:
                    continue
                at, kind = writer[r]
                gap = len(lines) - at - 1                     # instructions already between the two
                valu_w = kind not in ("dsr", "smov", "wait")
                want = 0
                if r in n.dpp_src and valu_w:
                    want = 2
                if kind == "rsq":
                    want = max(want, 1)
                if n.kind == "rl" and r.startswith("v") and valu_w:
                    want = max(want, 1)
                if kind == "rl" and r.startswith("s"):
                    want = max(want, 2)
                need = max(need, want - gap)
            if need > 0:
                lines.append(f"s_nop {need - 1}")
                nops += need
            lines.append(n.text)
            for w in n.wr:
                writer[w] = (len(lines) - 1, n.kind)
        return lines, cycles, len(order), nops


# ---------------------------------------------------------------------------------------------------------------------
# register maps (functions of the number of row slots S a lane carries)
K375 = 76
BARRIER_VARIANTS = [(1, 3), (2, 3), (2, 6), (3, 6)]        # (row slots, pivot behind which the barrier sits)
LAST_PANEL_PIVOTS = [8, 12]                                   # NZ - 16 (NT - 1) of the built horizons (vsmpc_device.hpp)
SLOTS = 1


class Map:
    def __init__(self, slots, keep_inv=0):
        """keep_inv: pivots whose 1 / L_jj gets a register pair of its own (the barrier variants: the rows join late and scale
        with it then)"""
        self.S = slots
        need = 32 * slots + 32 + 14 + 2 + 2 * slots + 2 * keep_inv
        self.A0 = min(100, (256 - need) & ~1)                  # a[s][c] = v[A0 + 32 s + 2c : +1]
        self.G0 = self.A0 + 32 * slots                         # g[c]
        t = self.G0 + 32
        self.Y, self.T, self.E, self.Z, self.W, self.INV, self.DP = t, t + 2, t + 4, t + 6, t + 8, t + 10, t + 12
        self.DGA, self.IVA = t + 14, t + 15                    # LDS byte addresses: diagonal row, 1 / L_jj
        self.LDA = [t + 16 + 2 * i for i in range(slots)]      # row load / store addresses per slot
        self.STA = [t + 17 + 2 * i for i in range(slots)]
        self.INVX = [t + 16 + 2 * slots + 2 * j for j in range(keep_inv)]
        assert t + 16 + 2 * slots + 2 * keep_inv <= 256

    def inv(self, j):
        return self.INVX[j] if j < len(self.INVX) else self.INV


M = Map(1)


def a(c, slot=0):
    return M.A0 + 32 * slot + 2 * c


def g(c):
    return M.G0 + 2 * c


def prologue(s, slots, bar=False):
    """Loads.  bar: the rows below the diagonal tile are requested only after the workgroup barrier inside the stream (the
    wavefronts that complete them arrive there later than the one tile this stream starts with)."""
    for c in range(NP):
        s.add("dsr", f"ds_read_b64 {vp(g(c))}, v{M.DGA} offset:{8 * c}", [f"v{M.DGA}"], regs("v", g(c)) + ["lgkm"])
    s.add("smov", f"s_mov_b32 s{K375}, 0", [], [f"s{K375}"])
    s.add("smov", f"s_mov_b32 s{K375 + 1}, 0x3fd80000", [], [f"s{K375 + 1}"])
    if bar:
        s.add("wait", "s_waitcnt lgkmcnt(0)", ["lgkm"], [r for c in range(NP) for r in regs("v", g(c))])
        s.bar = s.add("bar", "s_barrier", [], ["bar"])     # (held back until 1 / L_KB,KB is known: Stream.edges)
        for sl in range(slots):
            for c in range(NP):
                s.add("dsr", f"ds_read_b64 {vp(a(c, sl))}, v{M.LDA[sl]} offset:{8 * c}", [f"v{M.LDA[sl]}", "bar"], regs("v", a(c, sl)) + ["lgkm2"])
        s.add("wait", "s_waitcnt lgkmcnt(0)", ["lgkm2"], [r for sl in range(slots) for c in range(NP) for r in regs("v", a(c, sl))])
        return
    for sl in range(slots):
        for c in range(NP):
            s.add("dsr", f"ds_read_b64 {vp(a(c, sl))}, v{M.LDA[sl]} offset:{8 * c}", [f"v{M.LDA[sl]}"], regs("v", a(c, sl)) + ["lgkm"])
    # one wait for all loads (finer counts would let pivot 0 start earlier: ~100 cycles, not worth the bookkeeping)
    allr = [r for c in range(NP) for r in regs("v", g(c))] + [r for sl in range(slots) for c in range(NP) for r in regs("v", a(c, sl))]
    s.add("wait", "s_waitcnt lgkmcnt(0)", ["lgkm"], allr)


def epilogue(s):
    # the compiler's wait-count bookkeeping does not see the stores issued in here: they are complete when the block ends
    s.add("wait", "s_waitcnt lgkmcnt(0)", ["lds"], ["lds"])


def rsqrt_chain(s, d_src, d_regs, INV=None):
    """fast_rsqrt of vsmpc_kernels.hip on the pivot in `d_src` -> INV"""
    Y, T, E, Z, W = M.Y, M.T, M.E, M.Z, M.W
    INV = M.INV if INV is None else INV
    s.add("rsq", f"v_rsq_f64_e32 {vp(Y)}, {d_src}", d_regs, regs("v", Y))
    s.add("mul", f"v_mul_f64 {vp(T)}, {d_src}, {vp(Y)}", d_regs + regs("v", Y), regs("v", T))
    s.add("fma", f"v_fma_f64 {vp(E)}, -{vp(T)}, {vp(Y)}, 1.0", regs("v", T) + regs("v", Y), regs("v", E))
    s.add("mul", f"v_mul_f64 {vp(Z)}, {vp(Y)}, {vp(E)}", regs("v", Y) + regs("v", E), regs("v", Z))
    s.add("fma", f"v_fma_f64 {vp(W)}, {sp(K375)}, {vp(E)}, 0.5", regs("s", K375) + regs("v", E), regs("v", W))
    return s.add("fma", f"v_fma_f64 {vp(INV)}, {vp(Z)}, {vp(W)}, {vp(Y)}", regs("v", Z) + regs("v", W) + regs("v", Y), regs("v", INV))


def stream_dpp(npiv=NP, slots=1, bar_after=None):
    """slots = 0: the diagonal tile alone (the last panel: npiv pivots, its other rows carried as ordinary rows).
    bar_after = KB: the stream starts on the diagonal tile alone and joins a workgroup barrier once 1 / L_KB,KB is known; the
    rows below are loaded behind it, and their share of pivots 0 .. KB is caught up from there (same instructions, later)."""
    s = Stream()
    prologue(s, slots, bar_after is not None)
    DP = M.DP
    for j in range(npiv):
        INV = M.inv(j)
        s.add("movdpp", f"v_mov_b64_dpp {vp(DP)}, {vp(g(j))} row_newbcast:{j} row_mask:0xf bank_mask:0xf", regs("v", g(j)), regs("v", DP),
              dpp_src=regs("v", g(j)))
        inv_node = rsqrt_chain(s, vp(DP), regs("v", DP), INV)
        if bar_after == j:
            s.edges.append((inv_node, s.bar))
        # 1 / L_jj for P5 and the tile inverses: the same value from every lane to the same address (wavefront 0; the others
        # are handed a dummy).  A non-positive pivot needs no bookkeeping: its reciprocal square root is NaN and so is
        # everything computed from it, down to the last pivot's, which the caller tests.
        s.add("dsw", f"ds_write_b64 v{M.IVA}, {vp(INV)} offset:{8 * j}", [f"v{M.IVA}"] + regs("v", INV), ["lds"])
        s.add("mul", f"v_mul_f64 {vp(g(j))}, {vp(g(j))}, {vp(INV)}", regs("v", g(j)) + regs("v", INV), regs("v", g(j)))
        for sl in range(slots):
            s.add("mul", f"v_mul_f64 {vp(a(j, sl))}, {vp(a(j, sl))}, {vp(INV)}", regs("v", a(j, sl)) + regs("v", INV), regs("v", a(j, sl)))
            s.add("dsw", f"ds_write_b64 v{M.STA[sl]}, {vp(a(j, sl))} offset:{8 * j}", [f"v{M.STA[sl]}"] + regs("v", a(j, sl)), ["lds"])
        for c in range(j + 1, NP):
            for dst, src1 in [(g(c), g(j))] + [(a(c, sl), a(j, sl)) for sl in range(slots)]:
                s.add("dpp", f"v_fmac_f64_dpp {vp(dst)}, -{vp(g(j))}, {vp(src1)} row_newbcast:{c} row_mask:0xf bank_mask:0xf",
                      regs("v", g(j)) + regs("v", src1) + regs("v", dst), regs("v", dst), dpp_src=regs("v", g(j)))
    epilogue(s)
    return s


class RowsMap:
    """rows-only stream: a[s][c], g[c], inv[j] pairs, addresses"""
    def __init__(self, slots):
        self.S = slots
        self.A0 = 72
        self.G0 = self.A0 + 32 * slots
        self.I0 = self.G0 + 32
        t = self.I0 + 32
        self.DGA, self.IVA = t, t + 1
        self.LDA = [t + 2 + 2 * i for i in range(slots)]
        self.STA = [t + 3 + 2 * i for i in range(slots)]
        assert t + 2 + 2 * slots <= 256


def stream_rows(slots):
    """The rows below an already factored diagonal tile: a <- a L^-T, column by column (right-looking, the operation order of
    the fused stream: scale column j by 1 / L_jj, then a_c -= a_j l_cj for c > j).  The factored tile (row c of it in lane
    16 r + c) and the sixteen 1 / L_jj come from LDS."""
    R = RowsMap(slots)
    s = Stream()
    ar = lambda c, sl: R.A0 + 32 * sl + 2 * c
    gr = lambda c: R.G0 + 2 * c
    ir = lambda j: R.I0 + 2 * j
    for c in range(NP):
        s.add("dsr", f"ds_read_b64 {vp(gr(c))}, v{R.DGA} offset:{8 * c}", [f"v{R.DGA}"], regs("v", gr(c)) + ["lgkm"])
    for j in range(NP):
        s.add("dsr", f"ds_read_b64 {vp(ir(j))}, v{R.IVA} offset:{8 * j}", [f"v{R.IVA}"], regs("v", ir(j)) + ["lgkm"])
    for sl in range(slots):
        for c in range(NP):
            s.add("dsr", f"ds_read_b64 {vp(ar(c, sl))}, v{R.LDA[sl]} offset:{8 * c}", [f"v{R.LDA[sl]}"], regs("v", ar(c, sl)) + ["lgkm"])
    allr = [r for c in range(NP) for r in regs("v", gr(c)) + regs("v", ir(c))] + [r for sl in range(slots) for c in range(NP) for r in regs("v", ar(c, sl))]
    s.add("wait", "s_waitcnt lgkmcnt(0)", ["lgkm"], allr)
    for j in range(NP):
        for sl in range(slots):
            s.add("mul", f"v_mul_f64 {vp(ar(j, sl))}, {vp(ar(j, sl))}, {vp(ir(j))}", regs("v", ar(j, sl)) + regs("v", ir(j)), regs("v", ar(j, sl)))
            s.add("dsw", f"ds_write_b64 v{R.STA[sl]}, {vp(ar(j, sl))} offset:{8 * j}", [f"v{R.STA[sl]}"] + regs("v", ar(j, sl)), ["lds"])
        for c in range(j + 1, NP):
            for sl in range(slots):
                s.add("dpp", f"v_fmac_f64_dpp {vp(ar(c, sl))}, -{vp(gr(j))}, {vp(ar(j, sl))} row_newbcast:{c} row_mask:0xf bank_mask:0xf",
                      regs("v", gr(j)) + regs("v", ar(j, sl)) + regs("v", ar(c, sl)), regs("v", ar(c, sl)), dpp_src=regs("v", gr(j)))
    epilogue(s)
    return s, R


def stream_inverse():
    """X = L^-1 of a factored diagonal tile: the rows stream on the rows of the identity (lane c starts from e_c and ends with
    row c of L^-T = column c of X), stored transposed: X[i][c] at x_addr + 8 (17 i + c), the tile layout the readers expect.
    Lanes >= 16 repeat lanes 0..15 (same values to the same addresses)."""
    R = RowsMap(1)
    s = Stream()
    ar = lambda c: R.A0 + 2 * c
    gr = lambda c: R.G0 + 2 * c
    ir = lambda j: R.I0 + 2 * j
    LN = R.LDA[0]                      # lane & 15 (input)
    for c in range(NP):
        s.add("dsr", f"ds_read_b64 {vp(gr(c))}, v{R.DGA} offset:{8 * c}", [f"v{R.DGA}"], regs("v", gr(c)) + ["lgkm"])
    for j in range(NP):
        s.add("dsr", f"ds_read_b64 {vp(ir(j))}, v{R.IVA} offset:{8 * j}", [f"v{R.IVA}"], regs("v", ir(j)) + ["lgkm"])
    for c in range(NP):
        s.add("mov", f"v_mov_b32_e32 v{ar(c)}, 0", [], [f"v{ar(c)}"])
        s.add("cmp", f"v_cmp_eq_u32_e32 vcc, {c}, v{LN}", [f"v{LN}"], ["vcc"])
        s.add("cnd", f"v_cndmask_b32_e32 v{ar(c) + 1}, 0, v{R.STA[0] + 1}, vcc", ["vcc", f"v{R.STA[0] + 1}"], [f"v{ar(c) + 1}"])
    allr = [r for c in range(NP) for r in regs("v", gr(c)) + regs("v", ir(c))]
    s.add("wait", "s_waitcnt lgkmcnt(0)", ["lgkm"], allr)
    for j in range(NP):
        s.add("mul", f"v_mul_f64 {vp(ar(j))}, {vp(ar(j))}, {vp(ir(j))}", regs("v", ar(j)) + regs("v", ir(j)), regs("v", ar(j)))
        s.add("dsw", f"ds_write_b64 v{R.STA[0]}, {vp(ar(j))} offset:{8 * 17 * j}", [f"v{R.STA[0]}"] + regs("v", ar(j)), ["lds"])
        for c in range(j + 1, NP):
            s.add("dpp", f"v_fmac_f64_dpp {vp(ar(c))}, -{vp(gr(j))}, {vp(ar(j))} row_newbcast:{c} row_mask:0xf bank_mask:0xf",
                  regs("v", gr(j)) + regs("v", ar(j)) + regs("v", ar(c)), regs("v", ar(c)), dpp_src=regs("v", gr(j)))
    epilogue(s)
    return s, R


def function_inverse():
    s, R = stream_inverse()
    lines, cycles, ninstr, nops = s.emit()
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    one_hi = R.STA[0] + 1
    ins = [f'"{{v{R.DGA}}}"(diag_addr)', f'"{{v{R.IVA}}}"(invd_addr)', f'"{{v{R.LDA[0]}}}"(lane15)', f'"{{v{R.STA[0]}}}"(x_addr)',
           f'"{{v{one_hi}}}"(0x3ff00000u)']
    bound = set([f"v{R.DGA}", f"v{R.IVA}", f"v{R.LDA[0]}", f"v{R.STA[0]}", f"v{one_hi}"])
    touched = set()
    for n in s.nodes:
        touched.update(r for r in n.rd + n.wr if r[0] in "vs" and r[1:].isdigit())
    clob = sorted(touched - bound, key=lambda r: (r[0], int(r[1:])))
    clobbers = ", ".join(f'"{r}"' for r in clob) + ', "vcc", "memory"'
    text = ("\n// X = L^-1 of the factored diagonal tile at diag_addr (this lane's row: lane & 15), 1 / L_jj at invd_addr: the rows stream on\n"
            "// the rows of the identity, stored transposed -- X[i][c] at x_addr + 8 (17 i), x_addr = &X[0][lane & 15].\n"
            f"// {ninstr} instructions + {nops} wait states, {cycles} cycles in the generator's issue model.\n")
    text += f"VS_DEV void panel_inverse_dpp(unsigned diag_addr, unsigned invd_addr, unsigned x_addr, int lane15) {{\n    asm volatile(\n{body}\n"
    text += "        :\n        : " + ", ".join(ins) + "\n        : " + clobbers + ");\n}\n"
    return text, ninstr, nops, cycles


def function_rows(name, slots, comment):
    s, R = stream_rows(slots)
    lines, cycles, ninstr, nops = s.emit()
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    ins = [f'"{{v{R.DGA}}}"(diag_addr)', f'"{{v{R.IVA}}}"(invd_addr)']
    bound = set([f"v{R.DGA}", f"v{R.IVA}"])
    sig = ""
    for sl in range(slots):
        ins += [f'"{{v{R.LDA[sl]}}}"(load_addr{sl})', f'"{{v{R.STA[sl]}}}"(store_addr{sl})']
        bound.update([f"v{R.LDA[sl]}", f"v{R.STA[sl]}"])
        sig += f"unsigned load_addr{sl}, unsigned store_addr{sl}, "
    touched = set()
    for n in s.nodes:
        touched.update(r for r in n.rd + n.wr if r[0] in "vs" and r[1:].isdigit())
    clob = sorted(touched - bound, key=lambda r: (r[0], int(r[1:])))
    clobbers = ", ".join(f'"{r}"' for r in clob) + ', "memory"'
    sig += "unsigned diag_addr, unsigned invd_addr"
    text = f"\n// {comment}\n// {ninstr} instructions + {nops} wait states, {cycles} cycles in the generator's issue model.\n"
    text += f"VS_DEV void {name}({sig}) {{\n    asm volatile(\n{body}\n"
    text += "        :\n        : " + ", ".join(ins) + "\n        : " + clobbers + ");\n}\n"
    return text, ninstr, nops, cycles


HEADER = """// GENERATED by tools/gen_panel_asm.py -- do not edit (the why and the how are in that file's header).
// Sixteen-pivot panel streams of P3 as hand-scheduled gfx950 assembly.  Same arithmetic, operation by operation, as
// panel_factor<1, 16> (bit-identical results); what differs is how the pivot column reaches the other lanes and the order.
"""


def function(name, s, comment, slots):
    lines, cycles, ninstr, nops = s.emit()
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    outs = [f'"={{v[{g(c)}:{g(c) + 1}]}}"(g[{c}])' for c in range(NP)] + [f'"={{v[{M.INV}:{M.INV + 1}]}}"(inv_last)']
    ins = [f'"{{v{M.DGA}}}"(diag_addr)', f'"{{v{M.IVA}}}"(invd_addr)']
    bound = set(regs("v", M.INV) + [f"v{M.DGA}", f"v{M.IVA}"])
    sig = ""
    for sl in range(slots):
        ins += [f'"{{v{M.LDA[sl]}}}"(load_addr{sl})', f'"{{v{M.STA[sl]}}}"(store_addr{sl})']
        bound.update([f"v{M.LDA[sl]}", f"v{M.STA[sl]}"])
        sig += f"unsigned load_addr{sl}, unsigned store_addr{sl}, "
    for c in range(NP):
        bound.update(regs("v", g(c)))
    touched = set()
    for n in s.nodes:
        touched.update(r for r in n.rd + n.wr if r[0] in "vs" and r[1:].isdigit())
    clob = sorted(touched - bound, key=lambda r: (r[0], int(r[1:])))
    clobbers = ", ".join(f'"{r}"' for r in clob) + ', "memory"'
    sig += "unsigned diag_addr, unsigned invd_addr, double (&g)[16], double& inv_last"
    text = f"\n// {comment}\n// {ninstr} instructions + {nops} wait states, {cycles} cycles in the generator's issue model.\n"
    text += f"VS_DEV void {name}({sig}) {{\n    asm volatile(\n{body}\n"
    text += "        : " + ", ".join(outs) + "\n        : " + ", ".join(ins) + "\n        : " + clobbers + ");\n}\n"
    return text, ninstr, nops, cycles


def main():
    global M
    text = HEADER
    todo = []
    for slots in (1, 2, 3):
        todo.append((f"panel16x{slots}_dpp", 16, slots,
                     f"{slots} row slot(s): lane 16 r + c carries the panel rows at load_addr0.. (stored, finished, to store_addr0..: the same row, or a\n"
                     "// 16-double dummy for rows beyond the matrix) and row c of the diagonal tile (`diag_addr`, returned factored in g).  1 / L_jj\n"
                     "// goes to invd_addr[j] (every lane writes it: a dummy for all wavefronts but one); inv_last = 1 / L_15,15 is NaN iff a pivot\n"
                     "// was not positive."))
    for slots, kb in BARRIER_VARIANTS:
        todo.append((f"panel16x{slots}_b{kb}_dpp", 16, (slots, kb),
                     f"{slots} row slot(s), with a workgroup barrier INSIDE: the stream starts when the diagonal tile is ready, factors pivots 0 .. {kb} of it,\n"
                     f"// joins the barrier that says the rows below are complete (s_barrier: every other wavefront of the workgroup must execute a\n"
                     "// matching one), loads them and catches their share up.  Otherwise panel16xS_dpp."))
    todo.append(("panel_diag16_dpp", 16, 0, "The diagonal tile alone, 16 pivots (the pipelined schedule factors it ahead of the rows below it: "
                 "panel_rows*_dpp).\n// Lane 16 r + c carries row c; inv_last = 1 / L_15,15."))
    for npiv in LAST_PANEL_PIVOTS:
        todo.append((f"panel_last{npiv}_dpp", npiv, 0, f"The last panel: {npiv} pivots in the diagonal tile, whose other rows are carried as ordinary "
                     f"rows.\n// Lane 16 r + c carries row c (r = 0 is the copy that is stored); inv_last = 1 / L_{npiv - 1},{npiv - 1}."))
    for name, npiv, slots, comment in todo:
        kb = None
        if isinstance(slots, tuple):
            slots, kb = slots
        M = Map(max(slots, 1), 0 if kb is None else kb + 1)
        t, ninstr, nops, cycles = function(name, stream_dpp(npiv, slots, kb), comment, slots)
        text += t
        print(f"{name}: {ninstr} instructions, {nops} wait states, modelled {cycles} cycles")
    for slots in (1, 2, 3):
        t, ninstr, nops, cycles = function_rows(f"panel_rows{slots}_dpp", slots,
                                                f"{slots} row slot(s) below a diagonal tile that is already factored (in LDS at diag_addr, its 1 / L_jj at invd_addr): "
                                                "a <- a L^-T.\n// Same operations in the same order as the fused streams above.")
        text += t
        print(f"panel_rows{slots}_dpp: {ninstr} instructions, {nops} wait states, modelled {cycles} cycles")
    t, ninstr, nops, cycles = function_inverse()
    text += t
    print(f"panel_inverse_dpp: {ninstr} instructions, {nops} wait states, modelled {cycles} cycles")
    with open(OUT, "w") as f:
        f.write(text)
    print("->", OUT)


if __name__ == "__main__":
    main()
