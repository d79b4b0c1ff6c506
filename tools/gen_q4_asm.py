#!/usr/bin/env python3
"""Writes clipcap_amd/csrc/gemm_q4_asm.inc: the instruction-level K loop of the persistent 4-wave NT GEMM (gemm_q4.hip.h).

One `asm volatile` statement per output tile: head (K-steps 0..3), main loop (4 K-steps per iteration), tail (last 4 K-steps, whose
LDS-DMA instructions already stream the NEXT tile's first four stages).  Every K-step is

    s_waitcnt vmcnt(2 ND) lgkmcnt(0) ; s_barrier ; 8 NI x v_mfma_f32_16x16x32 with, hand-placed in the gaps,
        NI + 8 ds_read_b128 (fragments of step t + 1 -> the other register set), ND x { s_add_u32 m0 ; global_load_lds_dwordx4 }
        (step t + 4 -> ring stage t % 4) with the 64-byte K bump of each DMA's VGPR offset.

All registers are constraint operands (accumulators "=&a", fragments "v"), so hipcc allocates them and nothing inside the statement is
visible to its scheduler.  Hazards handled in the text: SALU write of M0 -> LDS-DMA needs one wait state (an MFMA sits between), the
statement opens with s_nop 4 (fresh readfirstlane SGPRs read by VMEM) and ends with s_nop 7 x 2 (MFMA result -> compiler's accvgpr reads).

Usage: python3 tools/gen_q4_asm.py  (re-run after editing; the .inc is committed so the build does not need python)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "clipcap_amd", "csrc", "gemm_q4_asm.inc")


class Gen:
    def __init__(self, ni, read_every=2, dma_gap=None, first_read_slot=1, nodma=False, nobarrier=False, dma_first=4):
        self.ni = ni
        self.sbm = 32 * ni
        self.sa = self.sbm * 64
        self.sstage = (self.sbm + 256) * 64
        self.na = ni // 2
        self.nb = 4
        self.nd = self.na + self.nb
        self.nmfma = 8 * ni
        self.read_every = read_every
        self.first_read_slot = first_read_slot
        self.dma_gap = dma_gap if dma_gap is not None else self.nmfma // self.nd
        self.nodma, self.nobarrier, self.dma_first = nodma, nobarrier, dma_first
        self.lines = []

    def emit(self, s):
        self.lines.append(s)

    # operand names -------------------------------------------------------------------------------------------------
    def acc(self, i, j):
        return "%%[c%d_%d]" % (i, j)

    def fa(self, s, i):
        return "%%[a%d_%d]" % (s, i)

    def fb(self, s, j):
        return "%%[b%d_%d]" % (s, j)

    def step(self, st, cur, vm, zero_c=False):
        """K-step reading ring stage (st + 1) & 3 into fragment set cur ^ 1 and streaming into ring stage st."""
        ni, nd = self.ni, self.nd
        nxt = cur ^ 1
        if vm is None:
            self.emit("s_waitcnt lgkmcnt(0)")
        else:
            self.emit("s_waitcnt vmcnt(%d) lgkmcnt(0)" % vm)
        if not self.nobarrier:
            self.emit("s_barrier")
        # fillers per MFMA slot
        slots = [[] for _ in range(self.nmfma)]
        rs = (st + 1) & 3
        base = "lo" if rs < 2 else "hi"
        reads = []
        for j in range(8):
            reads.append("ds_read_b128 %s, %%[lb_%s] offset:%d" % (self.fb(nxt, j), base, (rs & 1) * self.sstage + j * 1024))
        for i in range(ni):
            reads.append("ds_read_b128 %s, %%[la_%s] offset:%d" % (self.fa(nxt, i), base, (rs & 1) * self.sstage + i * 1024))
        for r, text in enumerate(reads):
            slots[self.first_read_slot + r * self.read_every].append(text)
        # DMA d: M0 write, one MFMA (the wait state an SALU write of M0 needs before an LDS-DMA), the load, one MFMA, the K bump of its offset
        dmas = []
        for i in range(self.na):
            dmas.append(("%%[ta%d]" % i, "%[pa]", st * self.sstage + 4 * i * 1024))
        for j in range(self.nb):
            dmas.append(("%%[tb%d]" % j, "%[pb]", st * self.sstage + self.sa + 4 * j * 1024))
        for d, (voff, ptr, imm) in enumerate(dmas):
            s0 = self.dma_first + d * self.dma_gap
            assert s0 + 2 < self.nmfma, "DMA slots run past the step"
            slots[s0].append("s_add_u32 m0, %%[slds], 0x%x" % imm)
            if not self.nodma:
                slots[s0 + 1].append("global_load_lds_dwordx4 %s, %s" % (voff, ptr))
            slots[s0 + 2].append("v_add_u32 %s, 64, %s" % (voff, voff))
        k = 0
        for j in range(8):
            for i in range(ni):
                c = self.acc(i, j)
                self.emit("Q4_MFMA \" %s, %s, %s, %s\\n\"" % (c, self.fb(cur, j), self.fa(cur, i), "0" if zero_c else c))
                for f in slots[k]:
                    self.emit(f)
                k += 1

    def set_offsets(self, src, add):
        for i in range(self.na):
            self.emit(("v_add_u32 %%[ta%d], %d, %%[%sa%d]" % (i, add, src, i)) if add else ("v_mov_b32 %%[ta%d], %%[%sa%d]" % (i, src, i)))
        for j in range(self.nb):
            self.emit(("v_add_u32 %%[tb%d], %d, %%[%sb%d]" % (j, add, src, j)) if add else ("v_mov_b32 %%[tb%d], %%[%sb%d]" % (j, src, j)))

    def tile(self):
        nd = self.nd
        self.emit("s_nop 4")
        self.set_offsets("o", 256)              # head: the DMA stream continues inside this tile at K-step 4
        self.step(0, 0, None, zero_c=True)
        self.step(1, 1, None)
        self.step(2, 0, 2 * nd)
        self.step(3, 1, 2 * nd)
        self.emit("s_cmp_eq_u32 %[nmain], 0")
        self.emit("s_cbranch_scc1 Q4_TAIL_%=")
        self.emit("s_mov_b32 %[cnt], %[nmain]")
        self.emit("Q4_LOOP_%=:")
        for s in range(4):
            self.step(s, s & 1, 2 * nd)
        self.emit("s_sub_u32 %[cnt], %[cnt], 1")
        self.emit("s_cmp_lg_u32 %[cnt], 0")
        self.emit("s_cbranch_scc1 Q4_LOOP_%=")
        self.emit("Q4_TAIL_%=:")
        self.set_offsets("n", 0)                # tail: the next tile's K-steps 0..3
        for s in range(4):
            self.step(s, s & 1, 2 * nd)
        self.emit("s_waitcnt vmcnt(%d) lgkmcnt(0)" % nd)
        self.emit("s_nop 7")
        self.emit("s_nop 7")

    def text(self, name):
        out = ["#define %s \\" % name]
        for l in self.lines:
            if l.startswith("Q4_MFMA"):
                out.append("    %s \\" % l)
            else:
                out.append("    \"%s\\n\" \\" % l)
        out.append("    \"\"")
        return "\n".join(out)

    def operands(self, name):
        ni = self.ni
        outs, ins = [], []
        n_acc = 0
        for j in range(8):
            for i in range(ni):
                # the AGPR half holds 64 accumulator tiles; a 320-row tile keeps its last 16 in VGPRs
                cons = "=&a" if n_acc < 64 else "=&v"
                outs.append("[c%d_%d] \"%s\"(acc[%d][%d][%d])" % (i, j, cons, j >> 2, i, j & 3))
                n_acc += 1
        for i in range(ni):
            outs.append("[a1_%d] \"=&v\"(af1[%d])" % (i, i))
        for j in range(8):
            outs.append("[b1_%d] \"=&v\"(bf1[%d])" % (j, j))
        for i in range(ni):
            outs.append("[a0_%d] \"+v\"(af0[%d])" % (i, i))
        for j in range(8):
            outs.append("[b0_%d] \"+v\"(bf0[%d])" % (j, j))
        for i in range(self.na):
            outs.append("[ta%d] \"=&v\"(toA[%d])" % (i, i))
        for j in range(self.nb):
            outs.append("[tb%d] \"=&v\"(toB[%d])" % (j, j))
        outs.append("[cnt] \"=&s\"(q4_cnt)")
        for i in range(self.na):
            ins.append("[oa%d] \"v\"(offA[%d])" % (i, i))
        for j in range(self.nb):
            ins.append("[ob%d] \"v\"(offB[%d])" % (j, j))
        for i in range(self.na):
            ins.append("[na%d] \"v\"(noffA[%d])" % (i, i))
        for j in range(self.nb):
            ins.append("[nb%d] \"v\"(noffB[%d])" % (j, j))
        for t in ("la_lo", "la_hi", "lb_lo", "lb_hi"):
            ins.append("[%s] \"v\"(q4_%s)" % (t, t))
        for t in ("pa", "pb", "slds", "nmain"):
            ins.append("[%s] \"s\"(q4_%s)" % (t, t))
        return ("#define %s_OUTS \\\n    " % name + ", \\\n    ".join(outs) + "\n" +
                "#define %s_INS \\\n    " % name + ", \\\n    ".join(ins) + "\n")


def main():
    parts = ["// Generated by tools/gen_q4_asm.py — do not edit.  The K loop of gemm_nt_q4_kernel as one asm statement per tile.\n"]
    # variant 0 is the product's; the others are the A/B arms of tools/probes/q4_bench.hip (CC_Q4_VARIANTS)
    variants = {
        0: dict(),
        1: dict(read_every=1),
        2: dict(read_every=3, dma_gap=7),
        3: dict(nodma=True),                      # ablation: no LDS-DMA (results meaningless)
        4: dict(nobarrier=True),                  # ablation: no barrier (results may be wrong)
        5: dict(read_every=1, first_read_slot=0, dma_first=17, dma_gap=5),   # all fragment reads first, DMAs behind them
    }
    for ni in (8, 10):
        for v, kw in variants.items():
            if v:
                parts.append("#ifdef CC_Q4_VARIANTS")
            g = Gen(ni, **kw)
            g.tile()
            parts.append(g.text("Q4_TILE_ASM_%d_%d" % (ni, v)))
            if v:
                parts.append("#endif")
            parts.append("")
        parts.append(Gen(ni).operands("Q4_TILE_%d" % ni))
    with open(OUT, "w") as f:
        f.write("\n".join(parts))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
