#!/usr/bin/env python3
"""Writes clipcap_amd/csrc/gemm_q4_asm.inc: the instruction-level K loop of the persistent 4-wave NT GEMM (gemm_q4.hip.h).

One `asm volatile` statement per output tile.  LDS holds NS stages of 64 K-columns: A image [32 NI rows][128 B] then B image
[256 rows][128 B], 16-B chunk index XORed with (row >> 1) & 7 — every LDS-DMA instruction (1 KiB = 8 rows x 128 B) fetches FULL
128-byte lines (the 8-wave staggered kernel's 32-deep stages fetch 16 rows x 64 B per instruction: every line is requested twice).
A stage is consumed in two half-steps of 32 K-columns; fragments are double-buffered in registers (set = K half):

    half a of stage T:  s_waitcnt lgkmcnt(0) ; s_barrier ; 8 NI MFMAs (set 0)  |  NI + 8 ds_read_b128: K half 1 of stage T -> set 1
    half b of stage T:  s_waitcnt vmcnt((NS - 2) ND) lgkmcnt(0) ; s_barrier ; 8 NI MFMAs (set 1)
                          |  NI + 8 ds_read_b128: K half 0 of stage T + 1 -> set 0
                          |  ND x { s_add_u32 m0 ; global_load_lds_dwordx4 ; v_add_u32 offset, 128 }: stage T + NS -> the buffer just consumed

The tile's statement = head (NS stages), main loop (NS stages per iteration), tail (NS stages whose DMA instructions stream the NEXT tile's first NS stages).
All registers are constraint operands (accumulators "+a", fragments "v"), so hipcc allocates them and nothing inside the statement is
visible to its scheduler.  Hazards handled in the text: SALU write of M0 -> LDS-DMA needs one wait state (an MFMA sits between), the
statement opens with s_nop 4 (fresh readfirstlane SGPRs read by VMEM) and ends with s_nop 7 x 2 (MFMA result -> compiler's accvgpr reads).

Usage: python3 tools/gen_q4_asm.py  (re-run after editing; the .inc is committed so the build does not need python)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "clipcap_amd", "csrc", "gemm_q4_asm.inc")


class Gen:
    def __init__(self, ni, ns, read_every=2, first_read_slot=1, dma_first=2, nodma=False, noread=False):
        self.ni, self.ns = ni, ns
        self.sbm = 32 * ni
        self.sa = self.sbm * 128
        self.sstage = (self.sbm + 256) * 128
        self.na = ni                     # A pieces (8 rows x 128 B) per wave per stage: 32 NI / 8 / 4 waves
        self.nb = 8
        self.nd = self.na + self.nb
        self.nmfma = 8 * ni
        self.read_every, self.first_read_slot, self.dma_first, self.nodma = read_every, first_read_slot, dma_first, nodma
        self.noread = noread
        self.dma_gap = (self.nmfma - dma_first - 3) // (self.nd - 1)
        assert self.dma_gap >= 2, "M0 write, one MFMA, the load: consecutive groups may overlap by one slot"
        assert ns * self.sstage <= 160 * 1024
        self.lines = []

    def emit(self, s):
        self.lines.append(s)

    def acc(self, i, j):
        return "%%[c%d_%d]" % (i, j)

    def fa(self, s, i):
        return "%%[a%d_%d]" % (s, i)

    def fb(self, s, j):
        return "%%[b%d_%d]" % (s, j)

    def half(self, buf, kk, vm, dma):
        """Half-step kk (0 = a, 1 = b) of the stage in ring buffer `buf`; vm: None = LDS wait only; dma: issue ND DMA instructions into `buf`."""
        ni = self.ni
        if vm is None:
            self.emit("s_waitcnt lgkmcnt(0)")
        else:
            self.emit("s_waitcnt vmcnt(%d) lgkmcnt(0)" % vm)
        self.emit("s_barrier")
        slots = [[] for _ in range(self.nmfma)]
        # reads: half a -> K half 1 of the same buffer into set 1; half b -> K half 0 of the next buffer into set 0
        rbuf, rk, rset = (buf, 1, 1) if kk == 0 else ((buf + 1) % self.ns, 0, 0)
        reads = []
        for j in range(8):
            reads.append("ds_read_b128 %s, %%[lb%d_%d] offset:%d" % (self.fb(rset, j), rbuf, rk, j * 2048))
        for i in range(ni):
            reads.append("ds_read_b128 %s, %%[la%d_%d] offset:%d" % (self.fa(rset, i), rbuf, rk, i * 2048))
        for r, text in enumerate(reads):
            if not self.noread:
                slots[self.first_read_slot + r * self.read_every].append(text)
        if dma:
            dmas = []
            for i in range(self.na):
                dmas.append(("%%[ta%d]" % i, "%[pa]", buf * self.sstage + 4 * i * 1024))
            for j in range(self.nb):
                dmas.append(("%%[tb%d]" % j, "%[pb]", buf * self.sstage + self.sa + 4 * j * 1024))
            for d, (voff, ptr, imm) in enumerate(dmas):
                s0 = self.dma_first + d * self.dma_gap
                assert s0 + 2 < self.nmfma
                slots[s0].append("s_add_u32 m0, %%[slds], 0x%x" % imm)
                if not self.nodma:
                    slots[s0 + 1].append("global_load_lds_dwordx4 %s, %s" % (voff, ptr))
                slots[s0 + 2].append("v_add_u32 %s, 128, %s" % (voff, voff))
        k = 0
        for j in range(8):
            for i in range(ni):
                c = self.acc(i, j)
                self.emit("Q4_MFMA \" %s, %s, %s, %s\\n\"" % (c, self.fb(kk, j), self.fa(kk, i), c))
                for f in slots[k]:
                    self.emit(f)
                k += 1

    def stage(self, buf, first=False, dma=True, vm=-1):
        self.half(buf, 0, None, False)
        self.half(buf, 1, None if first else ((self.ns - 2) * self.nd if vm == -1 else vm), dma)

    def set_offsets(self, src, add):
        for i in range(self.na):
            self.emit(("v_add_u32 %%[ta%d], %d, %%[%sa%d]" % (i, add, src, i)) if add else ("v_mov_b32 %%[ta%d], %%[%sa%d]" % (i, src, i)))
        for j in range(self.nb):
            self.emit(("v_add_u32 %%[tb%d], %d, %%[%sb%d]" % (j, add, src, j)) if add else ("v_mov_b32 %%[tb%d], %%[%sb%d]" % (j, src, j)))

    def tile(self):
        """One tile of one workgroup.  The tail (the tile's last NS stages) exists twice, chosen by the scalar %[more]: with a next tile its b half-steps
        stream that tile's stages 0 .. NS-1; on the workgroup's final tile they issue no DMA (the dummy stream cost 3 of 12 + 3 stages' worth of
        LDS-DMA on a K = 768 launch) and wait for exactly the stages still in flight.  Both tails in ONE statement: two asm statements on the two
        sides of a C++ branch made the compiler copy the accumulators (119 spilled registers)."""
        ns = self.ns
        self.emit("s_nop 4")
        self.set_offsets("o", 128 * ns)            # head: the DMA stream continues inside this tile at stage NS
        for b in range(ns):
            self.stage(b)        # (stage 0's b barrier waits like every other: a single-tile launch then starts on stage 0 alone, the prologue awaits no more)
        self.emit("s_cmp_eq_u32 %[nmain], 0")
        self.emit("s_cbranch_scc1 Q4_TAIL_%=")
        self.emit("s_mov_b32 %[cnt], %[nmain]")
        self.emit("Q4_LOOP_%=:")
        for b in range(ns):
            self.stage(b)
        self.emit("s_sub_u32 %[cnt], %[cnt], 1")
        self.emit("s_cmp_lg_u32 %[cnt], 0")
        self.emit("s_cbranch_scc1 Q4_LOOP_%=")
        self.emit("Q4_TAIL_%=:")
        self.emit("s_cmp_eq_u32 %[more], 0")
        self.emit("s_cbranch_scc1 Q4_FINAL_%=")
        self.set_offsets("n", 0)                   # tail: the next tile's stages 0 .. NS-1
        for b in range(ns):
            self.stage(b)
        # next tile: stages 0 and 1 landed (NS = 2: everything), the fragments of its first half-step are in set 0
        self.emit("s_waitcnt vmcnt(%d) lgkmcnt(0)" % ((ns - 2) * self.nd))
        self.emit("s_branch Q4_END_%=")
        self.emit("Q4_FINAL_%=:")
        # stage T = nbig - NS + b: still in flight behind it are the stages up to nbig - 1 -> (NS - 2 - b) of them may stay out at its b barrier
        for b in range(ns):
            left = ns - 2 - b
            self.stage(b, dma=False, vm=(left * self.nd if left >= 0 else None))
        self.emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.emit("Q4_END_%=:")
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
                cons = "+a" if n_acc < 64 else "+v"        # the AGPR half holds 64 accumulator tiles; the kernel zeroes them or starts them from the functor's additive input
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
        for b in range(self.ns):
            for kk in range(2):
                ins.append("[la%d_%d] \"v\"(q4_la[%d][%d])" % (b, kk, b, kk))
                ins.append("[lb%d_%d] \"v\"(q4_lb[%d][%d])" % (b, kk, b, kk))
        for t in ("pa", "pb", "slds", "nmain", "more"):
            ins.append("[%s] \"s\"(q4_%s)" % (t, t))
        return ("#define %s_OUTS \\\n    " % name + ", \\\n    ".join(outs) + "\n" +
                "#define %s_INS \\\n    " % name + ", \\\n    ".join(ins) + "\n")


# (NI, NS) forms: 256 x 256 and 160 x 256 with two or three 64-deep stages
FORMS = [(8, 2), (5, 3), (5, 2)]
# variant 0 is the product's; the others are the A/B arms of tools/probes/q4_bench.hip (CC_Q4_VARIANTS)
VARIANTS = {
    0: dict(),
    1: dict(read_every=1),
    2: dict(nodma=True),                          # ablation: no LDS-DMA (results meaningless)
    3: dict(noread=True),                         # ablation: no fragment reads in the K loop (results meaningless): what the ds_read_b128 stream costs
}


def main():
    parts = ["// Generated by tools/gen_q4_asm.py — do not edit.  The K loop of gemm_nt_q4_kernel as one asm statement per tile.\n"]
    for ni, ns in FORMS:
        for v, kw in VARIANTS.items():
            if v:
                parts.append("#ifdef CC_Q4_VARIANTS")
            g = Gen(ni, ns, **kw)
            g.tile()
            parts.append(g.text("Q4_TILE_ASM_%d_%d_%d" % (ni, ns, v)))
            if v:
                parts.append("#endif")
            parts.append("")
        parts.append(Gen(ni, ns).operands("Q4_TILE_%d_%d" % (ni, ns)))
    with open(OUT, "w") as f:
        f.write("\n".join(parts))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
