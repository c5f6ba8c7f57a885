#!/usr/bin/env python
"""Emit pips_amd/csrc/gemm_bf16_t4_asm.inc: the whole body of gemm_bf16_t4_res_kernel (gemm_bf16_t4.hip) as ONE assembly
statement -- the bf16 down-projection C = R + A.W^T + bias on a 128 x 256 tile, four waves (one per SIMD), wave tile 64 x 128 on
v_mfma_f32_16x16x32_bf16, operands global -> registers -> LDS (two tiles ahead), one LDS buffer, two barriers per 64 K values.

The schedule is static, so it is written down instruction by instruction: ONE memory instruction between two MFMAs wherever
there is one to place, every s_waitcnt counted by the issue model below (LDS and vector-memory operations return in order, so
"operation X has landed" = "at most as many operations are outstanding as were issued after X").

Registers (all clobbered by the statement):
    a[0:127]     accumulators: tile (i, j) = rows 16 i.., columns 16 j.. of the wave tile -> a[4 (i + 4 j) : +3]
                 (C^T: lane = output row r16 of the 16-row block, registers = 4 consecutive columns 4 g .. 4 g + 3)
    v[0:15]      A fragments of K step 0 (i = 0..3), v[16:47] W fragments of K step 0 (j = 0..7)
    v[48:63]     A fragments of K step 1,            v[64:95] W fragments of K step 1
    v[96:143]    the tile in flight (12 pieces of 16 B per thread: 4 of A, 8 of W)
    v[144:155]   per-piece global byte offsets;  v[156:159] scratch
    s[40:59]     buffer descriptors A, W, R, C, bias;  s[60:75] loop state and row offsets
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import asm_guards as G  # noqa: E402  (wait-state guards: the numbers live in tools/asm_hazard_lint.py)

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("PIPS_GEN_OUT", os.path.join(HERE, "..", "pips_amd", "csrc", "gemm_bf16_t4_asm.inc"))
POLICY = os.environ.get("PIPS_GEN_STORE_POLICY", "")      # tuning builds: cache-policy bits of the output stores, e.g. " sc1"

FA = [0, 48]          # A fragment base register of K step 0 / 1
FW = [16, 64]
ST = 96
VO = 144
TMP = 156
RS_A, RS_W, RS_R, RS_C, RS_B = 40, 44, 48, 52, 56
S_KT, S_SO, S_T = 60, 61, 62           # iterations left, byte offset of the tile being requested, temporary
S_RR = 64                               # s[64:67]: i * 16 * ldr * 4 (residual row-block offsets); s[68:71]: the same for C
S_CR = 68
S_LAST = 72                             # (KT - 1) * 128


class Emit:
    """Instruction list + in-order issue model of the two counters."""

    def __init__(self):
        self.lines = []
        self.lgkm = []          # outstanding LDS operations, oldest first (tags)
        self.vm = []            # outstanding vector-memory operations

    def raw(self, s):
        self.lines.append(s)

    def lds(self, s, tag):
        self.lines.append(s)
        self.lgkm.append(tag)

    def vmem(self, s, tag):
        self.lines.append(s)
        self.vm.append(tag)

    def need_lds(self, tags):
        """wait until every LDS operation in `tags` has returned"""
        idx = [k for k, t in enumerate(self.lgkm) if t in tags]
        if not idx:
            return
        left = min(len(self.lgkm) - 1 - max(idx), 15)       # (4-bit counter: waiting for a little more than needed is correct)
        self.lines.append("s_waitcnt lgkmcnt(%d)" % left)
        self.lgkm = self.lgkm[len(self.lgkm) - left:] if left else []

    def need_vm(self, tags):
        idx = [k for k, t in enumerate(self.vm) if t in tags]
        if not idx:
            return
        left = min(len(self.vm) - 1 - max(idx), 63)
        self.lines.append("s_waitcnt vmcnt(%d)" % left)
        self.vm = self.vm[len(self.vm) - left:] if left else []

    def barrier(self):
        if self.lgkm:
            self.lines.append("s_waitcnt lgkmcnt(0)")
            self.lgkm = []
        self.lines.append("s_barrier")


def acc(i, j):
    return 4 * (i + 4 * j)


def frag_read(e, ks, which, idx):
    """ds_read_b128 of A fragment i / W fragment j of K step ks"""
    if which == "a":
        reg = FA[ks] + 4 * idx
        e.lds("ds_read_b128 v[%d:%d], %%[rA%d] offset:%d" % (reg, reg + 3, ks, idx * 2048), ("fa", ks, idx))
    else:
        reg = FW[ks] + 4 * idx
        e.lds("ds_read_b128 v[%d:%d], %%[rW%d] offset:%d" % (reg, reg + 3, ks, idx * 2048), ("fw", ks, idx))


FRAG_ORDER = [("w", 0), ("a", 0), ("a", 1), ("a", 2), ("a", 3)] + [("w", j) for j in range(1, 8)]     # the order K-step MFMAs want them


def mfma(e, ks, n):
    i, j = n & 3, n >> 2
    e.need_lds({("fw", ks, j), ("fa", ks, i)})
    c = acc(i, j)
    e.raw("v_mfma_f32_16x16x32_bf16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" %
          (c, c + 3, FW[ks] + 4 * j, FW[ks] + 4 * j + 3, FA[ks] + 4 * i, FA[ks] + 4 * i + 3, c, c + 3))


def store_piece(e, s):
    e.need_vm({("st", s)})
    reg = ST + 4 * s
    wa, ww = "%[wA]", "%[wW]"
    if s < 4:
        e.lds("ds_write_b128 %s, v[%d:%d] offset:%d" % (wa, reg, reg + 3, s * 4096), ("wr", s))
    else:
        e.lds("ds_write_b128 %s, v[%d:%d] offset:%d" % (ww, reg, reg + 3, (s - 4) * 4096), ("wr", s))


def load_piece(e, s, soff):
    reg = ST + 4 * s
    rs = RS_A if s < 4 else RS_W
    e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], %s offen" % (reg, reg + 3, VO + s, rs, rs + 3, soff), ("st", s))


def descriptor(e, base, lo, hi):
    e.raw("s_mov_b32 s%d, %s" % (base, lo))
    e.raw("s_and_b32 s%d, %s, 0xffff" % (base + 1, hi))
    e.raw("s_mov_b32 s%d, 0x7fffffff" % (base + 2))
    e.raw("s_mov_b32 s%d, 0x00020000" % (base + 3))


def staging_slots():
    slots = []
    for s in range(12):
        slots += [("st", s), ("ld", s)]
    return slots


def iteration_single(e):
    """64 K values out of the ONE LDS buffer: two barriers.  (A two-buffer form with one barrier per iteration was built and
    measured: the same 1.03 us per iteration -- the loop is bound by what the wave issues, not by its barriers.)"""
    e.raw("s_add_u32 s%d, s%d, 128" % (S_SO, S_SO))
    e.raw("s_min_u32 s%d, s%d, s%d" % (S_SO, S_SO, S_LAST))
    # K step 0, first half: the fragments of K step 1 go out in between
    for n in range(16):
        mfma(e, 0, n)
        if n < 12:
            frag_read(e, 1, *FRAG_ORDER[n])
    e.barrier()                                              # every wave has read tile kt: the buffer may be overwritten
    # rest of K step 0 + first half of K step 1: next tile registers -> LDS, the one after global -> registers
    slots = staging_slots()
    k = 0
    for n in range(16, 48):
        mfma(e, n >> 5, n & 31)
        if k < len(slots):
            (store_piece(e, slots[k][1]) if slots[k][0] == "st" else load_piece(e, slots[k][1], "s%d" % S_SO))
            k += 1
    assert k == len(slots)
    e.barrier()                                              # tile kt + 1 is in LDS
    for n in range(16, 32):
        mfma(e, 1, n)
        f = n - 16
        if f < 12:
            frag_read(e, 0, *FRAG_ORDER[f])


def body(stream_bf16=False):
    """stream_bf16: the residual R and the output C are bf16 (the mixer's residual stream under autocast) -- the residual tile is
    loaded as 8-byte pieces into the (still free) fragment registers and widened into the accumulators behind the first barrier;
    the epilogue rounds to bf16 (v_cvt_pk_bf16_f32, RNE) and stores 8-byte pieces."""
    e = Emit()
    descriptor(e, RS_A, "%[alo]", "%[ahi]")
    descriptor(e, RS_W, "%[wlo]", "%[whi]")
    descriptor(e, RS_R, "%[rlo]", "%[rhi]")
    descriptor(e, RS_C, "%[clo]", "%[chi]")
    descriptor(e, RS_B, "%[blo]", "%[bhi]")
    # per-piece global offsets: piece s of A = rows lr + 32 s, of W likewise
    e.raw("v_mov_b32 v%d, %%[voA]" % VO)
    for s in range(1, 4):
        e.raw("v_add_u32 v%d, %%[passA], v%d" % (VO + s, VO + s - 1))
    e.raw("v_mov_b32 v%d, %%[voW]" % (VO + 4))
    for s in range(5, 12):
        e.raw("v_add_u32 v%d, %%[passW], v%d" % (VO + s, VO + s - 1))
    # row-block offsets of the residual / the output (i * 16 rows)
    e.raw("s_mov_b32 s%d, 0" % S_RR)
    e.raw("s_mov_b32 s%d, 0" % S_CR)
    for i in range(1, 4):
        e.raw("s_add_u32 s%d, s%d, %%[rstep]" % (S_RR + i, S_RR + i - 1))
        e.raw("s_add_u32 s%d, s%d, %%[cstep]" % (S_CR + i, S_CR + i - 1))
    e.raw("s_sub_u32 s%d, %%[kt], 1" % S_LAST)
    e.raw("s_lshl_b32 s%d, s%d, 7" % (S_LAST, S_LAST))
    # ---- tile 0 -> registers; the residual tile -> accumulators; tile 0 -> LDS; tile 1 -> registers
    for s in range(12):
        load_piece(e, s, "0")
    for j in range(8):
        for i in range(4):
            c = acc(i, j)
            if stream_bf16:       # 4 bf16 = 8 bytes -> v[2 t : 2 t + 1], t = i + 4 j (the fragment registers are free until the barrier)
                t = i + 4 * j
                e.vmem("buffer_load_dwordx2 v[%d:%d], %%[voR], s[%d:%d], s%d offen offset:%d" % (2 * t, 2 * t + 1, RS_R, RS_R + 3, S_RR + i, 32 * j),
                       ("res", i, j))
            else:
                e.vmem("buffer_load_dwordx4 a[%d:%d], %%[voR], s[%d:%d], s%d offen offset:%d" % (c, c + 3, RS_R, RS_R + 3, S_RR + i, 64 * j),
                       ("res", i, j))
    for s in range(12):
        store_piece(e, s)
    e.raw("s_min_u32 s%d, 128, s%d" % (S_SO, S_LAST))
    for s in range(12):
        load_piece(e, s, "s%d" % S_SO)
    e.barrier()
    if stream_bf16:
        # widen the residual tile into the accumulators before the fragment reads take its registers (bf16 -> fp32 = a shift / a mask)
        e.need_vm({("res", i, j) for i in range(4) for j in range(8)})
        for j in range(8):
            for i in range(4):
                c, t = acc(i, j), i + 4 * j
                e.raw("v_lshlrev_b32 v%d, 16, v%d" % (64, 2 * t))
                e.raw("v_and_b32 v%d, 0xffff0000, v%d" % (65, 2 * t))
                e.raw("v_lshlrev_b32 v%d, 16, v%d" % (66, 2 * t + 1))
                e.raw("v_and_b32 v%d, 0xffff0000, v%d" % (67, 2 * t + 1))
                for q in range(4):
                    e.raw("v_accvgpr_write_b32 a%d, v%d" % (c + q, 64 + q))
    for which, idx in FRAG_ORDER:
        frag_read(e, 0, which, idx)
    e.need_vm({("res", i, j) for i in range(4) for j in range(8)})
    e.raw("s_mov_b32 s%d, %%[kt]" % S_KT)
    e.raw("s_mov_b32 s%d, 128" % S_SO)                      # tile 1 is in flight; the loop requests tile kt + 2
    # the loop body is emitted with exactly this issue state at its head (12 fragment reads, 12 tile loads outstanding); it
    # ends in the same state, which the asserts below check
    head_lgkm, head_vm = list(e.lgkm), list(e.vm)
    e.raw("1:")
    iteration_single(e)
    assert e.lgkm == head_lgkm and e.vm == head_vm, (e.lgkm, head_lgkm, e.vm, head_vm)
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_KT, S_KT))
    e.raw("s_cmp_lg_u32 s%d, 0" % S_KT)
    e.raw("s_cbranch_scc1 1b")
    # ---- epilogue: + bias, 16-byte stores.  (the last iteration's speculative fragment reads / tile loads must have landed
    #      before their registers are reused)
    e.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e.lgkm, e.vm = [], []
    for j in range(8):
        e.vmem("buffer_load_dwordx4 v[%d:%d], %%[voB], s[%d:%d], 0 offen offset:%d" % (4 * j, 4 * j + 3, RS_B, RS_B + 3, 64 * j), ("bias", j))
    G.emit_mfma_result_guard(e.raw, "v_mfma_f32_16x16x32_bf16")    # MFMA results -> v_accvgpr_read
    for j in range(8):
        e.need_vm({("bias", j)})
        for i in range(4):
            c = acc(i, j)
            r = 32 + 4 * (i + 4 * (j & 1))                   # two alternating sets of 16 registers: a store reads its data after it issues
            for q in range(4):
                e.raw("v_accvgpr_read_b32 v%d, a%d" % (r + q, c + q))
            e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (r, r + 1, r, r + 1, 4 * j, 4 * j + 1))
            e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (r + 2, r + 3, r + 2, r + 3, 4 * j + 2, 4 * j + 3))
            if stream_bf16:
                e.raw("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (r, r, r + 1))
                e.raw("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (r + 1, r + 2, r + 3))
                e.vmem("buffer_store_dwordx2 v[%d:%d], %%[voC], s[%d:%d], s%d offen offset:%d" % (r, r + 1, RS_C, RS_C + 3, S_CR + i, 32 * j) + POLICY,
                       ("out", i, j))
            else:
                e.vmem("buffer_store_dwordx4 v[%d:%d], %%[voC], s[%d:%d], s%d offen offset:%d" % (r, r + 3, RS_C, RS_C + 3, S_CR + i, 64 * j) + POLICY,
                       ("out", i, j))
        if j >= 1:
            e.need_vm({("out", i, j - 1) for i in range(4)})   # the other register set is free again
    e.raw("s_waitcnt vmcnt(0)")
    return e.lines


def main():
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % i for i in range(128)] + ['"v%d"' % i for i in range(160)] + \
           ['"s%d"' % i for i in range(40, 76)]
    with open(OUT, "w") as f:
        f.write("// generated by tools/gen_gemm_bf16_t4.py -- do not edit\n")
        for name in ("PIPS_T4_TEXT", "PIPS_T4B_TEXT"):
            lines = body(stream_bf16=name == "PIPS_T4B_TEXT")
            f.write("#define %s \\\n" % name)
            for ln in lines:
                f.write('    "%s\\n\\t" \\\n' % ln)
            f.write('    ""\n\n')
            print("%s: %d instructions, %d MFMAs" % (name, len(lines), sum("v_mfma" in ln for ln in lines)))
        f.write("#define PIPS_T4_CLOBBER " + ", ".join(clob) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
