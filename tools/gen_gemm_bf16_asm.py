#!/usr/bin/env python
"""Emit pips_amd/csrc/gemm_bf16_tile_asm.inc: the assembly text of ONE 256 x 256 TILE (K = 512) of one wave of the persistent bf16
up-projection kernel (gemm_bf16_gelu256_asm_kernel, gemm_bf16_asm.hip) -- eight waves with 64 x 128 wave tiles on
v_mfma_f32_32x32x16_bf16, a ring of four 32-K stages of (256 + 256) rows x 64 B filled by LDS-DMA three stages ahead with one
barrier per stage, run-on into the block's next tile, and the epilogue (bf16 rounding of the Linear output, table GELU, 16-byte
stores) reading the accumulators directly.  Straight-line code, fixed registers: the C++ form of such a loop either carried ~25
scalar branches per 64 K values or, unrolled, spilled (tools/experiments/README.md).

Two texts: the ring runs on into the next tile (R1) / stops at this one (R0).  Rounds 2-3 also generated a 256 x 128 form and
the looped down-projection tiles here; round 4 moved the down-projection to tools/gen_gemm_bf16_t4.py and dropped them.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
TRACE = os.environ.get("PIPS_GEN_TRACE", "") == "1"   # tuning builds: s_memtime stamps in the lanes of %[tr]
OUT = os.environ.get("PIPS_GEN_OUT", os.path.join(HERE, "..", "pips_amd", "csrc", "gemm_bf16_tile_asm.inc"))

S_RD = 54
S_T = 56
S_TR = 58


def f32(x):
    import struct
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


class Asm:
    def __init__(self):
        self.lines = []
        self.vm = []            # vector-memory operations in issue order: ("dma", stage) / ("st",) / ("ld",)

    def __call__(self, s):
        self.lines.append(s)


def probe(a, idx):
    """trace builds: lane idx of %[tr] = low word of s_memtime (costs an lgkmcnt(0))"""
    if not TRACE:
        return
    a("s_memtime s[%d:%d]" % (S_TR, S_TR + 1))
    a("s_waitcnt lgkmcnt(0)")
    a("v_writelane_b32 %%[tr], s%d, %d" % (S_TR, idx))


FA8, FB8 = 64, 80
RA8, RB8 = 112, 113                # v113..v116: the four W fragment addresses
AX8, BX8 = 117, 118                # v117, v118..v121: K half 1 offsets
E8 = 64                            # epilogue temporaries v[64:111]
STAGE8 = 512 * 64
RING8 = 4 * STAGE8
S_S8, S_SD8, S_KO8 = 60, 61, 66    # stage index of the loop, the stage the DMA instructions bring, its K byte offset
S_BA8, S_BW8 = 68, 70              # s[68:69], s[70:71]: A / W tile base of the DMA target stage
S_TB8 = 72                         # LDS offset of the DMA target buffer
S_C64, S_C384, S_C767 = 73, 74, 75


def acc8(i, j):
    return 16 * (4 * i + j)


def fa8(kk, i):
    return FA8 + 8 * kk + 4 * i


def fb8(kk, j):
    return FB8 + 16 * kk + 4 * j


def reads8(a, kk):
    """6 fragment reads of K half kk from the stage at LDS offset s[S_RD] (the per-lane offsets carry the LDS base)"""
    a("v_add_u32 v%d, s%d, %s" % (RA8, S_RD, "%[aoff]" if kk == 0 else "v%d" % AX8))
    for j in range(4):
        a("v_add_u32 v%d, s%d, %s" % (RB8 + j, S_RD, ("%%[b%doff]" % j) if kk == 0 else "v%d" % (BX8 + j)))
    a("ds_read_b128 v[%d:%d], v%d" % (fa8(kk, 0), fa8(kk, 0) + 3, RA8))
    a("ds_read_b128 v[%d:%d], v%d" % (fb8(kk, 0), fb8(kk, 0) + 3, RB8))
    a("ds_read_b128 v[%d:%d], v%d offset:2048" % (fa8(kk, 1), fa8(kk, 1) + 3, RA8))
    for j in range(1, 4):
        a("ds_read_b128 v[%d:%d], v%d" % (fb8(kk, j), fb8(kk, j) + 3, RB8 + j))


def mfma8(a, kk, i, j):
    c = acc8(i, j)
    a("v_mfma_f32_32x32x16_bf16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" %
      (c, c + 15, fb8(kk, j), fb8(kk, j) + 3, fa8(kk, i), fa8(kk, i) + 3, c, c + 15))


def dma8_setup(a, runon):
    """the stage s + 3: its K offset and tile bases (this tile's below stage 16, the next tile's from there on), its buffer"""
    a("s_add_u32 s%d, s%d, 3" % (S_SD8, S_S8))
    a("s_and_b32 s%d, s%d, 15" % (S_T, S_SD8))
    a("s_lshl_b32 s%d, s%d, 6" % (S_KO8, S_T))
    a("s_cmp_lt_u32 s%d, 16" % S_SD8)
    a("s_cselect_b32 s%d, %%[cqa], %%[nqa]" % S_BA8)
    a("s_cselect_b32 s%d, %%[cqah], %%[nqah]" % (S_BA8 + 1))
    a("s_cselect_b32 s%d, %%[cqw], %%[nqw]" % S_BW8)
    a("s_cselect_b32 s%d, %%[cqwh], %%[nqwh]" % (S_BW8 + 1))
    a("s_add_u32 s%d, s%d, s%d" % (S_BA8, S_BA8, S_KO8))
    a("s_addc_u32 s%d, s%d, 0" % (S_BA8 + 1, S_BA8 + 1))
    a("s_add_u32 s%d, s%d, s%d" % (S_BW8, S_BW8, S_KO8))
    a("s_addc_u32 s%d, s%d, 0" % (S_BW8 + 1, S_BW8 + 1))
    a("s_add_u32 s%d, s%d, %d" % (S_TB8, S_RD, 3 * STAGE8))          # buffer of stage s + 3 = the one stage s - 1 left
    a("s_and_b32 s%d, s%d, %d" % (S_TB8, S_TB8, RING8 - 1))


def dma8(a, q):
    """piece q of this wave: 0, 1 = A rows, 2, 3 = W rows"""
    is_a = q < 2
    a("s_add_u32 s%d, s%d, %s" % (S_T, S_TB8, "%[wva]" if is_a else "%[wvw]"))
    a("s_add_u32 m0, s%d, %d" % (S_T, 1024 * (q & 1)))
    a("s_nop 0")
    a("global_load_lds_dwordx4 %%[ro%d], s[%d:%d]" % (q, (S_BA8 if is_a else S_BW8), (S_BA8 if is_a else S_BW8) + 1))


def stage8(a, dmas, vmcnt, nxt_reads):
    """one 32-K stage: the two MFMA groups with the barrier between them"""
    if dmas:
        dma8_setup(a, True)
    a("s_waitcnt lgkmcnt(6)")
    k = 0
    for i in range(2):
        for j in range(4):
            mfma8(a, 0, i, j)
            if dmas and k < 4:
                dma8(a, k)
            k += 1
    a("s_waitcnt lgkmcnt(0)")                         # K half 1 is in registers: this wave is done reading the stage
    a("s_waitcnt vmcnt(%d)" % vmcnt)                  # its pieces of the next stage have landed
    a("s_barrier")
    a("s_add_u32 s%d, s%d, %d" % (S_RD, S_RD, STAGE8))
    a("s_and_b32 s%d, s%d, %d" % (S_RD, S_RD, RING8 - 1))
    if nxt_reads:
        reads8(a, 0)
    for i in range(2):
        for j in range(4):
            mfma8(a, 1, i, j)
    if nxt_reads:
        reads8(a, 1)
    a("s_add_u32 s%d, s%d, 1" % (S_S8, S_S8))


def tile_up256(runon):
    """ONE 256 x 256 tile of the up-projection (K = 512 = 16 stages), bias in the accumulators' start, GELU epilogue.
    runon: the last three stages bring stages 0..2 of the block's next tile."""
    a = Asm()
    a("v_xor_b32 v%d, 32, %%[aoff]" % AX8)
    for j in range(4):
        a("v_xor_b32 v%d, 32, %%[b%doff]" % (BX8 + j, j))
    a("s_mov_b32 s%d, 0" % S_RD)
    a("s_mov_b32 s%d, 0" % S_S8)
    a("s_mov_b32 s%d, %s" % (S_C64, f32(64.0)))
    a("s_mov_b32 s%d, %s" % (S_C384, f32(384.0)))
    a("s_mov_b32 s%d, %s" % (S_C767, f32(767.0)))
    # accumulators start from the bias: acc[0][j][4g..] <- bias[cols], copied to acc[1][j]; column of register 4g + e of tile
    # j = 2G + jj: 64 G + (2 jj + (g >> 1)) * 16 + 4 (g & 1) + 8 half + e
    for j in range(4):
        G, jj = j >> 1, j & 1
        for g in range(4):
            off = (64 * G + (2 * jj + (g >> 1)) * 16 + 4 * (g & 1)) * 4
            r = acc8(0, j) + 4 * g
            a("global_load_dwordx4 a[%d:%d], %%[boff], %%[bias] offset:%d" % (r, r + 3, off))
    a("s_waitcnt vmcnt(0)")                           # (also: every stage issued so far has landed for this wave)
    for j in range(4):
        for r in range(16):
            a("v_accvgpr_mov_b32 a%d, a%d" % (acc8(1, j) + r, acc8(0, j) + r))
    a("s_barrier")                                    # stages 0..2 of this tile are in LDS for everyone
    reads8(a, 0)
    reads8(a, 1)
    a("1:")
    stage8(a, True, 8, True)                          # s = 0 .. 12: the DMA instructions bring stages 3 .. 15
    a("s_cmp_lt_u32 s%d, 13" % S_S8)
    a("s_cbranch_scc1 1b")
    if runon:
        stage8(a, True, 8, True)                      # s = 13, 14, 15: stages 0 .. 2 of the next tile
        stage8(a, True, 8, True)
        stage8(a, True, 8, False)
    else:
        stage8(a, False, 4, True)
        stage8(a, False, 0, True)
        stage8(a, False, 0, False)
    a("s_nop 15")
    a("s_nop 15")
    # ---- epilogue: per (i, j, q) eight consecutive columns: bf16 rounding of the Linear output, table GELU, one 16-byte store
    X, T, I, EV, VC = E8, E8 + 8, E8 + 16, E8 + 24, E8 + 40      # x[8], t[8], idx[8], table entries [8 pairs], 384.0
    a("v_mov_b32 v%d, %s" % (VC, f32(384.0)))
    for i in range(2):
        for j in range(4):
            G, jj = j >> 1, j & 1
            for q in range(2):
                for e in range(8):
                    a("v_accvgpr_read_b32 v%d, a%d" % (X + e, acc8(i, j) + 8 * q + e))
                for e in range(0, 8, 2):              # round to bf16 (hardware RNE) and widen again
                    a("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (T + e, X + e, X + e + 1))
                    a("v_lshlrev_b32 v%d, 16, v%d" % (X + e, T + e))
                    a("v_and_b32 v%d, 0xffff0000, v%d" % (X + e + 1, T + e))
                for e in range(8):
                    a("v_fma_f32 v%d, v%d, s%d, v%d" % (T + e, X + e, S_C64, VC))
                for e in range(8):
                    a("v_med3_f32 v%d, v%d, 0, s%d" % (I + e, T + e, S_C767))
                for e in range(8):
                    a("v_cvt_u32_f32 v%d, v%d" % (I + e, I + e))
                for e in range(8):
                    a("v_lshl_add_u32 v%d, v%d, 3, %%[tab]" % (X + e, I + e))
                for e in range(8):
                    a("ds_read_b64 v[%d:%d], v%d" % (EV + 2 * e, EV + 2 * e + 1, X + e))
                for e in range(8):
                    a("v_cvt_f32_u32 v%d, v%d" % (I + e, I + e))
                for e in range(8):
                    a("v_sub_f32 v%d, v%d, v%d" % (T + e, T + e, I + e))
                a("s_waitcnt lgkmcnt(0)")
                for e in range(8):
                    a("v_fma_f32 v%d, v%d, v%d, v%d" % (T + e, EV + 2 * e + 1, T + e, EV + 2 * e))
                # (the store reads its data registers after it issues: two alternating sets, v[112:119] -- the fragment
                # address registers, idle here -- so that the next piece does not overwrite them under it)
                OUT = RA8 + 4 * (q & 1)
                for e in range(0, 8, 2):
                    a("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (OUT + e // 2, T + e, T + e + 1))
                col = 64 * G + (2 * jj + q) * 16
                a("global_store_dwordx4 %%[stoff], v[%d:%d], %%[cb%d] offset:%d" % (OUT, OUT + 3, i, col * 2))
    return a


def main():
    out = ["// GENERATED by tools/gen_gemm_bf16_asm.py -- do not edit.", ""]
    for runon in (0, 1):
        a = tile_up256(runon)
        out.append("#define PIPS_TILE_TEXT_UP256_R%d \\" % runon)
        for i, ins in enumerate(a.lines):
            out.append('    "%s\\n\\t"' % ins + (" \\" if i + 1 < len(a.lines) else ""))
        out.append("")
        print("256x256 up-projection tile, runon=%d: %d instructions" % (runon, len(a.lines)))
    clob = ['"a%d"' % i for i in range(128)] + ['"v%d"' % i for i in range(64, 122)] + ['"s%d"' % i for i in range(40, 76)]
    out.append('#define PIPS_TILE_UP256_CLOBBER "memory", "scc", "vcc", ' + ", ".join(clob))
    out.append("")
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
