#!/usr/bin/env python
"""Emit pips_amd/csrc/gemm_bf16_tile_asm.inc: the assembly text of ONE 256x128 TILE (K = 512: eight super-stages of 64 K values) of one
wave of the persistent bf16 up-projection kernel (gemm_bf16_asm.hip) -- fragment reads, MFMAs, the wave's LDS-DMA
instructions of the ring three super-stages ahead, and the GELU / bf16 conversion / stores of the PREVIOUS tile (parked
as bf16 pairs in v[64:95]) between the MFMA pairs.  Straight-line code, fixed registers: the C++ form of the same loop
(gemm_bf16_dma.hip) either carried ~25 scalar branches per super-stage or, unrolled, spilled (tools/experiments/README.md).

Variants (one text each): the ring runs on into the next tile / stops at this one; with PIPS_GEN_DEFER=1 also the
forms that carry the GELU of the parked tile (the measured-slower alternative, see gemm_bf16_asm.hip); and the looped
tile of the down-projection (tile_res).

Registers (clobbered by the statement unless noted):
    a[0:63]      accumulators acc[i][j] -> a[16*(2i+j) : +15]  (C^T: lane = output row, registers = columns) -- AccVGPRs:
                 with the accumulators in ArchVGPRs the MFMA passes and the GELU's VALU instructions did not overlap
    v[64:95]     the parked tile, bf16 pairs: prev[i][j][d] -> v[64 + 8*(2i+j) + d]   (operands, live across statements)
    v[96:111]    A fragments fa[kk][i] -> v[96 + 4*(2kk+i) : +3];  v[112:127] W fragments fb[kk][j]
    v[128:139]   GELU: x0 x1 t0 t1 p0 p1 (pairs);  v[140:143] the 8 bf16 of a piece;  v[144:146] fragment addresses
    v[148:153]   a_off^32, b_off^32 ... scratch;  v[156:157] polynomial constant c4
    s[40:69]     constants, ring pointer, temporaries
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFER = os.environ.get("PIPS_GEN_DEFER", "") == "1"   # also emit the texts with the parked tile's GELU between the next tile's MFMA
                                                       # pairs (build gemm_bf16_asm.hip with -DPIPS_ASM_DEFER=1; measured slower)
TRACE = os.environ.get("PIPS_GEN_TRACE", "") == "1"   # tuning builds: s_memtime stamps in the lanes of %[tr]
OUT = os.environ.get("PIPS_GEN_OUT", os.path.join(HERE, "..", "pips_amd", "csrc", "gemm_bf16_tile_asm.inc"))

STAGE = (256 + 128) * 64           # bytes of one 32-K stage
SUP = 2 * STAGE                    # one super-stage
NSUP = 3
ACC, PREV, FA, FB = 0, 64, 96, 112
X0, X1, T0, T1, P0, P1 = 128, 130, 132, 134, 136, 138
OUTR = 140
RA, RB0, RB1 = 144, 145, 146
AX, B0X, B1X = 148, 149, 150       # the kk = 1 offsets (slot ^ 2 = byte offset ^ 32)
C4V = 156
S_C5, S_C3, S_C2, S_C1, S_C0, S_TMAX, S_MH = 40, 42, 44, 46, 48, 50, 52
S_RD, S_T, S_T2, S_TR = 54, 56, 57, 58
S_K, S_CNT = 60, 61                # looped tile: K byte offset of the current super-stage, iterations left
DV = 153                           # lane offset + K offset of a DMA instruction
# degree-5 exponent polynomial of the bf16-output GELU (gemm_bf16_dma.hip), c0..c5
COEF = [-1.150685204e+00, -4.602978599e-01, -5.192063601e-02, 7.452824686e-03, -6.529359078e-04, 2.554670494e-05]
TMAX = 5.65685425


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


def acc(i, j):
    return ACC + 16 * (2 * i + j)


def prev(hh):
    i, j, d = hh >> 3, (hh >> 2) & 1, 2 * (hh & 3)
    return PREV + 8 * (2 * i + j) + d


def fa(kk, i):
    return FA + 4 * (2 * kk + i)


def fb(kk, j):
    return FB + 4 * (2 * kk + j)


def reads(a, kk, sbase_expr_reg, stage_off):
    """4 fragment reads of K half kk from the stage at s[sbase] + stage_off"""
    a("s_add_u32 s%d, s%d, %d" % (S_T, sbase_expr_reg, stage_off))
    a("v_add_u32 v%d, s%d, %s" % (RA, S_T, "%[aoff]" if kk == 0 else "v%d" % AX))
    a("v_add_u32 v%d, s%d, %s" % (RB0, S_T, "%[b0off]" if kk == 0 else "v%d" % B0X))
    a("v_add_u32 v%d, s%d, %s" % (RB1, S_T, "%[b1off]" if kk == 0 else "v%d" % B1X))
    a("ds_read_b128 v[%d:%d], v%d" % (fa(kk, 0), fa(kk, 0) + 3, RA))
    a("ds_read_b128 v[%d:%d], v%d" % (fb(kk, 0), fb(kk, 0) + 3, RB0))
    a("ds_read_b128 v[%d:%d], v%d offset:2048" % (fa(kk, 1), fa(kk, 1) + 3, RA))
    a("ds_read_b128 v[%d:%d], v%d" % (fb(kk, 1), fb(kk, 1) + 3, RB1))


def mfma2(a, kk, i):
    for j in range(2):
        c = acc(i, j)
        a("v_mfma_f32_32x32x16_bf16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" %
          (c, c + 15, fb(kk, j), fb(kk, j) + 3, fa(kk, i), fa(kk, i) + 3, c, c + 15))


def dma(a, X, u, q, nxt):
    """one LDS-DMA instruction of super-stage X (tile-relative; >= 8: the next tile), half u, piece q of this wave.
    LDS side: s[S_T2] = the wave's piece 0 of the buffer being refilled.  The
    instruction's immediate offset would move the LDS destination as well, so the K offset is added to the lane offset."""
    tile = "n" if X >= 8 else "c"
    k_off = (X % 8) * 128 + u * 64
    a("s_add_u32 m0, s%d, %d" % (S_T2, u * STAGE + q * 1024))
    if k_off:
        a("v_add_u32 v%d, %d, %%[ro%d]" % (DV, k_off, q))
        a("global_load_lds_dwordx4 v%d, %%[%sq%d]" % (DV, tile, q))
    else:
        a("s_nop 0")
        a("global_load_lds_dwordx4 %%[ro%d], %%[%sq%d]" % (q, tile, q))
    a.vm.append(("dma", X))


def dma_k(a, ahead, u, q):
    """the same inside the K loop of the looped tile: super-stage ks + ahead, K byte offset s[S_K] + ahead*128 + u*64"""
    a("s_add_u32 m0, s%d, %d" % (S_T2, u * STAGE + q * 1024))
    a("s_add_u32 s%d, s%d, %d" % (S_T, S_K, ahead * 128 + u * 64))
    a("v_add_u32 v%d, s%d, %%[ro%d]" % (DV, S_T, q))
    a("global_load_lds_dwordx4 v%d, %%[cq%d]" % (DV, q))


def gelu_load(a, hh):
    d0, d1 = prev(hh), prev(hh) + 1
    a("v_lshlrev_b32 v%d, 16, v%d" % (X0, d0))
    a("v_and_b32 v%d, 0xffff0000, v%d" % (X0 + 1, d0))
    a("v_lshlrev_b32 v%d, 16, v%d" % (X1, d1))
    a("v_and_b32 v%d, 0xffff0000, v%d" % (X1 + 1, d1))


def gelu_step(a, n):
    pairs = [(X0, T0, P0), (X1, T1, P1)]
    if n == 1:
        for x, t, p in pairs:
            a("v_min_f32_e64 v%d, |v%d|, s%d" % (t, x, S_TMAX))
            a("v_min_f32_e64 v%d, |v%d|, s%d" % (t + 1, x + 1, S_TMAX))
        for x, t, p in pairs:
            a("v_pk_fma_f32 v[%d:%d], v[%d:%d], s[%d:%d], v[%d:%d] op_sel_hi:[1,0,1]" % (p, p + 1, t, t + 1, S_C5, S_C5 + 1, C4V, C4V + 1))
        for x, t, p in pairs:
            a("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], s[%d:%d] op_sel_hi:[1,1,0]" % (p, p + 1, p, p + 1, t, t + 1, S_C3, S_C3 + 1))
        for x, t, p in pairs:
            a("v_max_f32_e32 v%d, 0, v%d" % (x, x))
            a("v_max_f32_e32 v%d, 0, v%d" % (x + 1, x + 1))
    elif n == 2:
        for s_c in (S_C2, S_C1, S_C0):
            for x, t, p in pairs:
                a("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], s[%d:%d] op_sel_hi:[1,1,0]" % (p, p + 1, p, p + 1, t, t + 1, s_c, s_c + 1))
        for x, t, p in pairs:
            a("v_pk_mul_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (p, p + 1, p, p + 1, t, t + 1))
    elif n == 3:
        for x, t, p in pairs:
            a("v_exp_f32_e32 v%d, v%d" % (p, p))
            a("v_exp_f32_e32 v%d, v%d" % (p + 1, p + 1))
        a("s_nop 0")
        for x, t, p in pairs:
            a("v_pk_mul_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (p, p + 1, p, p + 1, t, t + 1))
    else:
        for x, t, p in pairs:
            a("v_pk_fma_f32 v[%d:%d], v[%d:%d], s[%d:%d], v[%d:%d] op_sel_hi:[1,0,1]" % (p, p + 1, p, p + 1, S_MH, S_MH + 1, x, x + 1))


def gelu_pack(a, dst):
    a("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (dst, P0, P0 + 1))
    a("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (dst + 1, P1, P1 + 1))


def vm_wait(a, need_stage):
    """s_waitcnt vmcnt(N): everything up to the last load of super-stage need_stage has landed"""
    last = max([k for k, op in enumerate(a.vm) if op == ("dma", need_stage)], default=-1)
    n = len(a.vm) - 1 - last
    a("s_waitcnt vmcnt(%d)" % min(n, 63))


def tile(gel, runon):
    """gel: the parked tile's GELU pieces are issued; runon: super-stages 8..10 (next tile) are opened"""
    a = Asm()
    # ---- constants
    for s, v in ((S_C5, COEF[5]), (S_C3, COEF[3]), (S_C2, COEF[2]), (S_C1, COEF[1]), (S_C0, COEF[0]), (S_TMAX, TMAX), (S_MH, -0.5)):
        a("s_mov_b32 s%d, %s" % (s, f32(v)))
    a("v_mov_b32 v%d, %s" % (C4V, f32(COEF[4])))
    a("v_mov_b32 v%d, %s" % (C4V + 1, f32(COEF[4])))
    a("v_xor_b32 v%d, 32, %%[aoff]" % AX)
    a("v_xor_b32 v%d, 32, %%[b0off]" % B0X)
    a("v_xor_b32 v%d, 32, %%[b1off]" % B1X)
    a("s_mov_b32 s%d, %%[rd]" % S_RD)                 # LDS address of the super-stage this tile starts with
    # ---- accumulators start from the bias: acc[0][j][4g..] <- bias[cols], copied to acc[1][j]
    probe(a, 0)
    for j in range(2):
        for g in range(4):
            off = ((2 * j + (g >> 1)) * 16 + 4 * (g & 1)) * 4
            r = acc(0, j) + 4 * g
            a("global_load_dwordx4 a[%d:%d], %%[boff], %%[bias] offset:%d" % (r, r + 3, off))
    a("s_waitcnt vmcnt(0)")                           # (also: every super-stage issued so far has landed for this wave)
    probe(a, 1)
    for j in range(2):
        for r in range(16):
            a("v_accvgpr_mov_b32 a%d, a%d" % (acc(1, j) + r, acc(0, j) + r))
    # ---- fragments of K block a of super-stage 0 (landed and published by the previous tile's last barrier)
    reads(a, 0, S_RD, 0)
    reads(a, 1, S_RD, 0)
    # the second half of super-stage 2 is still to be issued (its first half went out in the previous statement / the
    # C++ prologue); its buffer: two ahead of the read buffer
    for ks in range(8):
        Xa = ks + 2                                   # block a issues the second half of this stage (opened last iteration)
        Xb = ks + 3                                   # block b opens this one
        do_a = Xa <= 7 or runon
        do_b = Xb <= 7 or runon
        hh = 2 * ks
        probe(a, 2 + 4 * ks)
        # ================= K block a
        if do_a:                                      # LDS base of stage Xa's buffer + this wave's piece: (rd + 2 SUP) mod ring
            a("s_add_u32 s%d, s%d, %d" % (S_T2, S_RD, 2 * SUP))
            a("s_sub_u32 s%d, s%d, %d" % (S_T, S_T2, NSUP * SUP))
            a("s_cmp_ge_u32 s%d, %%[ringend]" % S_T2)
            a("s_cselect_b32 s%d, s%d, s%d" % (S_T2, S_T, S_T2))
            a("s_add_u32 s%d, s%d, %%[wvoff]" % (S_T2, S_T2))
        if gel:
            gelu_load(a, hh)
        a("s_waitcnt lgkmcnt(4)")
        mfma2(a, 0, 0)
        if do_a:
            dma(a, Xa, 1, 0, Xa >= 8)
        if gel:
            gelu_step(a, 1)
        probe(a, 44 + ks)
        mfma2(a, 0, 1)
        if do_a:
            dma(a, Xa, 1, 1, Xa >= 8)
        if gel:
            gelu_step(a, 2)
        a("s_waitcnt lgkmcnt(0)")
        reads(a, 0, S_RD, STAGE)                      # block b, kk = 0
        mfma2(a, 1, 0)
        if do_a:
            dma(a, Xa, 1, 2, Xa >= 8)
        if gel:
            gelu_step(a, 3)
        mfma2(a, 1, 1)
        if gel:
            gelu_step(a, 4)
            gelu_pack(a, OUTR)
        reads(a, 1, S_RD, STAGE)                      # block b, kk = 1
        # ================= K block b
        probe(a, 3 + 4 * ks)
        if gel:
            gelu_load(a, hh + 1)
        a("s_waitcnt lgkmcnt(4)")
        mfma2(a, 0, 0)
        if gel:
            gelu_step(a, 1)
        mfma2(a, 0, 1)
        if gel:
            gelu_step(a, 2)
        a("s_waitcnt lgkmcnt(0)")                     # this wave is done reading super-stage ks
        probe(a, 4 + 4 * ks)
        if ks < 7 or runon:
            vm_wait(a, ks + 1)                        # its part of super-stage ks+1 has landed
        else:
            a("s_waitcnt vmcnt(0)")
        probe(a, 36 + ks)
        a("s_barrier")
        probe(a, 5 + 4 * ks)
        if do_b:                                      # the buffer super-stage ks just left is refilled with stage ks+3
            a("s_add_u32 s%d, s%d, %%[wvoff]" % (S_T2, S_RD))
        # advance the ring pointer
        a("s_add_u32 s%d, s%d, %d" % (S_RD, S_RD, SUP))
        a("s_cmp_ge_u32 s%d, %%[ringend]" % S_RD)
        a("s_cselect_b32 s%d, %%[lds0], s%d" % (S_RD, S_RD))
        if ks < 7:
            reads(a, 0, S_RD, 0)                      # next block a, kk = 0
        mfma2(a, 1, 0)
        if do_b:
            dma(a, Xb, 0, 0, Xb >= 8)
            dma(a, Xb, 0, 1, Xb >= 8)
        if gel:
            gelu_step(a, 3)
        mfma2(a, 1, 1)
        if do_b:
            dma(a, Xb, 0, 2, Xb >= 8)
        if gel:
            gelu_step(a, 4)
            gelu_pack(a, OUTR + 2)
            i, jq = ks >> 2, ks & 3
            a("global_store_dwordx4 %%[stoff], v[%d:%d], %%[cb%d] offset:%d" % (OUTR, OUTR + 3, i, jq * 32))
            a.vm.append(("st",))
        if ks < 7:
            reads(a, 1, S_RD, 0)                      # next block a, kk = 1
    probe(a, 34)
    # ---- park the tile: prev <- bf16 pairs of the accumulators (the MFMAs have to have written them back)
    a("s_nop 15")
    a("s_nop 15")
    for i in range(2):
        for j in range(2):
            for d in range(8):
                a("v_accvgpr_read_b32 v%d, a%d" % (X0 + 2 * d, acc(i, j) + 2 * d))
                a("v_accvgpr_read_b32 v%d, a%d" % (X0 + 2 * d + 1, acc(i, j) + 2 * d + 1))
            for d in range(8):
                a("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (PREV + 8 * (2 * i + j) + d, X0 + 2 * d, X0 + 2 * d + 1))
    probe(a, 35)
    return a


def super_block(a, do_a, do_b, nxt_reads, vmcnt):
    """one super-stage of the looped tile (no GELU): K blocks a and b"""
    if do_a:
        a("s_add_u32 s%d, s%d, %d" % (S_T2, S_RD, 2 * SUP))
        a("s_sub_u32 s%d, s%d, %d" % (S_T, S_T2, NSUP * SUP))
        a("s_cmp_ge_u32 s%d, %%[ringend]" % S_T2)
        a("s_cselect_b32 s%d, s%d, s%d" % (S_T2, S_T, S_T2))
        a("s_add_u32 s%d, s%d, %%[wvoff]" % (S_T2, S_T2))
    a("s_waitcnt lgkmcnt(4)")
    mfma2(a, 0, 0)
    if do_a:
        dma_k(a, 2, 1, 0)
    mfma2(a, 0, 1)
    if do_a:
        dma_k(a, 2, 1, 1)
    a("s_waitcnt lgkmcnt(0)")
    reads(a, 0, S_RD, STAGE)
    mfma2(a, 1, 0)
    if do_a:
        dma_k(a, 2, 1, 2)
    mfma2(a, 1, 1)
    reads(a, 1, S_RD, STAGE)
    a("s_waitcnt lgkmcnt(4)")
    mfma2(a, 0, 0)
    mfma2(a, 0, 1)
    a("s_waitcnt lgkmcnt(0)")
    a("s_waitcnt vmcnt(%d)" % vmcnt)
    a("s_barrier")
    if do_b:
        a("s_add_u32 s%d, s%d, %%[wvoff]" % (S_T2, S_RD))
    a("s_add_u32 s%d, s%d, %d" % (S_RD, S_RD, SUP))
    a("s_cmp_ge_u32 s%d, %%[ringend]" % S_RD)
    a("s_cselect_b32 s%d, %%[lds0], s%d" % (S_RD, S_RD))
    if nxt_reads:
        reads(a, 0, S_RD, 0)
    mfma2(a, 1, 0)
    if do_b:
        dma_k(a, 3, 0, 0)
        dma_k(a, 3, 0, 1)
    mfma2(a, 1, 1)
    if do_b:
        dma_k(a, 3, 0, 2)
    if nxt_reads:
        reads(a, 1, S_RD, 0)


BIASV = 160                        # v[160:191]: the wave's 64 bias values (natural column order), looped tile


def tile_res():
    """ONE 256x128 tile of the down-projection: K loop over %[nks] super-stages (>= 4), accumulators start from the
    residual tile, bias added at the end, fp32 stores.  Natural column order (lane = row, registers 4g..4g+3 = columns
    j*32 + 8g + 4*half ..+3): 16-byte stores, 32 contiguous bytes per row."""
    a = Asm()
    a("v_xor_b32 v%d, 32, %%[aoff]" % AX)
    a("v_xor_b32 v%d, 32, %%[b0off]" % B0X)
    a("v_xor_b32 v%d, 32, %%[b1off]" % B1X)
    a("s_mov_b32 s%d, %%[rd]" % S_RD)
    for i in range(2):
        for j in range(2):
            for g in range(4):
                r = acc(i, j) + 4 * g
                a("global_load_dwordx4 a[%d:%d], %%[roff], %%[rb%d] offset:%d" % (r, r + 3, i, (j * 32 + 8 * g) * 4))
    for j in range(2):
        for g in range(4):
            r = BIASV + 16 * j + 4 * g
            a("global_load_dwordx4 v[%d:%d], %%[boff], %%[bias] offset:%d" % (r, r + 3, (j * 32 + 8 * g) * 4))
    a("s_waitcnt vmcnt(0)")                           # the residual tile, the bias AND the C++ prologue's three super-stages
    a("s_barrier")                                    # (their round trips overlap; the barrier publishes the stages)
    reads(a, 0, S_RD, 0)
    reads(a, 1, S_RD, 0)
    a("s_mov_b32 s%d, 0" % S_K)
    a("s_sub_u32 s%d, %%[nks], 3" % S_CNT)
    a("1:")
    super_block(a, True, True, True, 2 * 3)
    a("s_add_u32 s%d, s%d, 128" % (S_K, S_K))
    a("s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT))
    a("s_cmp_lg_u32 s%d, 0" % S_CNT)
    a("s_cbranch_scc1 1b")
    super_block(a, True, False, True, 2 * 3)          # ks = nks-3: the second half of the last super-stage goes out
    super_block(a, False, False, True, 0)             # ks = nks-2
    super_block(a, False, False, False, 0)            # ks = nks-1
    a("s_nop 15")
    a("s_nop 15")
    for i in range(2):
        for j in range(2):
            for r in range(16):
                a("v_accvgpr_read_b32 v%d, a%d" % (X0 + r, acc(i, j) + r))
            for r in range(16):
                a("v_add_f32 v%d, v%d, v%d" % (X0 + r, X0 + r, BIASV + 16 * j + r))
            for g in range(4):
                a("global_store_dwordx4 %%[soff], v[%d:%d], %%[cb%d] offset:%d" % (X0 + 4 * g, X0 + 4 * g + 3, i, (j * 32 + 8 * g) * 4))
    return a


# ------------------------------------------------------------------------------------------------------------------------------
# The down-projection with FOUR waves per block (one per SIMD), wave tile 128 x 64, and a ring of three 64-K STAGES whose LDS
# rows are 128 bytes: a DMA instruction brings 8 rows x 128 B (eight FULL cache lines) where the 32-K stages of the tiles above
# bring 16 rows x 64 B.  Measured (tools/bf16_res_probe.py, the 4-wave tile with 64-byte rows): the K loop's time is the number of
# vector-memory wave-instructions x ~52 clocks whatever they carry -- adding 6 (12) one-dword "touch" loads per super-stage to the 12
# DMA instructions took the loop from 1.08 to 1.56 (2.1) us per super-stage -- i.e. the texture addresser spends ~3.3 clocks per
# 64-byte row segment and delivers 47 GB/s per CU, while fully contiguous 1 KiB instructions reach 82 GB/s (tools/l2_stream_rate.hip).
# All per-iteration address arithmetic of the DMA is scalar.
# Registers: a[0:127] accumulators acc4(i, j) = 16 * (2i + j), i < 4; v[96:127] A fragments fa4(kk, i); v[128:143] W fragments;
# v[144:146] fragment addresses; v[148:156] the K-quarter offsets 1..3 of A / W0 / W1; v[160:175] epilogue; v[176:207] bias.
FA4, FB4 = 96, 128
X4, BIAS4 = 160, 176
OFFQ = 148                         # v[148 + 3*(kq-1) + {0,1,2}]: aoff / b0off / b1off ^ (kq * 32), kq = 1..3
S_Q = 62                           # s[62:63]: 64-bit source base of a DMA instruction
S_TA, S_TW = 64, 65                # LDS address of the target stage + this wave's A / W piece 0
NA4, NW4 = 8, 4                    # DMA pieces per wave and stage: 8 of A (8 rows x 128 B each), 4 of W
SUP4 = (256 + 128) * 128           # bytes of a 64-K stage


def acc4(i, j):
    return 16 * (2 * i + j)


def fa4(kk, i):
    return FA4 + 16 * kk + 4 * i


def fb4(kk, j):
    return FB4 + 8 * kk + 4 * j


def off4(kq, which):
    """register (or operand) holding the per-lane fragment offset of K quarter kq: which = 0 A, 1 W tile 0, 2 W tile 1"""
    if kq == 0:
        return ("%[aoff]", "%[b0off]", "%[b1off]")[which]
    return "v%d" % (OFFQ + 3 * (kq - 1) + which)


ABL = int(os.environ.get("PIPS_GEN_ABL", "0"))     # tuning: 1 = no fragment reads, 2 = no DMA in the loop, 4 = no MFMAs (4-wave tile)


def reads4(a, kq, nxt=False):
    """6 fragment reads of K quarter kq (fragment set kq & 1) from the stage at s[S_RD]"""
    kk = kq & 1
    if ABL & 1:
        return
    a("v_add_u32 v%d, s%d, %s" % (RA, S_RD, off4(kq, 0)))
    a("v_add_u32 v%d, s%d, %s" % (RB0, S_RD, off4(kq, 1)))
    a("v_add_u32 v%d, s%d, %s" % (RB1, S_RD, off4(kq, 2)))
    a("ds_read_b128 v[%d:%d], v%d" % (fa4(kk, 0), fa4(kk, 0) + 3, RA))
    a("ds_read_b128 v[%d:%d], v%d" % (fb4(kk, 0), fb4(kk, 0) + 3, RB0))
    a("ds_read_b128 v[%d:%d], v%d offset:4096" % (fa4(kk, 1), fa4(kk, 1) + 3, RA))
    a("ds_read_b128 v[%d:%d], v%d" % (fb4(kk, 1), fb4(kk, 1) + 3, RB1))
    a("ds_read_b128 v[%d:%d], v%d offset:8192" % (fa4(kk, 2), fa4(kk, 2) + 3, RA))
    a("ds_read_b128 v[%d:%d], v%d offset:12288" % (fa4(kk, 3), fa4(kk, 3) + 3, RA))


def mfma4(a, kk, i, j):
    c = acc4(i, j)
    if ABL & 4:
        return
    a("v_mfma_f32_32x32x16_bf16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" %
      (c, c + 15, fb4(kk, j), fb4(kk, j) + 3, fa4(kk, i), fa4(kk, i) + 3, c, c + 15))


def dma4(a, ahead, p):
    """piece p (0..7: A, 8..11: W) of this wave, stage ks + ahead: LDS side s[S_TA] / s[S_TW] + 1024 * index (through m0), source
    %[cqa] / %[cqw] + s[S_K] + ahead*128 (scalar), lane offsets %[ro<p>]"""
    is_a = p < NA4
    if ABL & 2:
        return
    a("s_add_u32 m0, s%d, %d" % (S_TA if is_a else S_TW, (p if is_a else p - NA4) * 1024))
    a("s_add_u32 s%d, s%d, %d" % (S_T, S_K, ahead * 128))
    a("s_add_u32 s%d, %%[%s], s%d" % (S_Q, "cqa" if is_a else "cqw", S_T))
    a("s_addc_u32 s%d, %%[%s], 0" % (S_Q + 1, "cqah" if is_a else "cqwh"))
    a("global_load_lds_dwordx4 %%[ro%d], s[%d:%d]" % (p, S_Q, S_Q + 1))


def mfma_group4(a, kk, dmas):
    """the 8 MFMAs of one K quarter (fragment set kk); dmas: list of (ahead, piece) issued one behind each of the first MFMAs"""
    k = 0
    for i in range(4):
        for j in range(2):
            mfma4(a, kk, i, j)
            if k < len(dmas):
                dma4(a, *dmas[k])
            k += 1


def target4(a, ahead_buffers):
    """s[S_TA], s[S_TW] <- the buffer ahead_buffers stages further round the ring than s[S_RD] (+ this wave's piece offsets)"""
    a("s_add_u32 s%d, s%d, %d" % (S_T2, S_RD, ahead_buffers * SUP4))
    a("s_sub_u32 s%d, s%d, %d" % (S_T, S_T2, NSUP * SUP4))
    a("s_cmp_ge_u32 s%d, %%[ringend]" % S_T2)
    a("s_cselect_b32 s%d, s%d, s%d" % (S_T2, S_T, S_T2))
    a("s_add_u32 s%d, s%d, %%[wvoffa]" % (S_TA, S_T2))
    a("s_add_u32 s%d, s%d, %%[wvoffw]" % (S_TW, S_T2))


FIRST4 = list(range(0, 6))         # the pieces of a stage that go out right behind the barrier that frees its buffer ...
SECOND4 = list(range(6, 12))       # ... and the ones that follow in the next iteration


def super_block4(a, do_a, do_b, nxt_reads, vmcnt):
    """one 64-K stage of the 4-wave tile: K quarters 0..3, fragment sets alternate"""
    if do_a:
        target4(a, 2)
    a("s_waitcnt lgkmcnt(6)")
    mfma_group4(a, 0, [(2, p) for p in SECOND4] if do_a else [])
    a("s_waitcnt lgkmcnt(0)")
    reads4(a, 2)
    mfma_group4(a, 1, [])
    reads4(a, 3)
    a("s_waitcnt lgkmcnt(6)")
    mfma_group4(a, 0, [])
    a("s_waitcnt lgkmcnt(0)")
    a("s_waitcnt vmcnt(%d)" % vmcnt)
    a("s_barrier")
    if do_b:                                          # the buffer this stage just left is refilled with stage ks + 3
        a("s_add_u32 s%d, s%d, %%[wvoffa]" % (S_TA, S_RD))
        a("s_add_u32 s%d, s%d, %%[wvoffw]" % (S_TW, S_RD))
    a("s_add_u32 s%d, s%d, %d" % (S_RD, S_RD, SUP4))
    a("s_cmp_ge_u32 s%d, %%[ringend]" % S_RD)
    a("s_cselect_b32 s%d, %%[lds0], s%d" % (S_RD, S_RD))
    if nxt_reads:
        reads4(a, 0)
    mfma_group4(a, 1, [(3, p) for p in FIRST4] if do_b else [])
    if nxt_reads:
        reads4(a, 1)


def tile_res4():
    """ONE 256x128 tile of the down-projection on four waves (wave tile 128 x 64): K loop over %[nks] stages of 64 (>= 4),
    accumulators start from the residual tile, bias added at the end, fp32 stores, natural column order."""
    a = Asm()
    for kq in (1, 2, 3):
        for which, name in enumerate(("%[aoff]", "%[b0off]", "%[b1off]")):
            a("v_xor_b32 v%d, %d, %s" % (OFFQ + 3 * (kq - 1) + which, kq * 32, name))
    a("s_mov_b32 s%d, %%[rd]" % S_RD)
    for i in range(4):
        for j in range(2):
            for g in range(4):
                r = acc4(i, j) + 4 * g
                a("global_load_dwordx4 a[%d:%d], %%[roff], %%[rb%d] offset:%d" % (r, r + 3, i, (j * 32 + 8 * g) * 4))
    for j in range(2):
        for g in range(4):
            r = BIAS4 + 16 * j + 4 * g
            a("global_load_dwordx4 v[%d:%d], %%[boff], %%[bias] offset:%d" % (r, r + 3, (j * 32 + 8 * g) * 4))
    a("s_waitcnt vmcnt(0)")                           # the residual tile, the bias AND the C++ prologue's stages
    a("s_barrier")
    reads4(a, 0)
    reads4(a, 1)
    a("s_mov_b32 s%d, 0" % S_K)
    a("s_sub_u32 s%d, %%[nks], 3" % S_CNT)
    a("1:")
    super_block4(a, True, True, True, 12)
    a("s_add_u32 s%d, s%d, 128" % (S_K, S_K))
    a("s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT))
    a("s_cmp_lg_u32 s%d, 0" % S_CNT)
    a("s_cbranch_scc1 1b")
    super_block4(a, True, False, True, 12)            # ks = nks-3: the second half of the last stage goes out
    super_block4(a, False, False, True, 0)            # ks = nks-2
    super_block4(a, False, False, False, 0)           # ks = nks-1
    a("s_nop 15")
    a("s_nop 15")
    for i in range(4):
        for j in range(2):
            for r in range(16):
                a("v_accvgpr_read_b32 v%d, a%d" % (X4 + r, acc4(i, j) + r))
            for r in range(16):
                a("v_add_f32 v%d, v%d, v%d" % (X4 + r, X4 + r, BIAS4 + 16 * j + r))
            for g in range(4):
                a("global_store_dwordx4 %%[soff], v[%d:%d], %%[cb%d] offset:%d" % (X4 + 4 * g, X4 + 4 * g + 3, i, (j * 32 + 8 * g) * 4))
    return a


# ------------------------------------------------------------------------------------------------------------------------------
# The up-projection on 256 x 256 TILES (eight waves, wave tile 64 x 128): per MFMA a third fewer operand bytes cross L2 -> LDS than
# with the 256 x 128 tile (512 instead of 768 DMA instructions per 2048 MFMAs) and a quarter fewer fragment reads (6 per 8 MFMAs
# instead of 4 per 4).  Ring of FOUR 32-K stages of (256 + 256) rows x 64 B (128 KiB) running three stages ahead, one barrier per
# stage (between the two MFMA groups of the stage: the fragments of K half 1 are in registers by then), run-on into the block's
# next tile; no parked tile -- the epilogue (bf16 rounding of the Linear output, table GELU, 16-byte stores) reads the accumulators.
# Registers: a[0:127] acc8(i, j) = 16 * (4i + j), i < 2, j < 4; v[64:79] A fragments, v[80:111] W fragments, v[112:116] fragment
# addresses, v[117:121] the K-half-1 offsets, v[64:111] again in the epilogue; s[40:75].
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
    for gel in ((0, 1) if DEFER else (0,)):
        for runon in (0, 1):
            a = tile(gel, runon)
            out.append("#define PIPS_TILE_TEXT_G%d_R%d \\" % (gel, runon))
            for i, ins in enumerate(a.lines):
                out.append('    "%s\\n\\t"' % ins + (" \\" if i + 1 < len(a.lines) else ""))
            out.append("")
            print("variant gelu=%d runon=%d: %d instructions" % (gel, runon, len(a.lines)))
    clob = ['"a%d"' % i for i in range(64)] + ['"v%d"' % i for i in range(96, 160)] + ['"s%d"' % i for i in range(40, 62)]
    out.append('#define PIPS_TILE_CLOBBER "memory", "scc", "vcc", ' + ", ".join(clob))
    out.append("")
    a = tile_res()
    out.append("#define PIPS_TILE_TEXT_RES \\")
    for i, ins in enumerate(a.lines):
        out.append('    "%s\\n\\t"' % ins + (" \\" if i + 1 < len(a.lines) else ""))
    out.append("")
    print("looped residual tile: %d instructions" % len(a.lines))
    clob = ['"a%d"' % i for i in range(64)] + ['"v%d"' % i for i in range(96, 192)] + ['"s%d"' % i for i in range(40, 62)]
    out.append('#define PIPS_TILE_RES_CLOBBER "memory", "scc", "vcc", ' + ", ".join(clob))
    out.append("")
    a = tile_res4()
    out.append("#define PIPS_TILE_TEXT_RES4 \\")
    for i, ins in enumerate(a.lines):
        out.append('    "%s\\n\\t"' % ins + (" \\" if i + 1 < len(a.lines) else ""))
    out.append("")
    print("looped residual tile, four waves: %d instructions" % len(a.lines))
    clob = ['"a%d"' % i for i in range(128)] + ['"v%d"' % i for i in range(96, 208)] + ['"s%d"' % i for i in range(40, 66)]
    out.append('#define PIPS_TILE_RES4_CLOBBER "memory", "scc", "vcc", ' + ", ".join(clob))
    out.append("")
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
