#!/usr/bin/env python
"""Emit pips_amd/csrc/conv_bf16_t4c_asm.inc: the whole body of conv3x3_c96_t4_kernel (conv_bf16_t4c.hip) as ONE assembly
statement -- the 3x3 / stride 1 / pad 1 convolution 96 -> 96 channels on bf16 channel-last maps (the three big layer-2 convolutions
of the bf16 encoder) as an implicit GEMM in the style of tools/gen_gemm_bf16_t4up.py: a 256-pixel x 96-channel tile on four waves
(64 pixels x 96 channels each, v_mfma_f32_16x16x32_bf16, 96 AccVGPR accumulators), operands global -> registers -> LDS two taps
ahead, one LDS buffer, two barriers per tap, a block walking every `pstep`-th tile of ONE frame with the tap pipeline running on
across the tile boundary.

With pixels numbered row-major inside a frame, tap (kh, kw) of output pixel p reads input pixel p + (kh - 1) W + (kw - 1): the A
operand of a tile and tap is ONE contiguous run of 256 pixels x 192 bytes -- a straight copy (piece q = tid + 256 s -> 16 bytes at
16 q), not a gather.  What the linear view gets wrong is fixed by the buffer descriptor and two flags per piece: rows above /
below the image are offsets outside [0, M 192) of the frame's descriptor (reads return zero, stores are dropped -- the ragged last
tile too), and a piece whose pixel sits in image column 0 (W - 1) is sent out of range for the kw = 0 (kw = 2) taps.

A tap is three K steps of 32 channels on two fragment register sets (p, !p, p; the next tap starts on !p, so two tiles -- 2 x 27
K steps -- are unrolled).  Slots of a tap, one per MFMA (72): K step 0: the last 2 fragments of its own set and the 10 of K step 1;
K step 1: the 10 of K step 2, barrier A (every wave has read the tap out of LDS), then the staging of tap + 1 (registers -> LDS)
alternating with the loads of tap + 2 -- 34 operations in 24 slots (the first 10 doubled); K step 2: barrier B, the first 8
fragments of the next tap.

Registers (all clobbered):
    a[0:95]      accumulators: tile (i, j) = pixels 16 i.., channels 16 j.. of the wave tile -> a[4 (i + 4 j) : +3]   (C^T: lane = pixel)
    v[0:15] / v[16:39]   pixel / weight fragments, set 0;  v[40:55] / v[56:79] set 1   (the idle set: the epilogue's temporaries)
    v[80:127]    the A pieces in flight (12 x 16 B per thread), v[128:147] the W pieces (5: 1152 pieces on 256 threads)
    v[148:159] A piece offsets, v[160:171] their LDS addresses, v[172:176] / v[177:181] the same for W, v[182:193] the pixel of a
    piece inside the tile, v[194:205] column flags of the pieces (bit 0: column 0, bit 1: column W - 1), v[206:208] scratch,
    v209 an offset outside every descriptor, v[210:217] the epilogue's store registers (two sets)
    s[40:55] descriptors A, W, C, bias;  s[56:64] tap offsets;  s[65:77] loop state
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import asm_guards as G  # noqa: E402  (wait-state guards: the numbers live in tools/asm_hazard_lint.py)

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("PIPS_GEN_OUT", os.path.join(HERE, "..", "pips_amd", "csrc", "conv_bf16_t4c_asm.inc"))

NI, NJ, NKS, NTAP = 4, 6, 3, 9      # 16-pixel / 16-channel blocks of the wave tile, K steps (32 channels) per tap, taps
FA = [0, 40]
FW = [16, 56]
ST = 80
NPA, NPW = 12, 5
VOA, LDA, VOW, LDW, PIX, FLG = 148, 160, 172, 177, 182, 194
TMP, VOOB, OREG = 206, 209, 210
RS_A, RS_W, RS_C, RS_B = 40, 44, 48, 52
S_TAP = 56                          # s[56:64]: ((kh - 1) W + (kw - 1)) * 192 of the nine taps, as wrapping unsigned numbers
S_M0, S_TL, S_T, S_T2, S_NEXT, S_WOFF = 65, 66, 67, 68, 69, 70
S_P, S_PNEXT = 76, 77               # first pixel of this / the next tile (S_M0 / S_NEXT: the same x 192 bytes)
TAB = 46                            # per-thread table entries the kernel leaves in LDS: entry k of thread t at tab + 1024 k + 4 t


class Emit:
    """Instruction list + in-order issue model of the two counters (see gen_gemm_bf16_t4.py)."""

    def __init__(self):
        self.lines, self.lgkm, self.vm = [], [], []

    def raw(self, s):
        self.lines.append(s)

    def lds(self, s, tag):
        self.lines.append(s)
        self.lgkm.append(tag)

    def vmem(self, s, tag):
        self.lines.append(s)
        self.vm.append(tag)

    def need_lds(self, tags):
        idx = [k for k, t in enumerate(self.lgkm) if t in tags]
        if not idx:
            return
        left = min(len(self.lgkm) - 1 - max(idx), 15)
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

    def drain(self):
        self.lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.lgkm, self.vm = [], []


def acc(i, j):
    return 4 * (i + NI * j)


# the order a K step's MFMAs want the fragments: W0, A0..A3, W1..W5
FRAG_ORDER = [("w", 0)] + [("a", i) for i in range(NI)] + [("w", j) for j in range(1, NJ)]


def frag_read(e, par, ks, which, idx):
    """fragment `idx` (A = pixel block i, W = channel block j) of K step ks of the tap in LDS, into register set `par`"""
    if which == "a":
        reg = FA[par] + 4 * idx
        e.lds("ds_read_b128 v[%d:%d], %%[rA%d] offset:%d" % (reg, reg + 3, ks, idx * 4096), ("fa", par, idx))
    else:
        reg = FW[par] + 4 * idx
        e.lds("ds_read_b128 v[%d:%d], %%[rW%d] offset:%d" % (reg, reg + 3, ks, idx * 4096), ("fw", par, idx))


def mfma(e, par, n, zero):
    i, j = n % NI, n // NI
    e.need_lds({("fw", par, j), ("fa", par, i)})
    c = acc(i, j)
    src_c = "0" if zero else "a[%d:%d]" % (c, c + 3)         # a tile's first K step starts from zero
    e.raw("v_mfma_f32_16x16x32_bf16 a[%d:%d], v[%d:%d], v[%d:%d], %s" %
          (c, c + 3, FW[par] + 4 * j, FW[par] + 4 * j + 3, FA[par] + 4 * i, FA[par] + 4 * i + 3, src_c))


def store_piece(e, s):
    e.need_vm({("st", s)})
    reg = ST + 4 * s
    addr = LDA + s if s < NPA else LDW + (s - NPA)
    e.lds("ds_write_b128 v%d, v[%d:%d]" % (addr, reg, reg + 3), ("wr", s))


def load_piece(e, s, kw):
    """piece s of the tap whose tile + tap byte offset sits in s[S_T] (A) / whose weight slice in s[S_WOFF] (W); kw = the tap's column
    (0 / 2: the flagged pieces go out of range).  The A offset is formed in a VECTOR register: a raw buffer's range check covers
    the vector offset only, and the zero padding above / below the image and behind the last pixel rides on that check."""
    reg = ST + 4 * s
    if s < NPA:
        off = TMP + 1 + (s & 1)                                # (alternating: the load before may not have read its address yet)
        e.raw("v_add_u32 v%d, s%d, v%d" % (off, S_T, VOA + s))
        if kw != 1:
            e.raw("v_and_b32 v%d, %d, v%d" % (TMP, 1 if kw == 0 else 2, FLG + s))
            e.raw("v_cmp_ne_u32 vcc, 0, v%d" % TMP)
            G.emit_sgpr_to_valu_guard(e.raw)                   # VCC written by a VALU compare -> v_cndmask reads it
            e.raw("v_cndmask_b32 v%d, v%d, v%d, vcc" % (off, off, VOOB))
        e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], 0 offen" % (reg, reg + 3, off, RS_A, RS_A + 3), ("st", s))
    else:
        e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], s%d offen" % (reg, reg + 3, VOW + s - NPA, RS_W, RS_W + 3, S_WOFF), ("st", s))


def descriptor(e, base, lo, hi, nrec):
    e.raw("s_mov_b32 s%d, %s" % (base, lo))
    e.raw("s_and_b32 s%d, %s, 0xffff" % (base + 1, hi))
    e.raw("s_mov_b32 s%d, %s" % (base + 2, nrec))
    e.raw("s_mov_b32 s%d, 0x00020000" % (base + 3))


def request_offsets(e, tap, next_tile):
    """s[S_T] = A byte offset of `tap` of this / the next tile, s[S_WOFF] = its weight slice"""
    e.raw("s_add_u32 s%d, s%d, s%d" % (S_T, S_NEXT if next_tile else S_M0, S_TAP + tap))
    e.raw("s_mov_b32 s%d, %d" % (S_WOFF, tap * 192))


def column_flags(e, pix0):
    """flags of the 12 A pieces for the tile whose first pixel is s[pix0]: bit 0 = image column 0, bit 1 = column W - 1"""
    for s in range(NPA):
        e.raw("v_add_u32 v%d, s%d, v%d" % (TMP, pix0, PIX + s))           # pixel index inside the frame
        e.raw("v_mul_hi_u32 v%d, v%d, %%[invW]" % (TMP + 1, TMP))         # row = floor(p / W)  (magic multiply, exact for p < 2^32 / W)
        e.raw("v_mul_lo_u32 v%d, v%d, %%[imgW]" % (TMP + 1, TMP + 1))
        e.raw("v_sub_u32 v%d, v%d, v%d" % (TMP, TMP, TMP + 1))            # column
        e.raw("v_cmp_eq_u32 vcc, 0, v%d" % TMP)
        G.emit_sgpr_to_valu_guard(e.raw)
        e.raw("v_cndmask_b32_e64 v%d, 0, 1, vcc" % (FLG + s))
        e.raw("v_cmp_eq_u32 vcc, %%[wm1], v%d" % TMP)
        G.emit_sgpr_to_valu_guard(e.raw)
        e.raw("v_cndmask_b32_e64 v%d, 0, 2, vcc" % (TMP + 1))
        e.raw("v_or_b32 v%d, v%d, v%d" % (FLG + s, FLG + s, TMP + 1))


def tap_iteration(e, tap, p):
    nmf = NI * NJ
    req = tap + 2                                             # the loads of this iteration: tap + 2 of this tile / tap + 2 - 9 of the next
    request_offsets(e, req % NTAP, req >= NTAP)
    if tap == 7:
        column_flags(e, S_PNEXT)                              # this tile's own loads are all out: the flags turn to the next tile
    kw = (req % NTAP) % 3
    pars = [p, p ^ 1, p]
    staging = []
    for s in range(NPA + NPW):
        staging += [("st", s), ("ld", s)]
    state = {"k": 0}

    def stage_ops(count):
        for _ in range(count):
            if state["k"] < len(staging):
                kind, s = staging[state["k"]]
                if kind == "st":
                    store_piece(e, s)
                else:
                    load_piece(e, s, kw)
                state["k"] += 1

    for ks in range(NKS):
        par = pars[ks]
        for n in range(nmf):
            mfma(e, par, n, zero=(tap == 0 and ks == 0))
            if ks == 0:
                if n < 2:
                    frag_read(e, par, 0, *FRAG_ORDER[8 + n])          # the rest of this K step's own fragments
                elif n < 12:
                    frag_read(e, par ^ 1, 1, *FRAG_ORDER[n - 2])
            elif ks == 1:
                if n < 10:
                    frag_read(e, par ^ 1, 2, *FRAG_ORDER[n])
                elif n == 14:
                    e.barrier()                                   # A: every wave has read this tap out of LDS
                elif n >= 15:
                    stage_ops(2)
            else:
                if n < 1:
                    stage_ops(2)
                elif n < 15:
                    stage_ops(1)
                elif n == 15:
                    assert state["k"] == len(staging), state["k"]
                    e.barrier()                                   # B: tap + 1 is in LDS
                else:
                    frag_read(e, par ^ 1, 0, *FRAG_ORDER[n - 16])  # 8 of the next tap's first fragments
    return p ^ 1                                                  # the next tap starts on the other set


def epilogue(e, free_par):
    """the finished tile: + bias, bf16, 16-byte stores (dropped by the frame's descriptor past its last pixel).  Temporaries: the
    fragment set that does not hold the next tile's first fragments.  Entered with a full wait, left with its stores in flight."""
    e.drain()
    G.emit_mfma_result_guard(e.raw, "v_mfma_f32_16x16x32_bf16")      # the tile's last MFMAs -> v_accvgpr_read
    base = FA[free_par]
    BIAS, XS, OS = base, [base + 8, base + 16], [OREG, OREG + 4]
    k = 0
    for jp in range(NJ // 2):
        e.vmem("buffer_load_dwordx4 v[%d:%d], %%[voB], s[%d:%d], 0 offen offset:%d" % (BIAS, BIAS + 3, RS_B, RS_B + 3, jp * 128), ("bias", 0))
        e.vmem("buffer_load_dwordx4 v[%d:%d], %%[voB], s[%d:%d], 0 offen offset:%d" % (BIAS + 4, BIAS + 7, RS_B, RS_B + 3, jp * 128 + 16), ("bias", 1))
        for i in range(NI):
            X, O = XS[k & 1], OS[k & 1]
            e.need_vm({("out", k - 2)})                      # the store that read this register set has taken its data
            for h in range(2):
                c = acc(i, 2 * jp + h)
                for q in range(4):
                    e.raw("v_accvgpr_read_b32 v%d, a%d" % (X + 4 * h + q, c + q))
            e.need_vm({("bias", 0), ("bias", 1)})
            for pp in range(4):
                e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (X + 2 * pp, X + 2 * pp + 1, X + 2 * pp, X + 2 * pp + 1, BIAS + 2 * pp, BIAS + 2 * pp + 1))
            for pp in range(4):
                e.raw("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (O + pp, X + 2 * pp, X + 2 * pp + 1))
            e.raw("s_add_u32 s%d, s%d, %d" % (S_T2, S_M0, i * 16 * 192 + jp * 64))
            e.raw("v_add_u32 v%d, s%d, %%[voC]" % (TMP + (k & 1), S_T2))       # (vector offset: range-checked -- the ragged last tile)
            e.vmem("buffer_store_dwordx4 v[%d:%d], v%d, s[%d:%d], 0 offen" % (O, O + 3, TMP + (k & 1), RS_C, RS_C + 3), ("out", k))
            k += 1
    # the next tile becomes this one; the one after it: + pstep pixels (clamped: the tile behind the block's last one is never used)
    e.raw("s_mov_b32 s%d, s%d" % (S_P, S_PNEXT))
    e.raw("s_mov_b32 s%d, s%d" % (S_M0, S_NEXT))
    e.raw("s_add_u32 s%d, s%d, %%[pstep]" % (S_PNEXT, S_PNEXT))
    e.raw("s_min_u32 s%d, s%d, %%[plast]" % (S_PNEXT, S_PNEXT))
    e.raw("s_mul_i32 s%d, s%d, 192" % (S_NEXT, S_PNEXT))


def tile(e, p0):
    p = p0
    for tap in range(NTAP):
        p = tap_iteration(e, tap, p)
    return p                                                  # nine taps: p0 ^ 1


def body():
    e = Emit()
    descriptor(e, RS_A, "%[alo]", "%[ahi]", "%[nrec]")
    descriptor(e, RS_W, "%[wlo]", "%[whi]", "0x7fffffff")
    descriptor(e, RS_C, "%[clo]", "%[chi]", "%[nrec]")
    descriptor(e, RS_B, "%[blo]", "%[bhi]", "0x7fffffff")
    order = [VOA + s for s in range(NPA)] + [LDA + s for s in range(NPA)] + [PIX + s for s in range(NPA)] + \
            [VOW + s for s in range(NPW)] + [LDW + s for s in range(NPW)]
    assert len(order) == TAB
    for k, reg in enumerate(order):
        e.lds("ds_read_b32 v%d, %%[tab] offset:%d" % (reg, 1024 * k), ("tab", k))
    e.raw("v_mov_b32 v%d, 0x80000000" % VOOB)
    e.raw("s_mul_i32 s%d, %%[imgW], 192" % S_T)
    for kh in range(3):
        for kwi in range(3):
            t, d = 3 * kh + kwi, ((kwi - 1) * 192) & 0xffffffff
            if kh == 0:
                e.raw("s_sub_u32 s%d, 0x%x, s%d" % (S_TAP + t, d, S_T))
            elif kh == 1:
                e.raw("s_mov_b32 s%d, 0x%x" % (S_TAP + t, d))
            else:
                e.raw("s_add_u32 s%d, s%d, 0x%x" % (S_TAP + t, S_T, d))
    e.raw("s_mov_b32 s%d, %%[p0]" % S_P)
    e.raw("s_mul_i32 s%d, s%d, 192" % (S_M0, S_P))
    e.raw("s_add_u32 s%d, s%d, %%[pstep]" % (S_PNEXT, S_P))
    e.raw("s_min_u32 s%d, s%d, %%[plast]" % (S_PNEXT, S_PNEXT))
    e.raw("s_mul_i32 s%d, s%d, 192" % (S_NEXT, S_PNEXT))
    e.raw("s_mov_b32 s%d, %%[ntile]" % S_TL)
    e.drain()                                                 # the table
    e.raw("s_barrier")                                        # ... which the staging below overwrites: every wave has read it
    column_flags(e, S_P)
    # ---- tap 0 -> registers -> LDS, tap 1 -> registers, the first 8 fragments of tap 0
    request_offsets(e, 0, False)
    for s in range(NPA + NPW):
        load_piece(e, s, 0)
    for s in range(NPA + NPW):
        store_piece(e, s)
    request_offsets(e, 1, False)
    for s in range(NPA + NPW):
        load_piece(e, s, 1)
    e.barrier()
    for n in range(8):
        frag_read(e, 0, 0, *FRAG_ORDER[n])
    e.drain()
    e.raw("2:")
    p = tile(e, 0)
    assert p == 1
    epilogue(e, free_par=0)
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_TL, S_TL))
    e.raw("s_cmp_eq_u32 s%d, 0" % S_TL)
    e.raw("s_cbranch_scc1 3f")
    p = tile(e, 1)
    assert p == 0
    epilogue(e, free_par=1)
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_TL, S_TL))
    e.raw("s_cmp_lg_u32 s%d, 0" % S_TL)
    e.raw("s_cbranch_scc1 2b")
    e.raw("3:")
    e.drain()
    return e.lines


def main():
    lines = body()
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % i for i in range(96)] + ['"v%d"' % i for i in range(218)] + \
           ['"s%d"' % i for i in range(40, 78)]
    with open(OUT, "w") as f:
        f.write("// generated by tools/gen_conv_bf16_t4c.py -- do not edit\n")
        f.write("#define PIPS_T4C_TEXT \\\n")
        for ln in lines:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n\n')
        f.write("#define PIPS_T4C_CLOBBER " + ", ".join(clob) + "\n")
    print("PIPS_T4C_TEXT: %d instructions, %d MFMAs" % (len(lines), sum("v_mfma" in ln for ln in lines)))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
