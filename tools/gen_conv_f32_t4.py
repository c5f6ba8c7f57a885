#!/usr/bin/env python
"""Emit pips_amd/csrc/conv_f32_t4_asm.inc: the bodies of the kernels of conv_f32_t4.hip, each ONE assembly statement -- the 3x3 /
stride 1 / pad 1 convolutions of the fp32 encoder on channel-last maps as exact-fp32 implicit GEMMs (v_mfma_f32_32x32x2_f32,
igemm_f32_kernel's arithmetic and K order: tap-major, 32 channels per stage) on four waves, one per SIMD, in the style of
tools/gen_gemm_f32_t4.py (static schedule, counted waits) with the operand addressing of tools/gen_conv_bf16_t4c.py.

A block tile is 128 NI pixels x 32 NJ channels: wave w owns pixels 32 NI w .. of it and all 32 NJ channels (NI x NJ MFMA blocks,
16 NI NJ AccVGPRs).  With the pixels of a frame numbered row-major, tap (kh, kw) of output pixel p reads input pixel
p + (kh - 1) W + (kw - 1): the A operand of a tile, tap and 32-channel block is 128 NI pixel rows of 128 bytes at a stride of
4 Cin -- plain strided copies (piece s of a thread = row 32 s + tid / 8, 16 bytes at 16 (tid & 7)).  What the linear view gets
wrong is fixed by the frame's buffer descriptor (rows above / below the image and behind the last pixel are out of range: reads
return zero, stores are dropped) and two flags per piece (image column 0 / W - 1: out of range for the kw = 0 / kw = 2 taps).

A stage = one tap x 32 channels = 16 NI NJ MFMAs; a tile = 9 Cin / 32 stages, unrolled whole (the stage count is a multiple of
three).  THREE LDS buffers and three register sets: stage t multiplies out of buffer t % 3 while stage t + 1 goes from register
set (t + 1) % 3 to buffer (t + 1) % 3 and stage t + 4 is requested into that set (three stages of latency cover); one barrier
per stage; the pipeline runs on across the tile boundary (the next tile's first stages arrive under this tile's last ones).  A
block walks every `pstep`-th tile of ONE frame.

Accumulators are C (the pixel fragment is the MFMA's row operand): a lane holds channel l & 31 of a 32-channel block and 16
pixels (rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)) of a 32-pixel block -- per-channel sums stay in the lane.  Epilogue: + bias, 4-byte
stores (a wave store = two full 128-byte rows), and the InstanceNorm partials {sum(x - p), sum((x - p)^2), p, n} of the wave's
32 NI pixels per channel (pivot p = the wave's first pixel, handed to the upper lane half by ds_bpermute; the halves' sums are added
the same way; rows behind the frame's last pixel are masked): one float4 per (wave with a pixel inside the frame, channel).

Registers: class Cfg.  All clobbered; v[216:255] stay with the compiler.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import asm_guards as G  # noqa: E402  (wait-state guards: the numbers live in tools/asm_hazard_lint.py)

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("PIPS_GEN_OUT", os.path.join(HERE, "..", "pips_amd", "csrc", "conv_f32_t4_asm.inc"))

LDROW = 144
NV = 216
ABLATE = set(os.environ.get("PIPS_GEN_ABLATE", "").split(","))      # timing experiments only (wrong results): noflags, noepi, nostats
RS_A, RS_W, RS_C, RS_B, RS_S = 40, 44, 48, 52, 56
S_TAP = 60                          # s[60:68]: ((kh - 1) W + (kw - 1)) * 4 Cin of the nine taps, as wrapping unsigned numbers
S_P, S_PNEXT, S_M0, S_NEXT, S_TL, S_T, S_WOFF, S_T2, S_T3, S_NVR, S_C0, S_S0 = 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80


class Cfg:
    def __init__(self, name, cin, cout, ni, nj):
        self.name, self.cin, self.cout, self.ni, self.nj = name, cin, cout, ni, nj      # cout: channels of the output MAP (its row stride)
        self.spt = cin // 32                        # stages per tap
        self.ns = 9 * self.spt                      # stages per tile
        self.px = 128 * ni                          # pixels per tile
        self.npa, self.npw = 4 * ni, nj             # staged pieces per thread and stage
        self.np = self.npa + self.npw
        self.nmf = 4 * ni * nj                      # MFMAs per group of 8 K values
        self.stage_bytes = (self.px + 32 * nj) * LDROW
        fs = 4 * (ni + nj)                          # a fragment set
        self.fa = [0, fs]
        self.fw = [4 * ni, fs + 4 * ni]
        self.st = 2 * fs                            # the three stages in flight
        r = self.st + 12 * self.np
        self.voa, r = r, r + self.npa               # per-piece byte offsets inside a tile's A run / inside W
        self.flg, r = r, r + self.npa               # column flags of the A pieces
        self.vow, r = r, r + self.npw
        self.bias, r = r, r + nj
        self.tmp, r = r, r + 4
        self.voob, r = r, r + 1
        r += r & 1
        self.stat, r = r, r + 4 * nj                # epilogue: {s1, s2, p, n} per channel block
        self.x, r = r, r + 2 * nj                   # two sets of output values
        self.d, r = r, r + 2
        self.off, r = r, r + 2
        self.cnt, r = r, r + 1
        assert r <= NV, r
        self.blocks = [(i, j) for j in range(nj) for i in range(ni)]
        self.frag_order = [("w", 0)] + [("a", i) for i in range(ni)] + [("w", j) for j in range(1, nj)]

    def acc(self, i, j):
        return 16 * (i + self.ni * j)


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

    def need_loads(self):
        self.need_vm({t for t in self.vm if t[0] != "out"})

    def barrier(self):
        if self.lgkm:
            self.lines.append("s_waitcnt lgkmcnt(0)")
            self.lgkm = []
        self.lines.append("s_barrier")

    def drain(self):
        self.lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.lgkm, self.vm = [], []


def frag_read(e, c, buf, fset, kk, which, idx):
    """fragment `idx` (A: pixel block i, W: channel block j) of the 8 K values kk of the stage in LDS buffer `buf`, into set `fset`"""
    reg = (c.fa if which == "a" else c.fw)[fset] + 4 * idx
    e.lds("ds_read_b128 v[%d:%d], %%[r%s%d] offset:%d" % (reg, reg + 3, "A" if which == "a" else "W", buf, idx * 32 * LDROW + kk * 32),
          ("f" + which, fset, idx))


def mfma(e, c, fset, k, i, j, zero):
    e.need_lds({("fw", fset, j), ("fa", fset, i)})
    a = c.acc(i, j)
    e.raw("v_mfma_f32_32x32x2_f32 a[%d:%d], v%d, v%d, %s" %
          (a, a + 15, c.fa[fset] + 4 * i + k, c.fw[fset] + 4 * j + k, "0" if zero else "a[%d:%d]" % (a, a + 15)))


def store_piece(e, c, buf, ring, s):
    e.need_vm({("st", ring, s)})
    reg = c.st + 4 * (ring * c.np + s)
    row = 32 * s if s < c.npa else c.px + 32 * (s - c.npa)
    e.lds("ds_write_b128 %%[wb%d], v[%d:%d] offset:%d" % (buf, reg, reg + 3, row * LDROW), ("wr", s))


def load_piece(e, c, ring, s, kw):
    """piece s of the stage whose tile + tap + channel-block byte offset sits in s[S_T] (A) / whose K offset in s[S_WOFF] (W); kw = the
    tap's column (0 / 2: the flagged pieces go out of range).  The A offset is formed in a VECTOR register: a raw buffer's range
    check covers the vector offset only, and the zero padding above / below the image and behind the last pixel rides on it."""
    reg = c.st + 4 * (ring * c.np + s)
    if s < c.npa:
        off = c.tmp + 1 + (s & 1)                              # (alternating: the load before may not have read its address yet)
        e.raw("v_add_u32 v%d, s%d, v%d" % (off, S_T, c.voa + s))
        if kw != 1 and "noflags" not in ABLATE:
            e.raw("v_and_b32 v%d, %d, v%d" % (c.tmp, 1 if kw == 0 else 2, c.flg + s))
            e.raw("v_cmp_ne_u32 vcc, 0, v%d" % c.tmp)
            G.emit_sgpr_to_valu_guard(e.raw)                   # VCC written by a VALU compare -> v_cndmask reads it
            e.raw("v_cndmask_b32 v%d, v%d, v%d, vcc" % (off, off, c.voob))
        e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], 0 offen" % (reg, reg + 3, off, RS_A, RS_A + 3), ("st", ring, s))
    else:
        e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], s%d offen" % (reg, reg + 3, c.vow + s - c.npa, RS_W, RS_W + 3, S_WOFF),
               ("st", ring, s))


def descriptor(e, base, lo, hi, nrec):
    e.raw("s_mov_b32 s%d, %s" % (base, lo))
    e.raw("s_and_b32 s%d, %s, 0xffff" % (base + 1, hi))
    e.raw("s_mov_b32 s%d, %s" % (base + 2, nrec))
    e.raw("s_mov_b32 s%d, 0x00020000" % (base + 3))


def request_offsets(e, c, t):
    """s[S_T] = A byte offset of stage t (of this tile; t >= ns: of the next), s[S_WOFF] = its K offset in W; returns the tap's column"""
    nxt, t = t >= c.ns, t % c.ns
    tap, cb = t // c.spt, t % c.spt
    e.raw("s_add_u32 s%d, s%d, s%d" % (S_T, S_NEXT if nxt else S_M0, S_TAP + tap))
    if cb:
        e.raw("s_add_u32 s%d, s%d, %d" % (S_T, S_T, 128 * cb))
    e.raw("s_mov_b32 s%d, %d" % (S_WOFF, 128 * t))
    return tap % 3


def column_flags(e, c, pix0):
    """flags of the A pieces for the tile whose first pixel is s[pix0]: bit 0 = image column 0, bit 1 = column W - 1"""
    T = c.tmp
    for s in range(c.npa):
        e.raw("s_add_u32 s%d, s%d, %d" % (S_T2, pix0, 32 * s))
        e.raw("v_add_u32 v%d, s%d, %%[vlr]" % (T, S_T2))                  # pixel index inside the frame
        e.raw("v_mul_hi_u32 v%d, v%d, %%[invW]" % (T + 1, T))             # row = floor(p / W)  (magic multiply, exact for p < 2^32 / W)
        e.raw("v_mul_lo_u32 v%d, v%d, %%[imgW]" % (T + 1, T + 1))
        e.raw("v_sub_u32 v%d, v%d, v%d" % (T, T, T + 1))                  # column
        e.raw("v_cmp_eq_u32 vcc, 0, v%d" % T)
        G.emit_sgpr_to_valu_guard(e.raw)
        e.raw("v_cndmask_b32_e64 v%d, 0, 1, vcc" % (c.flg + s))
        e.raw("v_cmp_eq_u32 vcc, %%[wm1], v%d" % T)
        G.emit_sgpr_to_valu_guard(e.raw)
        e.raw("v_cndmask_b32_e64 v%d, 0, 2, vcc" % (T + 1))
        e.raw("v_or_b32 v%d, v%d, v%d" % (c.flg + s, c.flg + s, T + 1))


def stage(e, c, t):
    """Stage t of the tile: 4 nmf MFMAs on LDS buffer t % 3; stage t + 1 goes from register set (t + 1) % 3 to buffer (t + 1) % 3
    and stage t + 4 is requested into that set."""
    buf, ring, nmf = t % 3, (t + 1) % 3, c.nmf
    if t + 4 == c.ns:
        column_flags(e, c, S_PNEXT)                           # this tile's own loads are all out: the flags turn to the next tile
    kw = request_offsets(e, c, t + 4)
    nfr = len(c.frag_order)
    slots = {}

    def put(n, op):
        slots.setdefault(n, []).append(op)

    for kk in range(3):                                       # fragments of the 8 K values kk + 1, set (kk + 1) & 1
        for r, (which, idx) in enumerate(c.frag_order):
            put(nmf * kk + 1 + r, ("fr", buf, (kk + 1) & 1, kk + 1, which, idx))
    bar = 3 * nmf - 1
    ops = []
    for s in range(c.np):
        ops += [("st", s), ("ld", s)]
    free = [n for n in range(nfr + 1, bar - 1) if n not in slots]
    assert len(free) >= len(ops) // 2, (len(free), len(ops))
    per = -(-len(ops) // len(free))                           # operations per slot
    for k, op in enumerate(ops):
        put(free[k // per], op)
    put(bar, ("bar",))
    for r, (which, idx) in enumerate(c.frag_order):           # the next stage's first fragments, set 0, from the next buffer
        put(bar + 2 + r, ("fr", (buf + 1) % 3, 0, 0, which, idx))
    n = 0
    for kk in range(4):
        for k in range(4):
            for (i, j) in c.blocks:
                mfma(e, c, kk & 1, k, i, j, t == 0 and kk == 0 and k == 0)
                for op in slots.get(n, []):
                    if op[0] == "fr":
                        frag_read(e, c, *op[1:])
                    elif op[0] == "st":
                        store_piece(e, c, (buf + 1) % 3, ring, op[1])
                    elif op[0] == "ld":
                        load_piece(e, c, ring, op[1], kw)
                    else:
                        e.barrier()
                n += 1


def epilogue(e, c):
    """the finished tile: + bias, 4-byte stores (dropped behind the frame's last pixel), InstanceNorm partials; left with its last
    stores in flight; then the tile state moves on"""
    if "noepi" in ABLATE:
        tile_advance(e, c)
        return
    e.need_loads()
    G.emit_mfma_result_guard(e.raw, "v_mfma_f32_32x32x2_f32")            # the tile's last MFMAs -> v_accvgpr_read
    T, CNT, D = c.tmp, c.cnt, c.d
    e.raw("s_sub_u32 s%d, %%[npix], s%d" % (S_NVR, S_P))                  # pixels of the frame from this tile's first on (>= 1)
    e.raw("s_mul_i32 s%d, s%d, %d" % (S_C0, S_P, 4 * c.cout))             # the tile's first output row, bytes
    # rows of this lane inside the frame: sum over (i, g) of clamp(nv - (row0 + 32 i + 8 g), 0, 4),  row0 = 32 NI wave + 4 half
    e.raw("v_mov_b32 v%d, 0" % CNT)
    for i in range(c.ni):
        for g in range(4):
            e.raw("s_sub_i32 s%d, s%d, %d" % (S_T2, S_NVR, 32 * i + 8 * g))
            e.raw("v_sub_u32 v%d, s%d, %%[vrow]" % (T, S_T2))
            e.raw("v_max_i32 v%d, 0, v%d" % (T, T))
            e.raw("v_min_i32 v%d, 4, v%d" % (T, T))
            e.raw("v_add_u32 v%d, v%d, v%d" % (CNT, CNT, T))
    e.raw("v_cvt_f32_u32 v%d, v%d" % (CNT, CNT))
    # the pivots: the wave's first pixel (row 0: lane half 0, register 0 of pixel block 0) per channel, + bias, to both lane halves
    for j in range(c.nj):
        S = c.stat + 4 * j
        e.raw("v_accvgpr_read_b32 v%d, a%d" % (T, c.acc(0, j)))
        e.raw("v_add_f32 v%d, v%d, v%d" % (T, T, c.bias + j))
        e.lds("ds_bpermute_b32 v%d, %%[vl31x4], v%d" % (S + 2, T), ("piv", j))
        e.raw("v_mov_b32 v%d, 0" % S)
        e.raw("v_mov_b32 v%d, 0" % (S + 1))
    k = 0
    for i in range(c.ni):
        for r in range(16):
            rho = 32 * i + (r & 3) + 8 * (r >> 2)
            X = c.x + c.nj * (k & 1)
            OFF = c.off + (k & 1)
            e.need_vm({("out", k - 2)})                      # the stores that read this register set have taken their data
            e.raw("s_sub_i32 s%d, s%d, %d" % (S_T2, S_NVR, rho))
            e.raw("s_add_u32 s%d, s%d, %d" % (S_T3, S_C0, rho * 4 * c.cout))
            e.raw("v_cmp_gt_i32 vcc, s%d, %%[vrow]" % S_T2)                # this row lies inside the frame
            e.raw("v_add_u32 v%d, s%d, %%[voC]" % (OFF, S_T3))             # (vector offset: range-checked -- the ragged last tile)
            for j in range(c.nj):
                S = c.stat + 4 * j
                e.raw("v_accvgpr_read_b32 v%d, a%d" % (X + j, c.acc(i, j) + r))
                e.raw("v_add_f32 v%d, v%d, v%d" % (X + j, X + j, c.bias + j))
                e.need_lds({("piv", j)})
                if "nostats" not in ABLATE:
                    e.raw("v_sub_f32 v%d, v%d, v%d" % (D, X + j, S + 2))
                    e.raw("v_cndmask_b32 v%d, 0, v%d, vcc" % (D, D))
                    e.raw("v_add_f32 v%d, v%d, v%d" % (S, S, D))
                    e.raw("v_fmac_f32 v%d, v%d, v%d" % (S + 1, D, D))
                e.vmem("buffer_store_dword v%d, v%d, s[%d:%d], 0 offen offset:%d" % (X + j, OFF, RS_C, RS_C + 3, 128 * j), ("out", k))
            k += 1
    # the lane halves' sums and counts meet (ds_bpermute with lane ^ 32), the lower half stores {s1, s2, p, n}
    for j in range(c.nj):
        S = c.stat + 4 * j
        e.lds("ds_bpermute_b32 v%d, %%[vswap], v%d" % (D, S), ("sw", j, 0))
        e.lds("ds_bpermute_b32 v%d, %%[vswap], v%d" % (D + 1, S + 1), ("sw", j, 1))
        if j == 0:
            e.lds("ds_bpermute_b32 v%d, %%[vswap], v%d" % (T, CNT), ("sw", "n"))
            e.need_lds({("sw", "n")})
            e.raw("v_add_f32 v%d, v%d, v%d" % (CNT, CNT, T))
        e.need_lds({("sw", j, 0), ("sw", j, 1)})
        e.raw("v_add_f32 v%d, v%d, v%d" % (S, S, D))
        e.raw("v_add_f32 v%d, v%d, v%d" % (S + 1, S + 1, D + 1))
        e.raw("v_mov_b32 v%d, v%d" % (S + 3, CNT))
    e.raw("s_mul_i32 s%d, s%d, %%[sbytes]" % (S_S0, S_P))                 # (sbytes = bytes of partials per pixel of tile start: 64 Cout / px)
    e.raw("v_add_u32 v%d, s%d, %%[voS]" % (T, S_S0))                      # (range-checked: waves without a pixel in the frame, upper lane half)
    for j in range(c.nj):
        S = c.stat + 4 * j
        e.vmem("buffer_store_dwordx4 v[%d:%d], v%d, s[%d:%d], 0 offen offset:%d" % (S, S + 3, T, RS_S, RS_S + 3, 512 * j), ("out", "s"))
    tile_advance(e, c)


def tile_advance(e, c):
    """the next tile becomes this one; the one after it: + pstep pixels (clamped: the tile behind the block's last one is never used)"""
    e.raw("s_mov_b32 s%d, s%d" % (S_P, S_PNEXT))
    e.raw("s_mov_b32 s%d, s%d" % (S_M0, S_NEXT))
    e.raw("s_add_u32 s%d, s%d, %%[pstep]" % (S_PNEXT, S_PNEXT))
    e.raw("s_min_u32 s%d, s%d, %%[plast]" % (S_PNEXT, S_PNEXT))
    e.raw("s_mul_i32 s%d, s%d, %d" % (S_NEXT, S_PNEXT, 4 * c.cin))


def body(c):
    e = Emit()
    descriptor(e, RS_A, "%[alo]", "%[ahi]", "%[nrecA]")
    descriptor(e, RS_W, "%[wlo]", "%[whi]", "0x7fffffff")
    descriptor(e, RS_C, "%[clo]", "%[chi]", "%[nrecC]")
    descriptor(e, RS_B, "%[blo]", "%[bhi]", "0x7fffffff")
    descriptor(e, RS_S, "%[slo]", "%[shi]", "%[nrecS]")
    e.raw("v_mov_b32 v%d, %%[voA]" % c.voa)
    for s in range(1, c.npa):
        e.raw("v_add_u32 v%d, %d, v%d" % (c.voa + s, 32 * 4 * c.cin, c.voa + s - 1))
    e.raw("v_mov_b32 v%d, %%[voW]" % c.vow)
    for s in range(1, c.npw):
        e.raw("v_add_u32 v%d, %d, v%d" % (c.vow + s, 32 * 9 * 4 * c.cin, c.vow + s - 1))
    e.raw("v_mov_b32 v%d, 0x80000000" % c.voob)
    e.raw("s_mul_i32 s%d, %%[imgW], %d" % (S_T, 4 * c.cin))
    for kh in range(3):
        for kwi in range(3):
            t, d = 3 * kh + kwi, ((kwi - 1) * 4 * c.cin) & 0xffffffff
            if kh == 0:
                e.raw("s_sub_u32 s%d, 0x%x, s%d" % (S_TAP + t, d, S_T))
            elif kh == 1:
                e.raw("s_mov_b32 s%d, 0x%x" % (S_TAP + t, d))
            else:
                e.raw("s_add_u32 s%d, s%d, 0x%x" % (S_TAP + t, S_T, d))
    e.raw("s_mov_b32 s%d, %%[p0]" % S_P)
    e.raw("s_mul_i32 s%d, s%d, %d" % (S_M0, S_P, 4 * c.cin))
    e.raw("s_add_u32 s%d, s%d, %%[pstep]" % (S_PNEXT, S_P))
    e.raw("s_min_u32 s%d, s%d, %%[plast]" % (S_PNEXT, S_PNEXT))
    e.raw("s_mul_i32 s%d, s%d, %d" % (S_NEXT, S_PNEXT, 4 * c.cin))
    e.raw("s_mov_b32 s%d, %%[ntile]" % S_TL)
    column_flags(e, c, S_P)
    # ---- stages 0, 1, 2 -> register sets, bias, stage 0 -> LDS buffer 0, stage 3 requested, fragments of the first 8 K values
    for t in range(3):
        kw = request_offsets(e, c, t)
        for s in range(c.np):
            load_piece(e, c, t, s, kw)
    for j in range(c.nj):
        e.vmem("buffer_load_dword v%d, %%[voB], s[%d:%d], 0 offen offset:%d" % (c.bias + j, RS_B, RS_B + 3, 128 * j), ("bias", j))
    for s in range(c.np):
        store_piece(e, c, 0, 0, s)
    kw = request_offsets(e, c, 3)
    for s in range(c.np):
        load_piece(e, c, 0, s, kw)
    e.barrier()
    for which, idx in c.frag_order:
        frag_read(e, c, 0, 0, 0, which, idx)
    e.raw("2:")
    for t in range(c.ns):
        stage(e, c, t)
    epilogue(e, c)
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_TL, S_TL))
    e.raw("s_cmp_lg_u32 s%d, 0" % S_TL)
    e.raw("s_cbranch_scc1 2b")
    e.drain()
    return e.lines


CONFIGS = [Cfg("C64", 64, 64, 2, 2), Cfg("C96", 96, 96, 1, 3), Cfg("C416", 416, 256, 1, 2)]


def main():
    with open(OUT, "w") as f:
        f.write("// generated by tools/gen_conv_f32_t4.py -- do not edit\n")
        nacc = 0
        for c in CONFIGS:
            lines = body(c)
            f.write("#define PIPS_CF32T4_%s_TEXT \\\n" % c.name)
            for ln in lines:
                f.write('    "%s\\n\\t" \\\n' % ln)
            f.write('    ""\n\n')
            nacc = max(nacc, 16 * c.ni * c.nj)
            print("PIPS_CF32T4_%s_TEXT: %d instructions, %d MFMAs, LDS %d bytes" %
                  (c.name, len(lines), sum("v_mfma" in ln for ln in lines), 3 * c.stage_bytes))
        clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % i for i in range(nacc)] + ['"v%d"' % i for i in range(NV)] + \
               ['"s%d"' % i for i in range(40, 82)]
        f.write("#define PIPS_CF32T4_CLOBBER " + ", ".join(clob) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
