#!/usr/bin/env python
"""Emit pips_amd/csrc/gemm_x3_t4_asm.inc: the bodies of the kernels of gemm_x3_t4.hip, each ONE assembly statement -- the mixer's
Linears in the split-bf16 matrix mode (gemm_x3.hip: every fp32 operand = three exact bf16 terms h + m + l, six bf16 products per
fp32 product, fp32 accumulation) on four waves, one per SIMD, with the static schedule of tools/gen_gemm_f32_t4.py.

128 x 128 tile, waves 2 x 2, every wave a 64 x 64 block = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16 (64 AccVGPR accumulators, C^T as
in the fp32 kernels: same epilogues).  A K step = 16 K values = 6 products x 4 blocks = 24 MFMAs (768 clocks), in the order of
gemm_x3_kernel, smallest first: Al Wh, Ah Wl, Am Wm, Am Wh, Ah Wm, Ah Wh.  A stage = 32 K values = two K steps; LDS: two buffers of
six planes (A h / m / l, W h / m / l: 128 rows x 64 bytes at a stride of 80 -- fragment reads and staging writes conflict-free).

Fragments live in ONE register set that is refilled behind its last use inside the K step (Al behind product 0, Wl behind 1, Am behind
3, Wm behind 4), only the h planes -- used by the last product -- are double buffered: 64 fragment registers instead of 96.

Staging: A stays fp32 in memory and is split while staged (9 vector instructions per pair of values), W comes as three bf16
planes.  With g = (stage t, K step ks):
    (t, 0)   W and the second A piece of stage t + 1: registers (-> split) -> LDS buffer (t + 1) & 1, the same of stage t + 3
             requested;  fragments of (t, 1) read
    (t, 1)   barrier (buffer (t + 1) & 1 is complete, nobody reads buffer t & 1 any more); the first A piece of stage t + 2:
             registers -> split -> LDS buffer t & 1, that piece of stage t + 4 requested;  fragments of (t + 1, 0) read from
             buffer (t + 1) & 1
so a stage's operands are requested two stages (3 072 clocks) ahead and are written to LDS one stage ahead.

Registers (all clobbered; v[216:255] stay with the compiler):
    a[0:63] accumulators (block (i, j) -> a[16 (i + 2 j) : +15]);  a[64:127] the tile's residual quads;  a[128:159] bias quads
    v[0:15] Ah fragments (two sets of two), v[16:31] Wh (two sets), v[32:39] Am, v[40:47] Wm, v[48:55] Al, v[56:63] Wl
    v[64:95] A in flight (2 stages x 2 pieces x 8 fp32), v[96:143] W in flight (2 stages x 6 pieces of 16 B)
    v[144:157] the split's outputs and scratch, v[158:159] the GELU polynomial's second coefficient
    v[160:167] per-piece global byte offsets, v[168:215] the epilogue's temporaries
    s[40:59] buffer descriptors A, W, C, bias, R;  s[60:61] / s[82:83] row-block offsets in C / R;  s[62:81] GELU constants;
    s[84:97] loop state
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("PIPS_GEN_OUT", os.path.join(HERE, "gemm_x3_t4_asm.inc"))

LDROW = 80
PLANE = 128 * LDROW                 # 10 240
WPART = 3 * PLANE                   # the W planes of a stage sit behind the A planes
AH, WH, AM, WM, AL, WL = 0, 16, 32, 40, 48, 56
ARAW, WST, SPL, VC, VO, EP = 64, 96, 144, 158, 160, 168
ARES, ABIAS = 64, 128
NV = 216
RS_A, RS_W, RS_C, RS_B, RS_R = 40, 44, 48, 52, 56
S_CR, S_RR = 60, 82
S_GC = 62
S_KL, S_SOA, S_SOW, S_TL, S_RQA, S_RQW, S_LASTA, S_T, S_MASK = 84, 85, 86, 87, 88, 89, 90, 91, 92
COEF = [3.208326405e-07, -6.917509381e-06, 6.041429151e-05, -2.428356966e-04, -5.105399032e-05, 6.989960559e-03,
        -5.246259645e-02, -4.592153430e-01, -1.151104689e+00]
TMAX = 5.65685425
PRODUCTS = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]      # (A plane, W plane): l h, h l, m m, m h, h m, h h


def f32(x):
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


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


def acc(i, j):
    return 16 * (i + 2 * j)


def frag_reg(which, plane, hset, idx):
    """registers of fragment idx (A: row block i, W: column block j) of a plane (0 h, 1 m, 2 l); the h planes have two sets"""
    base = {("a", 0): AH + 8 * hset, ("w", 0): WH + 8 * hset, ("a", 1): AM, ("w", 1): WM, ("a", 2): AL, ("w", 2): WL}[(which, plane)]
    return base + 4 * idx


def frag_read(e, buf, ks, which, plane, hset, idx):
    reg = frag_reg(which, plane, hset, idx)
    e.lds("ds_read_b128 v[%d:%d], %%[r%s%d] offset:%d" % (reg, reg + 3, "A" if which == "a" else "W", buf, plane * PLANE + idx * 32 * LDROW + ks * 32),
          ("f", which, plane, hset if plane == 0 else 0, idx))


def mfma(e, pa, pw, hset, i, j, zero):
    ra, rw = frag_reg("a", pa, hset, i), frag_reg("w", pw, hset, j)
    e.need_lds({("f", "a", pa, hset if pa == 0 else 0, i), ("f", "w", pw, hset if pw == 0 else 0, j)})
    a = acc(i, j)
    e.raw("v_mfma_f32_32x32x16_bf16 a[%d:%d], v[%d:%d], v[%d:%d], %s" %
          (a, a + 15, rw, rw + 3, ra, ra + 3, "0" if zero else "a[%d:%d]" % (a, a + 15)))


def descriptor(e, base, lo, hi):
    e.raw("s_mov_b32 s%d, %s" % (base, lo))
    e.raw("s_and_b32 s%d, %s, 0xffff" % (base + 1, hi))
    e.raw("s_mov_b32 s%d, 0x7fffffff" % (base + 2))
    e.raw("s_mov_b32 s%d, 0x00020000" % (base + 3))


def advance_a(e):
    """A request: one stage further; behind a tile's last stage the next tile's first; clamped behind the block's last tile"""
    e.raw("s_add_u32 s%d, s%d, 1" % (S_RQA, S_RQA))
    e.raw("s_add_u32 s%d, s%d, 128" % (S_SOA, S_SOA))
    e.raw("s_add_u32 s%d, s%d, %%[tstepA]" % (S_T, S_SOA))
    e.raw("s_cmp_eq_u32 s%d, %%[kt]" % S_RQA)
    e.raw("s_cselect_b32 s%d, 0, s%d" % (S_RQA, S_RQA))
    e.raw("s_cselect_b32 s%d, s%d, s%d" % (S_SOA, S_T, S_SOA))
    e.raw("s_min_u32 s%d, s%d, s%d" % (S_SOA, S_SOA, S_LASTA))


def advance_w(e):
    e.raw("s_add_u32 s%d, s%d, 1" % (S_RQW, S_RQW))
    e.raw("s_add_u32 s%d, s%d, 64" % (S_SOW, S_SOW))
    e.raw("s_cmp_eq_u32 s%d, %%[kt]" % S_RQW)
    e.raw("s_cselect_b32 s%d, 0, s%d" % (S_RQW, S_RQW))
    e.raw("s_cselect_b32 s%d, 0, s%d" % (S_SOW, S_SOW))


def load_a(e, ring, s):
    """piece s (rows 64 s + tid / 4, 8 K values) of the stage at s[S_SOA] -> A set `ring`"""
    reg = ARAW + 8 * (2 * ring + s)
    e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], s%d offen" % (reg, reg + 3, VO + s, RS_A, RS_A + 3, S_SOA), ("a", ring, s, 0))
    e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], s%d offen offset:16" % (reg + 4, reg + 7, VO + s, RS_A, RS_A + 3, S_SOA), ("a", ring, s, 1))


def load_w(e, ring, q):
    """piece q = 2 plane + s of W (rows 64 s + tid / 4 of the plane, 8 K values) of the stage at s[S_SOW] -> W set `ring`"""
    reg = WST + 4 * (6 * ring + q)
    e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], s%d offen" % (reg, reg + 3, VO + 2 + q, RS_W, RS_W + 3, S_SOW), ("w", ring, q))


def write_w(e, buf, ring, q):
    e.need_vm({("w", ring, q)})
    reg = WST + 4 * (6 * ring + q)
    e.lds("ds_write_b128 %%[wb%d], v[%d:%d] offset:%d" % (buf, reg, reg + 3, WPART + (q // 2) * PLANE + (q % 2) * 64 * LDROW), ("wr", "w", q))


def split_ops(e, ring, s, buf):
    """A piece s of set `ring` -> its three bf16 planes -> LDS buffer `buf`; returns the micro-operations (callables) in order"""
    R = ARAW + 8 * (2 * ring + s)
    H, M, L, T = SPL, SPL + 4, SPL + 8, SPL + 12
    ops = []

    def wait(e):
        e.need_vm({("a", ring, s, 0), ("a", ring, s, 1)})
    ops.append(wait)
    for q in range(4):
        def a(e, q=q):
            x = R + 2 * q
            e.raw("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (H + q, x, x + 1))
            e.raw("v_lshlrev_b32 v%d, 16, v%d" % (T, H + q))
            e.raw("v_and_b32 v%d, s%d, v%d" % (T + 1, S_MASK, H + q))
            e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d] neg_lo:[0,1] neg_hi:[0,1]" % (x, x + 1, x, x + 1, T, T + 1))

        def b(e, q=q):
            x = R + 2 * q
            e.raw("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (M + q, x, x + 1))
            e.raw("v_lshlrev_b32 v%d, 16, v%d" % (T, M + q))
            e.raw("v_and_b32 v%d, s%d, v%d" % (T + 1, S_MASK, M + q))
            e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d] neg_lo:[0,1] neg_hi:[0,1]" % (x, x + 1, x, x + 1, T, T + 1))
            e.raw("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (L + q, x, x + 1))
        ops += [a, b]
    for p, reg in enumerate((H, M, L)):
        def w(e, p=p, reg=reg):
            e.lds("ds_write_b128 %%[wb%d], v[%d:%d] offset:%d" % (buf, reg, reg + 3, p * PLANE + s * 64 * LDROW), ("wr", "a", s, p))
        ops.append(w)
    return ops


def kstep(e, t, ks, first, extra=()):
    """K step ks of stage t (t: parity inside an unrolled pair of stages).  24 MFMAs on the fragments in registers; the refills read
    the next K step's fragments (ks = 0: the other half of buffer t & 1; ks = 1: buffer (t + 1) & 1, behind the barrier)."""
    buf, hset = t & 1, ks
    nbuf, nks, nh = (buf, 1, 1) if ks == 0 else (1 - buf, 0, 0)
    slots = {}

    def put(n, fn):
        slots.setdefault(n, []).append(fn)

    def rd(which, plane, idx):
        return lambda e: frag_read(e, nbuf, nks, which, plane, nh, idx)

    for idx in range(2):
        put(4 + idx, rd("a", 2, idx))                          # Al: behind product 0
        put(8 + idx, rd("w", 2, idx))                          # Wl: behind product 1
        put(10 + idx, rd("w", 0, idx))                         # the h planes of the next K step -> the other set
        put(12 + idx, rd("a", 0, idx))
        put(16 + idx, rd("a", 1, idx))                         # Am: behind product 3
        put(20 + idx, rd("w", 1, idx))                         # Wm: behind product 4
    micro = []
    if ks == 0:
        # W of stage t + 1 and the SECOND A piece of stage t + 1 -> the other buffer; W and that A piece of stage t + 3 requested
        advance_w(e)
        ring = (t + 1) & 1
        amicro = split_ops(e, ring, 1, 1 - buf) + [lambda e: load_a(e, ring, 1)]
        wmicro = []
        for q in range(6):
            wmicro.append(lambda e, q=q: write_w(e, 1 - buf, ring, q))
            wmicro.append(lambda e, q=q: load_w(e, ring, q))
        while amicro or wmicro:                                # interleaved: the split's vector work between the W copies
            if amicro:
                micro.append(amicro.pop(0))
            if wmicro:
                micro.append(wmicro.pop(0))
        first_slot = 0
    else:
        # barrier; the FIRST A piece of stage t + 2 -> this stage's own buffer (nobody reads it any more); that piece of stage t + 4 requested
        advance_a(e)
        ring = t & 1
        put(1, lambda e: e.barrier())
        micro = split_ops(e, ring, 0, buf) + [lambda e: load_a(e, ring, 0)]
        first_slot = 2
    nslots = 24 - first_slot
    for k, fn in enumerate(micro):
        put(first_slot + (k * nslots) // len(micro), fn)
    rest = [n for n in range(24) if len(slots.get(n, [])) == 0] + [n for n in range(24) if len(slots.get(n, [])) == 1]
    for k, fn in enumerate(extra):
        put(rest[k % len(rest)], fn)
    n = 0
    for p, (pa, pw) in enumerate(PRODUCTS):
        for j in range(2):
            for i in range(2):
                mfma(e, pa, pw, hset, i, j, first and p == 0)
                for fn in slots.get(n, []):
                    fn(e)
                n += 1


def gelu4(e, X, T, Q):
    """exact GELU of the 8 values v[X:X+7] in place: gelu_exact2's arithmetic (common.h), four pairs side by side"""
    for p in range(4):
        for h in range(2):
            e.raw("v_min_f32_e64 v%d, |v%d|, s%d" % (T + 2 * p + h, X + 2 * p + h, S_GC + 18))
    for p in range(4):
        e.raw("v_pk_fma_f32 v[%d:%d], v[%d:%d], s[%d:%d], v[%d:%d] op_sel_hi:[1,0,1]" %
              (Q + 2 * p, Q + 2 * p + 1, T + 2 * p, T + 2 * p + 1, S_GC, S_GC + 1, VC, VC + 1))
    for c in range(2, len(COEF)):
        for p in range(4):
            e.raw("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], s[%d:%d] op_sel_hi:[1,1,0]" %
                  (Q + 2 * p, Q + 2 * p + 1, Q + 2 * p, Q + 2 * p + 1, T + 2 * p, T + 2 * p + 1, S_GC + 2 * c, S_GC + 2 * c + 1))
    for p in range(4):
        e.raw("v_pk_mul_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (Q + 2 * p, Q + 2 * p + 1, Q + 2 * p, Q + 2 * p + 1, T + 2 * p, T + 2 * p + 1))
    for p in range(4):
        for h in range(2):
            e.raw("v_exp_f32_e32 v%d, v%d" % (Q + 2 * p + h, Q + 2 * p + h))
    for p in range(4):
        for h in range(2):
            e.raw("v_max_f32_e32 v%d, 0, v%d" % (X + 2 * p + h, X + 2 * p + h))
    for p in range(4):
        e.raw("v_pk_mul_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (T + 2 * p, T + 2 * p + 1, T + 2 * p, T + 2 * p + 1, Q + 2 * p, Q + 2 * p + 1))
    for p in range(4):
        e.raw("v_pk_fma_f32 v[%d:%d], v[%d:%d], -0.5, v[%d:%d] op_sel_hi:[1,0,1]" %
              (X + 2 * p, X + 2 * p + 1, T + 2 * p, T + 2 * p + 1, X + 2 * p, X + 2 * p + 1))


def residual_loads():
    """the tile's residual quads -> a[64:127] (issued in the slots of the tile's last K step)"""
    ops = []
    for i in range(2):
        for j in range(2):
            for q in range(4):
                def op(e, i=i, j=j, q=q):
                    r = ARES + 4 * (q + 4 * j + 8 * i)
                    e.vmem("buffer_load_dwordx4 a[%d:%d], %%[voR], s[%d:%d], s%d offen offset:%d" %
                           (r, r + 3, RS_R, RS_R + 3, S_RR + i, (32 * j + 8 * q) * 4), ("res", i, j, q))
                ops.append(op)
    return ops


def epilogue(e, epi):
    """the wave's 64 x 64 block: + bias, GELU or + residual, 16-byte stores; left with its last stores in flight"""
    e.need_loads()
    e.raw("s_nop 15")
    e.raw("s_nop 15")
    sets = [(EP, EP + 8, EP + 16), (EP + 24, EP + 32, EP + 40)]
    k = 0
    for i in range(2):
        for j in range(2):
            for qq in range(2):
                X, T, Q = sets[k & 1]
                e.need_vm({("out", k - 2)})
                for h in range(2):
                    q = 2 * qq + h
                    for r in range(4):
                        e.raw("v_accvgpr_read_b32 v%d, a%d" % (X + 4 * h + r, acc(i, j) + 4 * q + r))
                        e.raw("v_accvgpr_read_b32 v%d, a%d" % (T + 4 * h + r, ABIAS + 4 * (4 * j + q) + r))
                for p in range(4):
                    e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (X + 2 * p, X + 2 * p + 1, X + 2 * p, X + 2 * p + 1, T + 2 * p, T + 2 * p + 1))
                if epi == "gelu":
                    gelu4(e, X, T, Q)
                else:
                    for h in range(2):
                        q = 2 * qq + h
                        for r in range(4):
                            e.raw("v_accvgpr_read_b32 v%d, a%d" % (T + 4 * h + r, ARES + 4 * (q + 4 * j + 8 * i) + r))
                    for p in range(4):
                        e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (X + 2 * p, X + 2 * p + 1, X + 2 * p, X + 2 * p + 1, T + 2 * p, T + 2 * p + 1))
                for h in range(2):
                    e.vmem("buffer_store_dwordx4 v[%d:%d], %%[voC], s[%d:%d], s%d offen offset:%d" %
                           (X + 4 * h, X + 4 * h + 3, RS_C, RS_C + 3, S_CR + i, (32 * j + 8 * (2 * qq + h)) * 4), ("out", k))
                k += 1
    e.raw("s_add_u32 s%d, s%d, %%[tstepC]" % (RS_C, RS_C))
    e.raw("s_addc_u32 s%d, s%d, 0" % (RS_C + 1, RS_C + 1))
    if epi == "res":
        e.raw("s_add_u32 s%d, s%d, %%[tstepR]" % (RS_R, RS_R))
        e.raw("s_addc_u32 s%d, s%d, 0" % (RS_R + 1, RS_R + 1))


def first_fragments(e, buf):
    """all fragments of K step 0 of the stage in `buf` (h planes -> set 0)"""
    for which in ("w", "a"):
        for plane in (0, 2, 1):
            for idx in range(2):
                frag_read(e, buf, 0, which, plane, 0, idx)


def body(epi):
    e = Emit()
    descriptor(e, RS_A, "%[alo]", "%[ahi]")
    descriptor(e, RS_W, "%[wlo]", "%[whi]")
    descriptor(e, RS_C, "%[clo]", "%[chi]")
    descriptor(e, RS_B, "%[blo]", "%[bhi]")
    if epi == "res":
        descriptor(e, RS_R, "%[rlo]", "%[rhi]")
    # per-piece global offsets: A piece s = rows 64 s + tid / 4; W piece q = 2 plane + s = rows 64 s + tid / 4 of that plane
    e.raw("v_mov_b32 v%d, %%[voA]" % VO)
    e.raw("v_add_u32 v%d, %%[passA], v%d" % (VO + 1, VO))
    e.raw("v_mov_b32 v%d, %%[voW]" % (VO + 2))
    e.raw("v_add_u32 v%d, %%[passW], v%d" % (VO + 3, VO + 2))
    for p in (1, 2):
        for s in range(2):
            e.raw("v_add_u32 v%d, %%[wplane], v%d" % (VO + 2 + 2 * p + s, VO + 2 * p + s))
    e.raw("s_mov_b32 s%d, 0xffff0000" % S_MASK)
    e.raw("s_mov_b32 s%d, 0" % S_CR)
    e.raw("s_mov_b32 s%d, %%[cstep]" % (S_CR + 1))
    if epi == "res":
        e.raw("s_mov_b32 s%d, 0" % S_RR)
        e.raw("s_mov_b32 s%d, %%[rstep]" % (S_RR + 1))
    if epi == "gelu":
        for c, v in enumerate(COEF):
            e.raw("s_mov_b32 s%d, %s" % (S_GC + 2 * c, f32(v)))
        e.raw("s_mov_b32 s%d, %s" % (S_GC + 18, f32(TMAX)))
        e.raw("v_mov_b32 v%d, s%d" % (VC, S_GC + 2))
        e.raw("v_mov_b32 v%d, s%d" % (VC + 1, S_GC + 2))
    # the last A stage a request may name: tile ntile - 1, stage kt - 1  (tstepA + 128 kt = the tile stride of A)
    e.raw("s_mul_i32 s%d, %%[kt], 128" % S_T)
    e.raw("s_sub_u32 s%d, s%d, 128" % (S_LASTA, S_T))
    e.raw("s_add_u32 s%d, s%d, %%[tstepA]" % (S_T, S_T))
    e.raw("s_sub_u32 s%d, %%[ntile], 1" % S_TL)
    e.raw("s_mul_i32 s%d, s%d, s%d" % (S_T, S_T, S_TL))
    e.raw("s_add_u32 s%d, s%d, s%d" % (S_LASTA, S_LASTA, S_T))
    # ---- prologue: A0 W0 A1 W1 requested; bias; stage 0 -> buffer 0; A2 W2 requested; A1 -> buffer 1; A3 requested
    e.raw("s_mov_b32 s%d, 0" % S_SOA)
    e.raw("s_mov_b32 s%d, 0" % S_SOW)
    e.raw("s_mov_b32 s%d, 0" % S_RQA)
    e.raw("s_mov_b32 s%d, 0" % S_RQW)
    for s in range(2):
        load_a(e, 0, s)
    for q in range(6):
        load_w(e, 0, q)
    advance_a(e)
    advance_w(e)
    for s in range(2):
        load_a(e, 1, s)
    for q in range(6):
        load_w(e, 1, q)
    for t in range(8):
        e.vmem("buffer_load_dwordx4 a[%d:%d], %%[voB], s[%d:%d], 0 offen offset:%d" % (ABIAS + 4 * t, ABIAS + 4 * t + 3, RS_B, RS_B + 3, 32 * t),
               ("bias", t))
    for s in range(2):
        for fn in split_ops(e, 0, s, 0):
            fn(e)
    for q in range(6):
        write_w(e, 0, 0, q)
    advance_a(e)
    advance_w(e)
    for s in range(2):
        load_a(e, 0, s)
    for q in range(6):
        load_w(e, 0, q)
    for fn in split_ops(e, 1, 0, 1):                         # (the second piece of A1 and of A3: K step (0, 0))
        fn(e)
    advance_a(e)
    load_a(e, 1, 0)
    e.barrier()
    first_fragments(e, 0)
    e.raw("s_mov_b32 s%d, %%[ntile]" % S_TL)
    extra = residual_loads() if epi == "res" else ()
    e.raw("2:")
    kstep(e, 0, 0, True)
    kstep(e, 0, 1, False)
    kstep(e, 1, 0, False)
    kstep(e, 1, 1, False)
    e.raw("s_lshr_b32 s%d, %%[kt], 1" % S_KL)
    e.raw("s_sub_u32 s%d, s%d, 2" % (S_KL, S_KL))
    e.raw("s_cmp_eq_u32 s%d, 0" % S_KL)
    e.raw("s_cbranch_scc1 3f")
    head = (list(e.lgkm), list(e.vm))
    e.raw("1:")
    for t in range(2):
        for ks in range(2):
            kstep(e, t, ks, False)
    assert (e.lgkm, e.vm) == head, "loop body does not reproduce its head state"
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_KL, S_KL))
    e.raw("s_cmp_lg_u32 s%d, 0" % S_KL)
    e.raw("s_cbranch_scc1 1b")
    e.raw("3:")
    kstep(e, 0, 0, False)
    kstep(e, 0, 1, False)
    kstep(e, 1, 0, False)
    kstep(e, 1, 1, False, extra)
    epilogue(e, epi)
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_TL, S_TL))
    e.raw("s_cmp_lg_u32 s%d, 0" % S_TL)
    e.raw("s_cbranch_scc1 2b")
    e.drain()
    return e.lines


def main():
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % i for i in range(160)] + ['"v%d"' % i for i in range(NV)] + \
           ['"s%d"' % i for i in range(40, 94)]
    with open(OUT, "w") as f:
        f.write("// generated by tools/gen_gemm_x3_t4.py -- do not edit\n")
        for name, epi in (("U_GELU", "gelu"), ("U_RES", "res")):
            lines = body(epi)
            f.write("#define PIPS_X3T4_%s_TEXT \\\n" % name)
            for ln in lines:
                f.write('    "%s\\n\\t" \\\n' % ln)
            f.write('    ""\n\n')
            print("PIPS_X3T4_%s_TEXT: %d instructions, %d MFMAs" % (name, len(lines), sum("v_mfma" in ln for ln in lines)))
        f.write("#define PIPS_X3T4_CLOBBER " + ", ".join(clob) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
