#!/usr/bin/env python
"""Emit pips_amd/csrc/gemm_f32_t4_asm.inc: the bodies of the kernels of gemm_f32_t4.hip, each ONE assembly statement -- the mixer's
exact-fp32 Linears (v_mfma_f32_32x32x2_f32, the arithmetic and K order of igemm_f32_kernel) on four waves, one per SIMD, with a
static schedule in the style of tools/gen_gemm_bf16_t4up.py: every wave owns a 64 x 64 output block (2 x 2 MFMA blocks, 64
AccVGPR accumulators), operands go global -> registers -> LDS two stages ahead (a stage = 32 K values per wave = 64 MFMAs), two
LDS buffers, ONE barrier per stage, fragments of the next 8 K values read while the MFMAs of the current 8 run.

Two block shapes:
    U   128 x 128 tile, waves 2 x 2, a stage = 32 K values (rows 0..127 of the LDS image: A, 128..255: W); the block walks
        `ntile` consecutive row tiles of one column tile with the pipeline running on across the tile boundary
    D   64 x 64 tile, every wave the whole tile on ONE of the four groups of 8 K values of each 32-wide stage (a stage = 16 MFMAs,
        four stages in flight in registers); the four partial tiles are summed through LDS in a fixed order (((0 + 1) + 2) + 3)
        and every wave finishes ONE of the four 32 x 32 blocks -- for the problems with too few 128 x 128 tiles to fill the
        chip (M = 2048, N = 512).  (First built with every wave on a quarter of a 128-wide stage: 64 KiB per stage and CU had
        to arrive before the first MFMA, 3 us of prologue.)
    E   shape D with the operands staged by LDS-DMA (see body_e below)
and three epilogues: + bias + exact GELU (U), + bias + residual (U, D, E).

LDS image of a stage: rows of 32 K values (128 B) at a stride of 144 B -- what makes the ds_read_b128 fragment reads
(lane -> row l & 31, 16 bytes at 32 kk + 16 (l >> 5)) conflict-free in every lane group; the lane halves take interleaved
groups of four K values, the same on A and W, so that one 16-byte read per operand feeds four MFMAs (igemm_f32_kernel's order:
results are bitwise those of that kernel's KS = 1 form for shape U).

Accumulators are C^T (the W fragment is the MFMA's A operand): a lane holds output row l & 31 of an MFMA block and, per
register quad q, the four consecutive columns 8 q + 4 (l >> 5) ..: 16-byte bias / residual loads and stores.

Registers (all clobbered):
    a[0:63]      accumulators: MFMA block (i, j) = rows 32 i.., columns 32 j.. of the wave block -> a[16 (i + 2 j) : +15]
    v[0:15]      fragments, set 0: A0 A1 W0 W1 (4 registers each); v[16:31] set 1
    v[32:95]     the stages in flight (U: 2 x 8 pieces of 16 B per thread, D: 4 x 4), then per-piece global byte offsets, bias
                 quads, residual quads (GELU form: the polynomial's second coefficient), the epilogue's temporaries (class
                 Shape); v[216:255] stay with the compiler
    s[40:59] buffer descriptors A, W, C, bias, R;  s[60:61] / s[82:83] row-block offsets in C / R;  s[62:81] GELU constants;
    s[84:95] loop state
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import asm_guards as G  # noqa: E402  (wait-state guards: the numbers live in tools/asm_hazard_lint.py)
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("PIPS_GEN_OUT", os.path.join(HERE, "..", "pips_amd", "csrc", "gemm_f32_t4_asm.inc"))
POLICY = os.environ.get("PIPS_GEN_STORE_POLICY", "")      # tuning builds: cache-policy bits of the output stores, e.g. " sc1"

LDROW = 144                         # LDS row stride in bytes
FA = [0, 16]
FW = [8, 24]
ST = 32
NV = 216                            # v[216:255] are left to the compiler (the statement's vector inputs)
RS_A, RS_W, RS_C, RS_B, RS_R = 40, 44, 48, 52, 56
S_CR, S_RR = 60, 82
S_GC = 62                           # GELU constants: c0..c8 at 62, 64, .., 78 (one per even register), TMAX at 80
S_KL, S_SOA, S_SOW, S_TL, S_RQK, S_LASTA, S_T, S_KSTEP = 84, 85, 86, 87, 88, 89, 90, 91
# exponent polynomial A8 of common.h's gelu_exact2, highest power first, and its clamp
COEF = [3.208326405e-07, -6.917509381e-06, 6.041429151e-05, -2.428356966e-04, -5.105399032e-05, 6.989960559e-03,
        -5.246259645e-02, -4.592153430e-01, -1.151104689e+00]
TMAX = 5.65685425


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
        """every load issued so far has landed (stores may stay in flight)"""
        self.need_vm({t for t in self.vm if t[0] != "out"})

    def barrier(self):
        if self.lgkm:
            self.lines.append("s_waitcnt lgkmcnt(0)")
            self.lgkm = []
        self.lines.append("s_barrier")

    def drain(self):
        self.lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.lgkm, self.vm = [], []


class Shape:
    def __init__(self, name, epi):
        self.name = name
        self.npa = 4 if name == "U" else 2          # staged pieces per thread and stage: A, W
        self.np = 2 * self.npa
        self.nsub = 4 if name == "U" else 1         # groups of 8 K values a wave multiplies per stage
        self.ring = 2 if name == "U" else 4         # stages in flight in registers
        self.unroll = 2 if name == "U" else 4       # lcm(2 LDS buffers, ring)
        self.blocks = [(i, j) for j in range(2) for i in range(2)]
        # registers behind the staged pieces (v[32:95]): per-piece offsets, bias quads, residual quads, the epilogue's temporaries
        if name == "U":
            self.vo, self.bias, self.res, self.ep = 96, 104, 136, (138 if epi == "gelu" else 200)
        else:
            self.vo, self.bias, self.res, self.ep = 96, 100, 116, 132
        self.vc = self.res                          # GELU form: the polynomial's second coefficient as a vector pair


def acc(i, j):
    return 16 * (i + 2 * j)


D4 = os.environ.get("PIPS_GEN_D4", "1") == "1"      # shape D on four LDS buffers (0: two)
DSTAGE = 128 * LDROW


def frag_read(e, par, fset, kk, which, idx):
    """fragment `idx` (A: row block i, W: column block j) of the 8 K values kk of the stage in LDS buffer `par`, into set `fset`"""
    reg = (FA if which == "a" else FW)[fset] + 4 * idx
    off = idx * 32 * LDROW + kk * 32
    e.lds("ds_read_b128 v[%d:%d], %%[r%s%d] offset:%d" % (reg, reg + 3, "A" if which == "a" else "W", par, off), ("f" + which, fset, idx))


def frag_read_d(e, buf, fset, which, idx):
    """shape D on four buffers: one base register, the buffer in the offset"""
    reg = (FA if which == "a" else FW)[fset] + 4 * idx
    e.lds("ds_read_b128 v[%d:%d], %%[r%s0] offset:%d" % (reg, reg + 3, "A" if which == "a" else "W", buf * DSTAGE + idx * 32 * LDROW),
          ("f" + which, fset, idx))


FRAG_ORDER = [("w", 0), ("a", 0), ("a", 1), ("w", 1)]


def mfma(e, fset, c, i, j, zero):
    e.need_lds({("fw", fset, j), ("fa", fset, i)})
    a = acc(i, j)
    e.raw("v_mfma_f32_32x32x2_f32 a[%d:%d], v%d, v%d, %s" %
          (a, a + 15, FW[fset] + 4 * j + c, FA[fset] + 4 * i + c, "0" if zero else "a[%d:%d]" % (a, a + 15)))


def store_piece(e, sh, par, ring, s):
    """staged piece s of register set `ring` -> LDS buffer `par`"""
    e.need_vm({("st", ring, s)})
    reg = ST + 4 * (ring * sh.np + s)
    if s < sh.npa:
        e.lds("ds_write_b128 %%[wA%d], v[%d:%d] offset:%d" % (par, reg, reg + 3, s * 32 * LDROW), ("wr", s))
    else:
        e.lds("ds_write_b128 %%[wW%d], v[%d:%d] offset:%d" % (par, reg, reg + 3, (s - sh.npa) * 32 * LDROW), ("wr", s))


def store_piece_d(e, sh, buf, ring, s):
    e.need_vm({("st", ring, s)})
    reg = ST + 4 * (ring * sh.np + s)
    name, k = ("A", s) if s < sh.npa else ("W", s - sh.npa)
    e.lds("ds_write_b128 %%[w%s0], v[%d:%d] offset:%d" % (name, reg, reg + 3, buf * DSTAGE + k * 32 * LDROW), ("wr", s))


def load_piece(e, sh, ring, s):
    reg = ST + 4 * (ring * sh.np + s)
    if s < sh.npa:
        e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], s%d offen" % (reg, reg + 3, sh.vo + s, RS_A, RS_A + 3, S_SOA), ("st", ring, s))
    else:
        e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], s%d offen" % (reg, reg + 3, sh.vo + s, RS_W, RS_W + 3, S_SOW), ("st", ring, s))


def descriptor(e, base, lo, hi):
    e.raw("s_mov_b32 s%d, %s" % (base, lo))
    e.raw("s_and_b32 s%d, %s, 0xffff" % (base + 1, hi))
    e.raw("s_mov_b32 s%d, 0x7fffffff" % (base + 2))
    e.raw("s_mov_b32 s%d, 0x00020000" % (base + 3))


def advance_request(e):
    """offsets of the stage to request next: one stage further; behind a tile's last stage comes the next tile's first (same W tile,
    A rows + the tile stride); behind the block's last tile the last stage again (never used)"""
    e.raw("s_add_u32 s%d, s%d, 1" % (S_RQK, S_RQK))
    e.raw("s_add_u32 s%d, s%d, s%d" % (S_SOA, S_SOA, S_KSTEP))
    e.raw("s_add_u32 s%d, s%d, s%d" % (S_SOW, S_SOW, S_KSTEP))
    e.raw("s_add_u32 s%d, s%d, %%[tstepA]" % (S_T, S_SOA))               # soA - K bytes + tile stride
    e.raw("s_cmp_eq_u32 s%d, %%[kt]" % S_RQK)                            # (the selects below read SCC: nothing in between may write it)
    e.raw("s_cselect_b32 s%d, 0, s%d" % (S_RQK, S_RQK))
    e.raw("s_cselect_b32 s%d, 0, s%d" % (S_SOW, S_SOW))
    e.raw("s_cselect_b32 s%d, s%d, s%d" % (S_SOA, S_T, S_SOA))
    e.raw("s_min_u32 s%d, s%d, s%d" % (S_SOA, S_SOA, S_LASTA))


def stage(e, sh, t, first, extra=()):
    """Stage t of an unrolled group: 16 nsub MFMAs on LDS buffer t & 1; stage t + 1 goes from register set (t + 1) % ring to the
    other buffer and stage t + 1 + ring is requested into that set; `extra`: operations (callables) of the tile's last stage
    (residual prefetch) for the free slots."""
    advance_request(e)
    par, ring, nsub = t & 1, (t + 1) % sh.ring, sh.nsub
    g0 = t * nsub                                             # fragment set of the 8 K values kk of this stage: (g0 + kk) & 1
    slots = {}

    def put(n, op):
        slots.setdefault(n, []).append(op)

    ops = []
    for s in range(sh.np):
        ops += [("st", s), ("ld", s)]
    if nsub == 4:
        for kk in range(3):
            for r, (which, idx) in enumerate(FRAG_ORDER):
                put(16 * kk + 1 + r, ("fr", par, (g0 + kk + 1) & 1, kk + 1, which, idx))
        free = [n for n in range(5, 47) if n not in slots]
        step = len(free) / float(len(ops))
        for k, op in enumerate(ops):
            put(free[int(k * step)], op)
        bar, nxt = 47, 49
        rest = [n for n in range(53, 64)] + [n for n in range(5, 47) if n not in slots]
    else:
        for k, op in enumerate(ops):
            put(k, op)
        bar, nxt = 8, 9
        rest = [13, 14, 15, 12]
    put(bar, ("bar",))
    for r, (which, idx) in enumerate(FRAG_ORDER):             # the next stage's first fragments, from the other buffer
        put(nxt + r, ("fr", 1 - par, (g0 + nsub) & 1, 0, which, idx))
    for k, op in enumerate(extra):
        put(rest[k], ("extra", op))
    n = 0
    for kk in range(nsub):
        for c in range(4):
            for (i, j) in sh.blocks:
                mfma(e, (g0 + kk) & 1, c, i, j, first and kk == 0 and c == 0)
                for op in slots.get(n, []):
                    if op[0] == "fr":
                        frag_read(e, *op[1:])
                    elif op[0] == "st":
                        store_piece(e, sh, 1 - par, ring, op[1])
                    elif op[0] == "ld":
                        load_piece(e, sh, ring, op[1])
                    elif op[0] == "bar":
                        e.barrier()
                    else:
                        op[1](e)
                n += 1


def stage_d4(e, sh, t, first, extra=()):
    """Shape D on FOUR LDS buffers, stage t of a group of four: 16 MFMAs on buffer t; the fragments of stage t + 1 are read at
    once (its buffer was completed before the barrier of stage t - 1); stage t + 2 goes from register set (t + 2) % 4 to its
    buffer and stage t + 6 is requested into that set; the barrier sits at the stage's end, well behind its writes -- with two
    buffers the writes, the barrier and the next stage's reads had to follow each other inside 16 MFMAs and every wait was live."""
    advance_request(e)
    ring = (t + 2) % 4
    slots = {}

    def put(n, op):
        slots.setdefault(n, []).append(op)

    for r, (which, idx) in enumerate(FRAG_ORDER):
        put(1 + r, ("fr", which, idx))
    for s in range(sh.np):
        put(5 + 2 * s, ("st", s))
        put(6 + 2 * s, ("ld", s))
    put(15, ("bar",))
    rest = [13, 14, 12, 11]
    for k, op in enumerate(extra):
        put(rest[k], ("extra", op))
    n = 0
    for c in range(4):
        for (i, j) in sh.blocks:
            mfma(e, t & 1, c, i, j, first and c == 0)
            for op in slots.get(n, []):
                if op[0] == "fr":
                    frag_read_d(e, (t + 1) % 4, (t + 1) & 1, op[1], op[2])
                elif op[0] == "st":
                    store_piece_d(e, sh, ring, ring, op[1])
                elif op[0] == "ld":
                    load_piece(e, sh, ring, op[1])
                elif op[0] == "bar":
                    e.barrier()
                else:
                    op[1](e)
            n += 1


def gelu4(e, X, T, Q, VC):
    """exact GELU of the 8 values v[X:X+7] in place: gelu_exact2's arithmetic (common.h), four pairs side by side"""
    for p in range(4):
        for h in range(2):
            e.raw("v_min_f32_e64 v%d, |v%d|, s%d" % (T + 2 * p + h, X + 2 * p + h, S_GC + 18))
    for p in range(4):          # q = c0 t + c1
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


def res_index(sh, i, j, q):
    return sh.res + 4 * (q + 4 * j + 8 * i)


def residual_loads(sh):
    """the tile's residual quads -> v[RES..] (issued in the free slots of the tile's last stage)"""
    ops = []
    if sh.name == "U":
        for i in range(2):
            for j in range(2):
                for q in range(4):
                    def op(e, i=i, j=j, q=q):
                        r = res_index(sh, i, j, q)
                        e.vmem("buffer_load_dwordx4 v[%d:%d], %%[voR], s[%d:%d], s%d offen offset:%d" %
                               (r, r + 3, RS_R, RS_R + 3, S_RR + i, (32 * j + 8 * q) * 4), ("res", i, j, q))
                    ops.append(op)
    else:
        for q in range(4):
            def op(e, q=q):
                r = res_index(sh, 0, 0, q)
                e.vmem("buffer_load_dwordx4 v[%d:%d], %%[voR], s[%d:%d], 0 offen offset:%d" % (r, r + 3, RS_R, RS_R + 3, 32 * q),
                       ("res", 0, 0, q))
            ops.append(op)
    return ops


def epilogue_u(e, sh, epi):
    """the wave's 64 x 64 block: + bias, GELU or + residual, 16-byte stores; left with its last stores in flight"""
    e.need_loads()
    G.emit_mfma_result_guard(e.raw, "v_mfma_f32_32x32x2_f32")      # the tile's last MFMAs -> v_accvgpr_read
    EP, BIAS = sh.ep, sh.bias
    sets = [(EP, EP + 8, EP + 16), (EP + 24, EP + 32, EP + 40)] if epi == "gelu" else [(EP, 0, 0), (EP + 8, 0, 0)]
    k = 0
    for i in range(2):
        for j in range(2):
            for qq in range(2):
                X, T, Q = sets[k & 1]
                e.need_vm({("out", k - 2)})                  # the stores that read this register set have taken their data
                for h in range(2):
                    q = 2 * qq + h
                    for r in range(4):
                        e.raw("v_accvgpr_read_b32 v%d, a%d" % (X + 4 * h + r, acc(i, j) + 4 * q + r))
                for h in range(2):
                    b = BIAS + 4 * (4 * j + 2 * qq + h)
                    for p in range(2):
                        e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" %
                              (X + 4 * h + 2 * p, X + 4 * h + 2 * p + 1, X + 4 * h + 2 * p, X + 4 * h + 2 * p + 1, b + 2 * p, b + 2 * p + 1))
                if epi == "gelu":
                    gelu4(e, X, T, Q, sh.vc)
                else:
                    for h in range(2):
                        r = res_index(sh, i, j, 2 * qq + h)
                        for p in range(2):
                            e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" %
                                  (X + 4 * h + 2 * p, X + 4 * h + 2 * p + 1, X + 4 * h + 2 * p, X + 4 * h + 2 * p + 1, r + 2 * p, r + 2 * p + 1))
                for h in range(2):
                    e.vmem("buffer_store_dwordx4 v[%d:%d], %%[voC], s[%d:%d], s%d offen offset:%d" %
                           (X + 4 * h, X + 4 * h + 3, RS_C, RS_C + 3, S_CR + i, (32 * j + 8 * (2 * qq + h)) * 4) + POLICY, ("out", k))
                k += 1
    # the next tile's rows of C / R (scalar writes: the stores in flight have read their descriptor)
    e.raw("s_add_u32 s%d, s%d, %%[tstepC]" % (RS_C, RS_C))
    e.raw("s_addc_u32 s%d, s%d, 0" % (RS_C + 1, RS_C + 1))
    if epi == "res":
        e.raw("s_add_u32 s%d, s%d, %%[tstepR]" % (RS_R, RS_R))
        e.raw("s_addc_u32 s%d, s%d, 0" % (RS_R + 1, RS_R + 1))


def epilogue_d(e, sh):
    """the four partial 64 x 64 tiles -> LDS -> every wave sums ONE 32 x 32 block in the order ((0 + 1) + 2) + 3, + bias + residual"""
    e.need_loads()
    e.barrier()                                              # every wave is done with the stage buffers
    G.emit_mfma_result_guard(e.raw, "v_mfma_f32_32x32x2_f32")      # the tile's last MFMAs -> v_accvgpr_read
    for b in range(4):                                       # MFMA block b = i + 2 j, quad q -> red[ks][4 b + q][lane] (16 B each)
        for q in range(4):
            a = 16 * b + 4 * q
            e.lds("ds_write_b128 %%[redW], a[%d:%d] offset:%d" % (a, a + 3, (4 * b + q) * 1024), ("rw", b, q))
    e.barrier()
    EP, BIAS = sh.ep, sh.bias
    P = EP                                                   # 4 partials x 2 quads x 4 registers, twice
    for half in range(2):
        for ks in range(4):
            for q2 in range(2):
                q = 2 * half + q2
                r = P + 4 * (2 * ks + q2)
                e.lds("ds_read_b128 v[%d:%d], %%[redR] offset:%d" % (r, r + 3, ks * 16384 + q * 1024), ("rr", ks, q2))
        for q2 in range(2):
            q = 2 * half + q2
            X = P + 4 * q2
            for ks in range(1, 4):
                e.need_lds({("rr", 0, q2), ("rr", ks, q2)})
                r = P + 4 * (2 * ks + q2)
                for p in range(2):
                    e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (X + 2 * p, X + 2 * p + 1, X + 2 * p, X + 2 * p + 1, r + 2 * p, r + 2 * p + 1))
            b = BIAS + 4 * q
            rr = res_index(sh, 0, 0, q)
            O = EP + 32 + 4 * q                              # (the store's data sits outside the registers the second half reads into)
            for p in range(2):
                e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (X + 2 * p, X + 2 * p + 1, X + 2 * p, X + 2 * p + 1, b + 2 * p, b + 2 * p + 1))
            for p in range(2):
                e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (O + 2 * p, O + 2 * p + 1, X + 2 * p, X + 2 * p + 1, rr + 2 * p, rr + 2 * p + 1))
            e.vmem("buffer_store_dwordx4 v[%d:%d], %%[voC], s[%d:%d], 0 offen offset:%d" % (O, O + 3, RS_C, RS_C + 3, 32 * q) + POLICY, ("out", q))


def body(shape, epi):
    sh = Shape(shape, epi)
    e = Emit()
    descriptor(e, RS_A, "%[alo]", "%[ahi]")
    descriptor(e, RS_W, "%[wlo]", "%[whi]")
    descriptor(e, RS_C, "%[clo]", "%[chi]")
    descriptor(e, RS_B, "%[blo]", "%[bhi]")
    if epi == "res":
        descriptor(e, RS_R, "%[rlo]", "%[rhi]")
    # per-piece global offsets: piece s = rows 32 s.. of the tile (A, then W)
    for base, v0 in (("%[voA]", sh.vo), ("%[voW]", sh.vo + sh.npa)):
        pas = "%[passA]" if v0 == sh.vo else "%[passW]"
        e.raw("v_mov_b32 v%d, %s" % (v0, base))
        for s in range(1, sh.npa):
            e.raw("v_add_u32 v%d, %s, v%d" % (v0 + s, pas, v0 + s - 1))
    e.raw("s_mov_b32 s%d, 128" % S_KSTEP)
    if shape == "U":
        e.raw("s_mov_b32 s%d, 0" % S_CR)
        e.raw("s_mov_b32 s%d, %%[cstep]" % (S_CR + 1))
        if epi == "res":
            e.raw("s_mov_b32 s%d, 0" % S_RR)
            e.raw("s_mov_b32 s%d, %%[rstep]" % (S_RR + 1))
    if epi == "gelu":
        for c, v in enumerate(COEF):
            e.raw("s_mov_b32 s%d, %s" % (S_GC + 2 * c, f32(v)))
        e.raw("s_mov_b32 s%d, %s" % (S_GC + 18, f32(TMAX)))
        e.raw("v_mov_b32 v%d, s%d" % (sh.vc, S_GC + 2))
        e.raw("v_mov_b32 v%d, s%d" % (sh.vc + 1, S_GC + 2))
    # the last stage a request may name: tile ntile - 1, stage kt - 1  (tstepA + kstep kt = the tile stride of A)
    e.raw("s_mul_i32 s%d, s%d, %%[kt]" % (S_T, S_KSTEP))
    e.raw("s_sub_u32 s%d, s%d, s%d" % (S_LASTA, S_T, S_KSTEP))          # kstep (kt - 1)
    e.raw("s_add_u32 s%d, s%d, %%[tstepA]" % (S_T, S_T))
    e.raw("s_sub_u32 s%d, %%[ntile], 1" % S_TL)
    e.raw("s_mul_i32 s%d, s%d, s%d" % (S_T, S_T, S_TL))
    e.raw("s_add_u32 s%d, s%d, s%d" % (S_LASTA, S_LASTA, S_T))
    # ---- stages 0 .. ring - 1 -> register sets, bias, stage 0 -> LDS buffer 0, stage `ring` requested, fragments of the first 8 K values
    e.raw("s_mov_b32 s%d, 0" % S_SOA)
    e.raw("s_mov_b32 s%d, 0" % S_SOW)
    e.raw("s_mov_b32 s%d, 0" % S_RQK)
    nb = 8 if shape == "U" else 4
    d4 = shape == "D" and D4
    for r in range(sh.ring):
        if r:
            advance_request(e)
        for s in range(sh.np):
            load_piece(e, sh, r, s)
    for t in range(nb):                                      # bias quad t: columns 8 t + 4 (l >> 5) .. of the wave's (U) / the block's (D)
        e.vmem("buffer_load_dwordx4 v[%d:%d], %%[voB], s[%d:%d], 0 offen offset:%d" % (sh.bias + 4 * t, sh.bias + 4 * t + 3, RS_B, RS_B + 3, 32 * t),
               ("bias", t))
    if d4:                                                   # stages 0, 1 -> buffers 0, 1; stages 4, 5 requested
        for r in range(2):
            for s in range(sh.np):
                store_piece_d(e, sh, r, r, s)
            advance_request(e)
            for s in range(sh.np):
                load_piece(e, sh, r, s)
        e.barrier()
        for which, idx in FRAG_ORDER:
            frag_read_d(e, 0, 0, which, idx)
    else:
        for s in range(sh.np):
            store_piece(e, sh, 0, 0, s)
        advance_request(e)
        for s in range(sh.np):
            load_piece(e, sh, 0, s)
        e.barrier()
        for which, idx in FRAG_ORDER:
            frag_read(e, 0, 0, 0, which, idx)
    e.raw("s_mov_b32 s%d, %%[ntile]" % S_TL)
    extra = residual_loads(sh) if epi == "res" else ()
    U = sh.unroll
    stage_fn = stage_d4 if d4 else stage
    e.raw("2:")
    for t in range(U):
        stage_fn(e, sh, t, t == 0)
    e.raw("s_lshr_b32 s%d, %%[kt], %d" % (S_KL, 1 if U == 2 else 2))
    e.raw("s_sub_u32 s%d, s%d, 2" % (S_KL, S_KL))
    e.raw("s_cmp_eq_u32 s%d, 0" % S_KL)
    e.raw("s_cbranch_scc1 3f")
    head = (list(e.lgkm), list(e.vm))
    e.raw("1:")
    for t in range(U):
        stage_fn(e, sh, t, False)
    assert (e.lgkm, e.vm) == head, "loop body does not reproduce its head state"
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_KL, S_KL))
    e.raw("s_cmp_lg_u32 s%d, 0" % S_KL)
    e.raw("s_cbranch_scc1 1b")
    e.raw("3:")
    for t in range(U):
        stage_fn(e, sh, t, False, extra if t == U - 1 else ())
    if shape == "U":
        epilogue_u(e, sh, epi)
        e.raw("s_sub_u32 s%d, s%d, 1" % (S_TL, S_TL))
        e.raw("s_cmp_lg_u32 s%d, 0" % S_TL)
        e.raw("s_cbranch_scc1 2b")
    else:
        epilogue_d(e, sh)
    e.drain()
    return e.lines


# ---------------------------------------------------------------------------------------------------------------------------------
# Shape E = shape D with the operands staged by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write).  Shape D's K loop
# pays for its staging instructions (4 loads + 4 ds_writes per 16 MFMAs: 130 TF against 158 without them).  LDS image of a stage:
# 128 dense rows of 128 bytes (A rows 0..63, W rows 64..127), chunk slot j of row r holding global chunk j ^ ((r >> 1) & 7) -- the
# swizzle is applied on the GLOBAL side (a DMA instruction writes 1 KiB linearly: lane l -> 16 bytes at 16 l) and makes the fragment
# reads conflict-free.  Wave w stages rows 32 w .. 32 w + 31 (four DMA instructions of 8 rows), FOUR buffers, a stage requested three
# stages ahead, waited for (by its own wave) before the barrier at the end of the stage two before its use.
EBUF = 128 * 128


def body_e():
    sh = Shape("D", "res")
    sh.vo = 96
    e = Emit()
    descriptor(e, RS_A, "%[xlo]", "%[xhi]")                  # this wave's operand (A for waves 0, 1; W for waves 2, 3)
    descriptor(e, RS_C, "%[clo]", "%[chi]")
    descriptor(e, RS_B, "%[blo]", "%[bhi]")
    descriptor(e, RS_R, "%[rlo]", "%[rhi]")
    e.raw("v_mov_b32 v%d, %%[vo0]" % sh.vo)
    e.raw("v_mov_b32 v%d, %%[vo1]" % (sh.vo + 1))
    e.raw("v_add_u32 v%d, %%[pass2], v%d" % (sh.vo + 2, sh.vo))
    e.raw("v_add_u32 v%d, %%[pass2], v%d" % (sh.vo + 3, sh.vo + 1))
    e.raw("s_sub_u32 s%d, %%[kt], 1" % S_LASTA)
    e.raw("s_lshl_b32 s%d, s%d, 7" % (S_LASTA, S_LASTA))     # the last stage a request may name
    e.raw("s_mov_b32 s%d, 0" % S_SOA)

    def dma(buf, k):
        e.raw("s_add_u32 m0, %%[ldsw], %d" % (buf * EBUF + 8 * k * 128))
        G.emit_m0_guard(e.raw)                               # SALU writes M0 -> the LDS-DMA load reads it: one wait state
        e.vmem("buffer_load_dwordx4 v%d, s[%d:%d], s%d offen lds" % (sh.vo + k, RS_A, RS_A + 3, S_SOA), ("st", buf, k))

    def advance():
        e.raw("s_add_u32 s%d, s%d, 128" % (S_SOA, S_SOA))
        e.raw("s_min_u32 s%d, s%d, s%d" % (S_SOA, S_SOA, S_LASTA))

    def fread(buf, fset, which, idx):
        reg = (FA if which == "a" else FW)[fset] + 4 * idx
        e.lds("ds_read_b128 v[%d:%d], %%[r%s0] offset:%d" % (reg, reg + 3, "A" if which == "a" else "W", buf * EBUF + idx * 32 * 128),
              ("f" + which, fset, idx))

    for t in range(4):                                       # bias quad t of the block's 32 columns
        e.vmem("buffer_load_dwordx4 v[%d:%d], %%[voB], s[%d:%d], 0 offen offset:%d" % (sh.bias + 4 * t, sh.bias + 4 * t + 3, RS_B, RS_B + 3, 32 * t),
               ("bias", t))
    for b in range(3):                                       # stages 0, 1, 2 requested
        if b:
            advance()
        for k in range(4):
            dma(b, k)
    e.need_vm({("st", 1, k) for k in range(4)})              # stages 0 and 1 of this wave have landed
    e.barrier()
    for which, idx in FRAG_ORDER:
        fread(0, 0, which, idx)

    def stage_e(t, first, extra=()):
        advance()
        slots = {}

        def put(n, fn):
            slots.setdefault(n, []).append(fn)

        for r, (which, idx) in enumerate(FRAG_ORDER):
            put(1 + r, lambda which=which, idx=idx: fread((t + 1) % 4, (t + 1) & 1, which, idx))
        for k in range(4):
            put(5 + 2 * k, lambda k=k: dma((t + 3) % 4, k))
        put(15, lambda: (e.need_vm({("st", (t + 2) % 4, k) for k in range(4)}), e.barrier()))
        for k, op in enumerate(extra):
            put([13, 14, 12, 10][k], lambda op=op: op(e))
        n = 0
        for c in range(4):
            for (i, j) in sh.blocks:
                mfma(e, t & 1, c, i, j, first and c == 0)
                for fn in slots.get(n, []):
                    fn()
                n += 1

    extra = residual_loads(sh)
    for t in range(4):
        stage_e(t, t == 0)
    e.raw("s_lshr_b32 s%d, %%[kt], 2" % S_KL)
    e.raw("s_sub_u32 s%d, s%d, 2" % (S_KL, S_KL))
    e.raw("s_cmp_eq_u32 s%d, 0" % S_KL)
    e.raw("s_cbranch_scc1 3f")
    head = (list(e.lgkm), list(e.vm))
    e.raw("1:")
    for t in range(4):
        stage_e(t, False)
    assert (e.lgkm, e.vm) == head, "loop body does not reproduce its head state"
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_KL, S_KL))
    e.raw("s_cmp_lg_u32 s%d, 0" % S_KL)
    e.raw("s_cbranch_scc1 1b")
    e.raw("3:")
    for t in range(4):
        stage_e(t, False, extra if t == 3 else ())
    epilogue_d(e, sh)
    e.drain()
    return e.lines


def main():
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % i for i in range(64)] + ['"v%d"' % i for i in range(NV)] + \
           ['"s%d"' % i for i in range(40, 92)]
    with open(OUT, "w") as f:
        f.write("// generated by tools/gen_gemm_f32_t4.py -- do not edit\n")
        for name, shape, epi in (("U_GELU", "U", "gelu"), ("U_RES", "U", "res"), ("D_RES", "D", "res")):
            lines = body(shape, epi)
            f.write("#define PIPS_F32T4_%s_TEXT \\\n" % name)
            for ln in lines:
                f.write('    "%s\\n\\t" \\\n' % ln)
            f.write('    ""\n\n')
            print("PIPS_F32T4_%s_TEXT: %d instructions, %d MFMAs" % (name, len(lines), sum("v_mfma" in ln for ln in lines)))
        lines = body_e()
        f.write("#define PIPS_F32T4_E_RES_TEXT \\\n")
        for ln in lines:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n\n')
        print("PIPS_F32T4_E_RES_TEXT: %d instructions, %d MFMAs" % (len(lines), sum("v_mfma" in ln for ln in lines)))
        f.write("#define PIPS_F32T4_CLOBBER " + ", ".join(clob) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
