#!/usr/bin/env python
"""Emit pips_amd/csrc/gemm_bf16_t4up_asm.inc: the whole body of gemm_bf16_t4_gelu_kernel (gemm_bf16_t4.hip) as ONE assembly
statement -- the bf16 up-projection C = bf16(gelu(bf16(A.W^T + bias))) (K = 512) on 256 x 256 tiles, four waves (one per SIMD),
wave tile 128 x 128 on v_mfma_f32_16x16x32_bf16 with all 256 AccVGPRs as accumulators, operands global -> registers -> LDS two
K blocks ahead, one LDS buffer (64 KiB), two barriers per 64 K values, a block walking `ntile` consecutive row tiles of one
column tile with the K pipeline running on across the tile boundary (the next tile's first blocks arrive under this tile's
epilogue).  Same issue model / counted waits as tools/gen_gemm_bf16_t4.py.

Registers (all clobbered by the statement):
    a[0:255]     accumulators: tile (i, j) = rows 16 i.., columns 16 j.. of the wave tile -> a[4 (i + 8 j) : +3]
                 (C^T: lane = output row r16, registers = 4 consecutive columns; the W rows sit in LDS permuted so that the
                 tiles 2 j', 2 j' + 1 together give a lane 8 consecutive columns: one 16-byte bf16 store)
    v[0:31]      A fragments of K step 0 (i = 0..7), v[32:63] W fragments of K step 0 (j = 0..7)
    v[64:95]     A fragments of K step 1,            v[96:127] W fragments of K step 1   (v[64:127]: the epilogue's temporaries)
    v[128:191]   the K block in flight (16 pieces of 16 B per thread: 8 of A, 8 of W)
    v[192:207]   per-piece global byte offsets
    s[40:55]     buffer descriptors A, W, C, bias;  s[56:61] loop state;  s[62:81] GELU constants;  s[82:89] row-block offsets of C
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import asm_guards as G  # noqa: E402  (wait-state guards: the numbers live in tools/asm_hazard_lint.py)
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("PIPS_GEN_OUT", os.path.join(HERE, "..", "pips_amd", "csrc", "gemm_bf16_t4up_asm.inc"))
ABL = os.environ.get("PIPS_GEN_ABLATE", "")                 # timing probes (wrong results): gelu = no GELU arithmetic in the epilogue; vmwait / barrier / fragwait =
                                                            # the K loop without its waits on staged loads / its barriers / its waits on fragments
POLICY = os.environ.get("PIPS_GEN_STORE_POLICY", "")      # tuning builds: cache-policy bits of the output stores, e.g. " sc1"

NI, NJ = 8, 8                       # 16-row / 16-column blocks of the wave tile
FA = [0, 64]
FW = [32, 96]
ST = 128
NP = 16                             # staged pieces per thread and K block: 8 of A, 8 of W
VO = 192
E = 64                              # epilogue temporaries v[64:127]
RS_A, RS_W, RS_C, RS_B = 40, 44, 48, 52
S_KL, S_SOA, S_SOW, S_TL, S_RQK, S_LASTA = 56, 57, 58, 59, 60, 61
S_GC = 62                           # GELU constants, one per even register: c5..c0 (62..72), TMAX (80)
VC = 208                            # v[208:209]: the second coefficient as a vector pair (an instruction takes ONE scalar operand)
S_CR = 82                           # s[82:89]: i * 16 * ldc * 2
S_T = 90
KT = 8                              # K = 512
# exponent polynomial of the GELU, erfc(t / sqrt 2) = exp2(t A(t)), highest power first.  Degree 5 (a weighted minimax fit like
# common.h's degree-8 A8): max |error| of the GELU 5.1e-6, 3.3e-5 relative where |gelu| > 1e-3 -- two orders below the bf16
# rounding of the result (2^-9), five times below the LDS table it replaces (2.4e-5); three packed FMAs per pair fewer than A8
COEF = [2.554670494e-05, -6.529359078e-04, 7.452824686e-03, -5.192063601e-02, -4.602978599e-01, -1.150685204e+00]
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

    def need_lds(self, tags, emit=True):
        idx = [k for k, t in enumerate(self.lgkm) if t in tags]
        if not idx:
            return
        left = min(len(self.lgkm) - 1 - max(idx), 15)
        if emit:
            self.lines.append("s_waitcnt lgkmcnt(%d)" % left)
        self.lgkm = self.lgkm[len(self.lgkm) - left:] if left else []

    def need_vm(self, tags, emit=True):
        idx = [k for k, t in enumerate(self.vm) if t in tags]
        if not idx:
            return
        left = min(len(self.vm) - 1 - max(idx), 63)
        if emit:
            self.lines.append("s_waitcnt vmcnt(%d)" % left)
        self.vm = self.vm[len(self.vm) - left:] if left else []

    def barrier(self):
        if self.lgkm:
            self.lines.append("s_waitcnt lgkmcnt(0)")
            self.lgkm = []
        if "barrier" not in ABL:                             # (timing probe: the waves of a block run unsynchronised)
            self.lines.append("s_barrier")

    def drain(self):
        self.lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.lgkm, self.vm = [], []


def acc(i, j):
    return 4 * (i + NI * j)


def frag_read(e, ks, which, idx):
    if "frags" in ABL:                                       # (timing probe: the K loop without its fragment reads)
        return
    if which == "a":
        reg = FA[ks] + 4 * idx
        e.lds("ds_read_b128 v[%d:%d], %%[rA%d] offset:%d" % (reg, reg + 3, ks, idx * 2048), ("fa", ks, idx))
    else:
        reg = FW[ks] + 4 * idx
        e.lds("ds_read_b128 v[%d:%d], %%[rW%d] offset:%d" % (reg, reg + 3, ks, idx * 2048), ("fw", ks, idx))


# the order a K step's MFMAs want the fragments: W0, A0..A7, W1..W7
FRAG_ORDER = [("w", 0)] + [("a", i) for i in range(NI)] + [("w", j) for j in range(1, NJ)]


def mfma(e, ks, n, first):
    i, j = n % NI, n // NI
    e.need_lds({("fw", ks, j), ("fa", ks, i)}, emit="fragwait" not in ABL)     # (timing probe "fragwait": MFMAs on fragments that may not have landed)
    c = acc(i, j)
    src_c = "0" if first and ks == 0 else "a[%d:%d]" % (c, c + 3)         # a tile's first K step starts from zero
    e.raw("v_mfma_f32_16x16x32_bf16 a[%d:%d], v[%d:%d], v[%d:%d], %s" %
          (c, c + 3, FW[ks] + 4 * j, FW[ks] + 4 * j + 3, FA[ks] + 4 * i, FA[ks] + 4 * i + 3, src_c))


def store_piece(e, s):
    if "stage" in ABL:                                       # (timing probe: the K loop without its staging instructions)
        return
    e.need_vm({("st", s)}, emit="vmwait" not in ABL)         # (timing probe "vmwait": stores of registers whose load may not have landed)
    reg = ST + 4 * s
    if s < 8:
        e.lds("ds_write_b128 %%[wA], v[%d:%d] offset:%d" % (reg, reg + 3, s * 4096), ("wr", s))
    else:
        e.lds("ds_write_b128 %%[wW], v[%d:%d] offset:%d" % (reg, reg + 3, (s - 8) * 4096), ("wr", s))


def load_piece(e, s):
    if "stage" in ABL:
        return
    reg = ST + 4 * s
    if s < 8:
        e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], s%d offen" % (reg, reg + 3, VO + s, RS_A, RS_A + 3, S_SOA), ("st", s))
    else:
        e.vmem("buffer_load_dwordx4 v[%d:%d], v%d, s[%d:%d], s%d offen" % (reg, reg + 3, VO + s, RS_W, RS_W + 3, S_SOW), ("st", s))


def descriptor(e, base, lo, hi):
    e.raw("s_mov_b32 s%d, %s" % (base, lo))
    e.raw("s_and_b32 s%d, %s, 0xffff" % (base + 1, hi))
    e.raw("s_mov_b32 s%d, 0x7fffffff" % (base + 2))
    e.raw("s_mov_b32 s%d, 0x00020000" % (base + 3))


def advance_request(e):
    """offsets of the K block to request next: one block further; behind a tile's last block comes the next tile's first
    (same W tile, A rows + strideA); behind the block's last tile the last block again (never used)"""
    e.raw("s_add_u32 s%d, s%d, 1" % (S_RQK, S_RQK))
    e.raw("s_add_u32 s%d, s%d, 128" % (S_SOA, S_SOA))
    e.raw("s_add_u32 s%d, s%d, 128" % (S_SOW, S_SOW))
    e.raw("s_add_u32 s%d, s%d, %%[tstepA]" % (S_T, S_SOA))               # soA - K bytes + strideA (tstepA = strideA - 128 KT)
    e.raw("s_cmp_eq_u32 s%d, %d" % (S_RQK, KT))                          # (the selects below read SCC: nothing in between may write it)
    e.raw("s_cselect_b32 s%d, 0, s%d" % (S_RQK, S_RQK))
    e.raw("s_cselect_b32 s%d, 0, s%d" % (S_SOW, S_SOW))
    e.raw("s_cselect_b32 s%d, s%d, s%d" % (S_SOA, S_T, S_SOA))
    e.raw("s_min_u32 s%d, s%d, s%d" % (S_SOA, S_SOA, S_LASTA))


def iteration(e, first):
    """64 K values: 128 MFMAs, 32 fragment reads, 16 + 16 staging operations, two barriers"""
    advance_request(e)
    slots = {}
    for k in range(16):
        slots[k] = ("fr1", k)                                # K step 1's fragments, in the order its MFMAs want them
    slots[22] = ("bar", 0)                                   # every wave has read this K block: the buffer may be overwritten
    for s in range(NP):
        slots[24 + 2 * s] = ("st", s)
        slots[25 + 2 * s] = ("ld", s)
    slots[100] = ("bar", 0)                                  # the next K block is in LDS
    for k in range(16):
        slots[102 + k] = ("fr0", k)
    for n in range(2 * NI * NJ):
        mfma(e, n // (NI * NJ), n % (NI * NJ), first)
        kind, k = slots.get(n, ("none", 0))
        if kind == "fr1":
            frag_read(e, 1, *FRAG_ORDER[k])
        elif kind == "fr0":
            frag_read(e, 0, *FRAG_ORDER[k])
        elif kind == "st":
            store_piece(e, k)
        elif kind == "ld":
            load_piece(e, k)
        elif kind == "bar":
            e.barrier()


def gelu4(e, X, T, Q):
    """GELU of the 8 values v[X:X+7] in place (gelu_exact2's form (common.h) with the degree-5 exponent polynomial, four pairs side by side); T, Q: 8 scratch registers each"""
    for p in range(4):
        for h in range(2):
            e.raw("v_min_f32_e64 v%d, |v%d|, s%d" % (T + 2 * p + h, X + 2 * p + h, S_GC + 18))
    for p in range(4):          # q = t c5 + c4
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


def epilogue(e):
    """the finished tile: + bias, rounded to bf16 (the Linear's output under autocast), exact GELU, bf16, 16-byte stores.
    Entered with a full wait (the next tile's first fragments / blocks have landed under the K loop's tail); left with its
    last stores in flight -- they retire under the next tile's K loop."""
    e.drain()
    G.emit_mfma_result_guard(e.raw, "v_mfma_f32_16x16x32_bf16")    # the tile's last MFMAs -> v_accvgpr_read
    BIAS = E                                                 # 8 registers
    # (X, T, Q, O) x 2.  The O registers (what a store reads, some time after it issues) sit OUTSIDE the fragment area: the
    # next tile's K loop starts while this tile's last stores are still reading their data / on their way to memory
    sets = [(E + 8, E + 16, E + 24, VC + 2), (E + 32, E + 40, E + 48, VC + 6)]
    k = 0
    for jp in range(NJ // 2):
        e.vmem("buffer_load_dwordx4 v[%d:%d], %%[voB], s[%d:%d], 0 offen offset:%d" % (BIAS, BIAS + 3, RS_B, RS_B + 3, jp * 128), ("bias", 0))
        e.vmem("buffer_load_dwordx4 v[%d:%d], %%[voB], s[%d:%d], 0 offen offset:%d" % (BIAS + 4, BIAS + 7, RS_B, RS_B + 3, jp * 128 + 16), ("bias", 1))
        for i in range(NI):
            X, T, Q, O = sets[k & 1]
            e.need_vm({("out", k - 2)})                      # the store that read this register set has taken its data
            for h in range(2):
                c = acc(i, 2 * jp + h)
                for q in range(4):
                    e.raw("v_accvgpr_read_b32 v%d, a%d" % (X + 4 * h + q, c + q))
            e.need_vm({("bias", 0), ("bias", 1)})
            for p in range(4):
                e.raw("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (X + 2 * p, X + 2 * p + 1, X + 2 * p, X + 2 * p + 1, BIAS + 2 * p, BIAS + 2 * p + 1))
            for p in range(4):                               # the Linear's bf16 output, back as fp32
                e.raw("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (O + p, X + 2 * p, X + 2 * p + 1))
            for p in range(4):
                e.raw("v_lshlrev_b32 v%d, 16, v%d" % (X + 2 * p, O + p))
                e.raw("v_and_b32 v%d, 0xffff0000, v%d" % (X + 2 * p + 1, O + p))
            if "gelu" not in ABL:
                gelu4(e, X, T, Q)
            for p in range(4):
                e.raw("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (O + p, X + 2 * p, X + 2 * p + 1))
            e.vmem("buffer_store_dwordx4 v[%d:%d], %%[voC], s[%d:%d], s%d offen offset:%d" % (O, O + 3, RS_C, RS_C + 3, S_CR + i, jp * 64) + POLICY,
                   ("out", k))
            k += 1
    # the next tile's rows of C (scalar writes: the stores in flight have read their descriptor)
    e.raw("s_add_u32 s%d, s%d, %%[tstepC]" % (RS_C, RS_C))
    e.raw("s_addc_u32 s%d, s%d, 0" % (RS_C + 1, RS_C + 1))


def body():
    e = Emit()
    descriptor(e, RS_A, "%[alo]", "%[ahi]")
    descriptor(e, RS_W, "%[wlo]", "%[whi]")
    descriptor(e, RS_C, "%[clo]", "%[chi]")
    descriptor(e, RS_B, "%[blo]", "%[bhi]")
    e.raw("v_mov_b32 v%d, %%[voA]" % VO)
    for s in range(1, 8):
        e.raw("v_add_u32 v%d, %%[passA], v%d" % (VO + s, VO + s - 1))
    e.raw("v_mov_b32 v%d, %%[voW]" % (VO + 8))
    for s in range(9, 16):
        e.raw("v_add_u32 v%d, %%[passW], v%d" % (VO + s, VO + s - 1))
    e.raw("s_mov_b32 s%d, 0" % S_CR)
    for i in range(1, NI):
        e.raw("s_add_u32 s%d, s%d, %%[cstep]" % (S_CR + i, S_CR + i - 1))
    for c, v in enumerate(COEF):
        e.raw("s_mov_b32 s%d, %s" % (S_GC + 2 * c, f32(v)))
    e.raw("s_mov_b32 s%d, %s" % (S_GC + 18, f32(TMAX)))
    e.raw("v_mov_b32 v%d, s%d" % (VC, S_GC + 2))
    e.raw("v_mov_b32 v%d, s%d" % (VC + 1, S_GC + 2))
    # the last block a request may name: tile ntile - 1, K block KT - 1  (tstepA + 128 KT = the tile stride of A)
    e.raw("s_add_u32 s%d, %%[tstepA], %d" % (S_T, 128 * KT))
    e.raw("s_sub_u32 s%d, %%[ntile], 1" % S_LASTA)
    e.raw("s_mul_i32 s%d, s%d, s%d" % (S_LASTA, S_LASTA, S_T))
    e.raw("s_add_u32 s%d, s%d, %d" % (S_LASTA, S_LASTA, 128 * (KT - 1)))
    # ---- K block 0 -> registers -> LDS, K block 1 -> registers, fragments of K step 0
    e.raw("s_mov_b32 s%d, 0" % S_SOA)
    e.raw("s_mov_b32 s%d, 0" % S_SOW)
    e.raw("s_mov_b32 s%d, 0" % S_RQK)
    for s in range(NP):
        load_piece(e, s)
    for s in range(NP):
        store_piece(e, s)
    advance_request(e)
    for s in range(NP):
        load_piece(e, s)
    e.barrier()
    for which, idx in FRAG_ORDER:
        frag_read(e, 0, which, idx)
    e.raw("s_mov_b32 s%d, %%[ntile]" % S_TL)
    e.drain()                                                # (the tile loop is entered with nothing outstanding, as its epilogue leaves it)
    e.raw("2:")
    iteration(e, True)
    head = (list(e.lgkm), list(e.vm))
    e.raw("s_mov_b32 s%d, %d" % (S_KL, KT - 1))
    e.raw("1:")
    iteration(e, False)
    assert (e.lgkm, e.vm) == head, "loop body does not reproduce its head state"
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_KL, S_KL))
    e.raw("s_cmp_lg_u32 s%d, 0" % S_KL)
    e.raw("s_cbranch_scc1 1b")
    epilogue(e)
    e.raw("s_sub_u32 s%d, s%d, 1" % (S_TL, S_TL))
    e.raw("s_cmp_lg_u32 s%d, 0" % S_TL)
    e.raw("s_cbranch_scc1 2b")
    e.drain()
    return e.lines


def main():
    lines = body()
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % i for i in range(256)] + ['"v%d"' % i for i in range(218)] + \
           ['"s%d"' % i for i in range(40, 92)]
    with open(OUT, "w") as f:
        f.write("// generated by tools/gen_gemm_bf16_t4up.py -- do not edit\n")
        f.write("#define PIPS_T4UP_TEXT \\\n")
        for ln in lines:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n\n')
        f.write("#define PIPS_T4UP_CLOBBER " + ", ".join(clob) + "\n")
    print("PIPS_T4UP_TEXT: %d instructions, %d MFMAs" % (len(lines), sum("v_mfma" in ln for ln in lines)))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
