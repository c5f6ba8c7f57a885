#!/usr/bin/env python
"""Emit pips_amd/csrc/gather_item_asm.inc: the assembly text of ONE WORK ITEM (one map tile, <= 96 particles) of one
wave of gather_tiled_kernel -- window addresses, the 8 chunk phases (DMA issue, feature loads, LDS reads, FMAs) and
the 2x2 blend + store of the 196 taps.  The design lives in gather_tiled.hip (comment above gather_item()); this
script only unrolls it and does the register arithmetic.

    python tools/gen_gather_asm.py        # rewrites the .inc in place

Why assembly for the whole item: the accumulators (48 VGPRs) and window addresses (24) of a wave live across
8 barrier-separated phases.  Written in C++ around per-phase asm statements, hipcc spilled ~60 VGPRs around the
set-up and the blend (every reload a ~1 us scratch round trip) and those two parts cost as much as the phases.

Registers (everything from v23 / s16 up is private to the statement):
    v23          in-map bits of this lane's window pixel, bit 4*slot + level
    v[24:39]     fragment B (16 channels of this lane's pixel, odd slot of a pair) / temporaries
    v[40:87]     accumulators, unit u = level*6 + slot -> v[40+2u : 41+2u] (even / odd channel partial sums)
    v[88:111]    LDS byte address of this lane's window pixel of unit u (stage parity 0); staged rows are padded by 16 B
    v[112:127]   fragment A (even slot of a pair) / temporaries
    s[16:21]     byte offsets of the six slots' feature rows
    s[22:35]     counters, temporaries (not s32)
    s[36:67]     feature set 0: two slots x 16 channels;  s[68:99] feature set 1
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import asm_guards as G  # noqa: E402  (wait-state guards: the numbers live in tools/asm_hazard_lint.py)

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("PIPS_GEN_OUT", os.path.join(HERE, "..", "pips_amd", "csrc", "gather_item_asm.inc"))
TRACE = os.environ.get("PIPS_GEN_TRACE", "") == "1"   # tuning builds: per-stage clock counts in the lanes of %[tr]
ABL = os.environ.get("PIPS_GEN_ABLATE", "")        # tuning builds: any of reads,feats,fma,dma,warm,epi (wrong results)

LEVELS, SLOTS, Q, NCH = 4, 6, 4, 8
SZ = [34 * 1024, 19 * 1024, 11 * 1024, 8 * 1024]    # stage bytes per level (whole 1 KiB DMA pieces)
LB = [0, 2 * SZ[0], 2 * (SZ[0] + SZ[1]), 2 * (SZ[0] + SZ[1] + SZ[2])]
MISC = 2 * sum(SZ)                                   # scratch area behind the stages (L2-warm landing zone)
KIN_PAD, C, NCORR = 544, 128, 49

INM, FB, ACC, ADR, FA = 23, 24, 40, 88, 112
SET = [36, 68]
RO = 16                                              # s[16:21]
S_C, S_FOFF, S_DSOFF, S_NONEXT, S_PARMASK, S_WQ = 22, 23, 24, 25, 26, 27
S_ALLIN = 33                                         # bit 4k+l: unit's window wholly inside the map
S_EX = 28                                            # s[28:29] exec mask of the L2-warm lanes / saved exec
S_T = 30                                             # s[30:31] temporaries
S_U = 34                                             # s[34:35] temporaries   (s32 is left alone: hipcc reserves it)


class Asm:
    def __init__(self):
        self.lines, self.n = [], 0

    def __call__(self, s):
        self.lines.append(s)

    def label(self):
        self.n += 1
        return 100 + self.n


def probe(a, idx):
    """trace builds: lane idx of %[tr] += clocks since the previous probe (lane 0 = last time stamp)"""
    if not TRACE:
        return
    a("s_memtime s[%d:%d]" % (S_U, S_U + 1))
    a("s_waitcnt lgkmcnt(0)")
    a("v_readlane_b32 s%d, %%[tr], 0" % S_T)
    a("s_sub_u32 s%d, s%d, s%d" % (S_T + 1, S_U, S_T))
    a("v_readlane_b32 s%d, %%[tr], %d" % (S_T, idx))
    a("s_add_u32 s%d, s%d, s%d" % (S_T, S_T, S_T + 1))
    a("v_writelane_b32 %%[tr], s%d, %d" % (S_T, idx))
    a("v_writelane_b32 %%[tr], s%d, 0" % S_U)


def acc(l, k):
    return ACC + 2 * (l * SLOTS + k)


def adr(l, k):
    return ADR + l * SLOTS + k


def row_offsets(a):
    for k in range(SLOTS):
        a("v_readlane_b32 s%d, %%[row], %d" % (RO + k, 4 * k))
        a("s_lshl_b32 s%d, s%d, 9" % (RO + k, RO + k))             # * C * 4 bytes


def setup(a):
    """A[u], in-map bits, zeroed accumulators; the first stage's DMA pieces go out between the units.
    Temporaries: v[112:119], s[22:31], s[34:35]."""
    wi, wj, px, py, rx, ry, t, p = (FA + i for i in range(8))
    x0, y0, RW, RH, W, H, RWm, lb = 22, 23, 24, 25, 26, 27, 28, 29        # (s33 holds the inside-the-map bits)
    a("v_mbcnt_lo_u32_b32 v%d, -1, 0" % t)
    a("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (t, t))                 # lane id
    a("v_and_b32 v%d, 7, v%d" % (wi, t))
    a("v_lshrrev_b32 v%d, 3, v%d" % (wj, t))
    a("v_lshlrev_b32 v%d, 6, v%d" % (p, wi))                       # wi * 64: the lane's column offset inside a window row
    a("v_mov_b32 v%d, 0" % INM)
    a("s_mov_b32 s%d, 0" % S_ALLIN)                                # bit 4k+l: the unit's window lies wholly inside the map
    dma_piece(a, 0, first=True)
    pieces = 1
    for l in range(LEVELS):
        # wave-uniform geometry of level l out of the lane-parallel input %[gpk]: lanes l, 4+l, 8+l = x0|y0<<16, RW|RH<<16, W|H<<16
        a("v_readlane_b32 s%d, %%[gpk], %d" % (y0, l))
        a("s_and_b32 s%d, s%d, 0xffff" % (x0, y0))
        a("s_lshr_b32 s%d, s%d, 16" % (y0, y0))
        a("v_readlane_b32 s%d, %%[gpk], %d" % (RH, 4 + l))
        a("s_and_b32 s%d, s%d, 0xffff" % (RW, RH))
        a("s_lshr_b32 s%d, s%d, 16" % (RH, RH))
        a("v_readlane_b32 s%d, %%[gpk], %d" % (H, 8 + l))
        a("s_and_b32 s%d, s%d, 0xffff" % (W, H))
        a("s_lshr_b32 s%d, s%d, 16" % (H, H))
        a("s_sub_u32 s%d, s%d, 1" % (RWm, RW))
        a("s_lshl_b32 s%d, s%d, 6" % (RW, RW))
        a("s_add_u32 s%d, s%d, 16" % (RW, RW))                     # row pitch of the staged region: RW pixels of 64 B + 16 B pad
        a("s_add_u32 s%d, %%[ldsb], %d" % (lb, LB[l]))             # LDS address of the level's stage pair
        nextl = a.label()
        for k in range(SLOTS):
            bit = 4 * k + l
            if k > 0:                                              # slots past the wave's own particles: nothing to set up
                a("s_cmp_lt_u32 %d, %%[nslot]" % k)
                a("s_cbranch_scc0 %df" % nextl)
            slow, done = a.label(), a.label()
            a("v_readlane_b32 s%d, %%[bx], %d" % (S_U, bit))
            a("v_readlane_b32 s%d, %%[by], %d" % (S_U + 1, bit))
            # ---- window wholly inside the map (then also inside the staged region): no clamps, no per-lane mask
            a("s_sub_i32 s%d, s%d, 8" % (S_T, W))                  # 0 <= bx <= W - 8 (signed: W - 8 may be negative)
            a("s_cmp_le_i32 s%d, s%d" % (S_U, S_T))
            a("s_cbranch_scc0 %df" % slow)
            a("s_cmp_ge_i32 s%d, 0" % S_U)
            a("s_cbranch_scc0 %df" % slow)
            a("s_sub_i32 s%d, s%d, 8" % (S_T, H))
            a("s_cmp_le_i32 s%d, s%d" % (S_U + 1, S_T))
            a("s_cbranch_scc0 %df" % slow)
            a("s_cmp_ge_i32 s%d, 0" % (S_U + 1))
            a("s_cbranch_scc0 %df" % slow)
            a("s_bitset1_b32 s%d, %d" % (S_ALLIN, bit))
            a("s_sub_u32 s%d, s%d, s%d" % (S_T, S_U, x0))          # dx = bx - x0
            a("s_sub_u32 s%d, s%d, s%d" % (S_T + 1, S_U + 1, y0))  # dy = by - y0
            a("s_lshl_b32 s%d, s%d, 6" % (S_T, S_T))
            a("s_add_u32 s%d, s%d, s%d" % (S_T, S_T, lb))          # lb + 64 dx
            a("v_add_u32 v%d, s%d, v%d" % (ry, S_T + 1, wj))
            a("v_mad_u32_u24 v%d, v%d, s%d, v%d" % (adr(l, k), ry, RW, p))     # ry * pitch + wi * 64
            a("v_add_u32 v%d, s%d, v%d" % (adr(l, k), S_T, adr(l, k)))
            a("s_branch %df" % done)
            a("%d:" % slow)
            a("v_add_u32 v%d, s%d, v%d" % (px, S_U, wi))
            a("v_add_u32 v%d, s%d, v%d" % (py, S_U + 1, wj))
            a("v_cmp_gt_u32 vcc, s%d, v%d" % (W, px))              # px < W (unsigned: also px >= 0)
            G.emit_sgpr_to_valu_guard(a)                           # VCC from a VALU compare -> v_cndmask: two wait states (gfx940+)
            a("v_cndmask_b32 v%d, 0, 1, vcc" % t)
            a("v_cmp_gt_u32 vcc, s%d, v%d" % (H, py))
            G.emit_sgpr_to_valu_guard(a)
            a("v_cndmask_b32 v%d, 0, v%d, vcc" % (t, t))
            a("v_lshl_or_b32 v%d, v%d, %d, v%d" % (INM, t, bit, INM))
            a("v_subrev_u32 v%d, s%d, v%d" % (rx, x0, px))         # px - x0, clamped into the staged region
            a("v_med3_i32 v%d, v%d, 0, s%d" % (rx, rx, RWm))
            a("v_subrev_u32 v%d, s%d, v%d" % (ry, y0, py))
            a("s_sub_u32 s%d, s%d, 1" % (S_T, RH))
            a("v_med3_i32 v%d, v%d, 0, s%d" % (ry, ry, S_T))
            a("v_lshlrev_b32 v%d, 6, v%d" % (rx, rx))
            a("v_mad_u32_u24 v%d, v%d, s%d, v%d" % (adr(l, k), ry, RW, rx))     # ry * pitch + rx * 64
            a("v_add_u32 v%d, s%d, v%d" % (adr(l, k), lb, adr(l, k)))
            a("%d:" % done)
            if k == 0:                                             # (units every wave runs, whatever its particle count)
                dma_piece(a, pieces, first=True)                   # the first stage (chunk 0 -> parity 0) goes out between the units
                pieces += 1
        a("%d:" % nextl)
    while pieces < 5:
        dma_piece(a, pieces, first=True)
        pieces += 1
    for r in range(ACC, ACC + 2 * LEVELS * SLOTS):
        a("v_mov_b32 v%d, 0" % r)


def wave_quarter(a):
    a("s_lshr_b32 s%d, %%[wave], 2" % S_WQ)


def warm_mask(a):
    """s[S_EX:S_EX+1] = exec mask of the lanes that touch feature lines: lanes < clamp(count - 64 wave, 0, 64)"""
    a("s_lshl_b32 s%d, %%[wave], 6" % S_T)
    a("s_sub_i32 s%d, %%[count], s%d" % (S_T, S_T))
    a("s_max_i32 s%d, s%d, 0" % (S_T, S_T))
    a("s_min_i32 s%d, s%d, 64" % (S_T, S_T))
    a("s_bfm_b64 s[%d:%d], s%d, 0" % (S_EX, S_EX + 1, S_T))                 # (a count of 64 gives 0)
    a("s_cmp_eq_u32 s%d, 64" % S_T)
    a("s_cselect_b64 s[%d:%d], -1, s[%d:%d]" % (S_EX, S_EX + 1, S_EX, S_EX + 1))


def dma_piece(a, r, first=False):
    """Piece r of the next chunk's stage (first: of chunk 0 into parity 0, during the set-up): global -> LDS by the DMA engine.  LDS offset = %[gpk] lane 16+r (+ the
    level's stage size, lane 24+r, when the target parity mask s26 is set) -- negative: this wave has no such piece."""
    if "dma" in ABL:
        return
    skip = a.label()
    a("v_readlane_b32 s%d, %%[gpk], %d" % (S_T, 16 + r))
    a("s_cmp_lt_i32 s%d, 0" % S_T)
    a("s_cbranch_scc1 %df" % skip)
    if not first:
        a("v_readlane_b32 s%d, %%[gpk], %d" % (S_T + 1, 24 + r))
        a("s_and_b32 s%d, s%d, s%d" % (S_T + 1, S_T + 1, S_PARMASK))
        a("s_add_u32 s%d, s%d, s%d" % (S_T, S_T, S_T + 1))
    a("s_add_u32 m0, s%d, %%[ldsb]" % S_T)
    a("s_nop 0")
    a("buffer_load_dwordx4 %%[doff%d], %%[rsrc], %s offen lds" % (r, "0" if first else "s%d" % S_DSOFF))
    a("%d:" % skip)


def feat_request(a, dst, k, off_reg):
    """s_load the 16-channel chunk of slots k, k+1 into set dst (rows past the wave's particles repeat a valid one)."""
    if "feats" in ABL:
        return
    for i in range(2):
        a("s_add_u32 s%d, s%d, s%d" % (S_U + i, RO + k + i, off_reg))
    for i in range(2):
        a("s_load_dwordx16 s[%d:%d], %%[fb], s%d" % (SET[dst] + 16 * i, SET[dst] + 16 * i + 15, S_U + i))


def reads(a, l, k, frag, par):
    if "reads" in ABL:
        return
    ad = adr(l, k)
    for i in range(Q):                               # the four channel quads of the pixel: consecutive 16 B (rows are padded, not swizzled)
        a("ds_read_b128 v[%d:%d], v%d offset:%d" % (frag + 4 * i, frag + 4 * i + 3, ad, par * SZ[l] + 16 * i))


def fmas(a, l, k, s, frag):
    for i in range(1 if "fma" in ABL else 2 * Q):
        a("v_pk_fma_f32 v[%d:%d], s[%d:%d], v[%d:%d], v[%d:%d]" % (acc(l, k), acc(l, k) + 1, s + 2 * i, s + 2 * i + 1,
                                                                    frag + 2 * i, frag + 2 * i + 1, acc(l, k), acc(l, k) + 1))


def phase(a, NP, par):
    """Chunk c (loop counter s22, c & 1 == par) of a wave with NP slot pairs (the last one possibly half full)."""
    a("s_waitcnt vmcnt(0)")                          # this wave's pieces of chunk c landed (its earlier stores left)
    probe(a, 2)
    a("s_barrier")                                   # everyone's did; everyone is done with the other parity
    probe(a, 3)
    a("s_lshl_b32 s%d, s%d, 6" % (S_FOFF, S_C))      # byte offset of chunk c in a feature row / map pixel
    a("s_add_u32 s%d, s%d, 64" % (S_DSOFF, S_FOFF))  # ... of chunk c + 1 (staged now)
    a("s_cmp_lt_u32 s%d, %d" % (S_C, NCH - 1))       # is there a chunk c + 1 ?
    a("s_cselect_b32 s%d, 0, -1" % S_NONEXT)
    a("s_mov_b32 s%d, %d" % (S_PARMASK, 0 if par else -1))   # it goes to the other parity
    if par == 0 and "warm" not in ABL:
        # keep the item's feature lines in the L2: thread p < count touches particle p's line of chunk c + 2
        nowarm = a.label()
        a("s_cmp_lt_u32 s%d, %d" % (S_C, NCH - 2))
        a("s_cbranch_scc0 %df" % nowarm)
        a("s_mov_b64 s[%d:%d], exec" % (S_T, S_T + 1))
        a("s_mov_b64 exec, s[%d:%d]" % (S_EX, S_EX + 1))
        a("s_add_u32 s%d, s%d, 128" % (S_U, S_FOFF))      # (not as an instruction offset: that moves the LDS side too)
        a("v_add_u32 v%d, s%d, %%[warm]" % (FB, S_U))
        a("s_add_u32 m0, %%[ldsb], %d" % MISC)
        a("s_nop 0")
        a("global_load_lds_dword v%d, %%[fb]" % FB)
        a("s_mov_b64 exec, s[%d:%d]" % (S_T, S_T + 1))
        a("%d:" % nowarm)

    def dma_slot(g):
        """Group boundary g of the phase: the wave's piece r = g - (wave >> 2) goes out here (if there is a next chunk):
        the 72 pieces of a stage enter the texture-address queue spread over ~8 FMA groups instead of all at once"""
        lab = a.label()
        a("s_cmp_eq_u32 s%d, -1" % S_NONEXT)
        a("s_cbranch_scc1 %df" % lab)
        for r in range(5):
            if 0 <= g - r <= 3:
                nxt = a.label()
                a("s_cmp_eq_u32 s%d, %d" % (S_WQ, g - r))
                a("s_cbranch_scc0 %df" % nxt)
                dma_piece(a, r)
                a("s_branch %df" % lab)
                a("%d:" % nxt)
        a("%d:" % lab)

    def dma_rest(gfrom):
        """a wave with fewer groups than boundaries issues what is left at its last one"""
        for g in range(gfrom, 8):
            dma_slot(g)

    dma_slot(0)
    for j in range(NP):
        ka, kb = 2 * j, 2 * j + 1
        s = SET[(par * NP + j) & 1]
        last = j == NP - 1
        for l in range(LEVELS):
            # ---- LDS reads of the pair's two units at this level
            reads(a, l, ka, FA, par)
            nob = a.label()
            if last:                                  # the wave's last pair may hold one slot only
                a("s_cmp_lt_u32 %d, %%[nslot]" % kb)
                a("s_cbranch_scc0 %df" % nob)
            a("s_bitcmp1_b32 %%[same], %d" % (4 * kb + l))   # same window anchor as slot a: its fragment serves both
            a("s_cbranch_scc1 %df" % nob)
            reads(a, l, kb, FB, par)
            a("%d:" % nob)
            a("s_waitcnt lgkmcnt(0)")
            if l == 0:
                # features of the NEXT pair (this chunk), or of pair 0 of the next chunk, into the other set
                if not last:
                    feat_request(a, (par * NP + j + 1) & 1, 2 * (j + 1), S_FOFF)
                else:
                    nof = a.label()
                    a("s_cmp_eq_u32 s%d, -1" % S_NONEXT)
                    a("s_cbranch_scc1 %df" % nof)
                    feat_request(a, ((1 - par) * NP) & 1, 0, S_DSOFF)
                    a("%d:" % nof)
            g = 4 * j + l + 1
            if g < 8:
                dma_slot(g)
            if NP == 1 and l == LEVELS - 1:
                dma_rest(5)
            fmas(a, l, ka, s, FA)
            done = a.label()
            if last:
                a("s_cmp_lt_u32 %d, %%[nslot]" % kb)
                a("s_cbranch_scc0 %df" % done)
            usea = a.label()
            a("s_bitcmp1_b32 %%[same], %d" % (4 * kb + l))
            a("s_cbranch_scc1 %df" % usea)
            fmas(a, l, kb, s + 16, FB)
            a("s_branch %df" % done)
            a("%d:" % usea)
            fmas(a, l, kb, s + 16, FA)
            a("%d:" % done)
        probe(a, 4 + j)


def phases(a):
    """The 8 phases, one code variant per number of slot pairs of the wave (1..3); a wave without particles only stages."""
    end = a.label()
    var = {n: a.label() for n in (0, 1, 2)}
    a("s_mov_b32 s%d, 0" % S_C)
    a("s_cmp_eq_u32 %[nslot], 0")
    a("s_cbranch_scc1 %df" % var[0])
    a("s_cmp_le_u32 %[nslot], 2")
    a("s_cbranch_scc1 %df" % var[1])
    a("s_cmp_le_u32 %[nslot], 4")
    a("s_cbranch_scc1 %df" % var[2])
    for NP in (3, 2, 1, 0):
        if NP < 3:
            a("%d:" % var[NP])
        top = a.label()
        a("%d:" % top)
        if NP == 0:
            a("s_waitcnt vmcnt(0)")
            a("s_barrier")
            nxt = a.label()
            a("s_cmp_lt_u32 s%d, %d" % (S_C, NCH - 1))
            a("s_cbranch_scc0 %df" % nxt)
            a("s_lshl_b32 s%d, s%d, 6" % (S_DSOFF, S_C))
            a("s_add_u32 s%d, s%d, 64" % (S_DSOFF, S_DSOFF))
            a("s_and_b32 s%d, s%d, 1" % (S_PARMASK, S_C))
            a("s_sub_u32 s%d, s%d, 1" % (S_PARMASK, S_PARMASK))       # chunk c even -> target parity 1 -> mask -1
            for r in range(5):
                dma_piece(a, r)
            a("%d:" % nxt)
            a("s_add_u32 s%d, s%d, 1" % (S_C, S_C))
            a("s_cmp_lt_u32 s%d, %d" % (S_C, NCH))
            a("s_cbranch_scc1 %db" % top)
        else:
            phase(a, NP, 0)
            a("s_add_u32 s%d, s%d, 1" % (S_C, S_C))
            phase(a, NP, 1)
            a("s_add_u32 s%d, s%d, 1" % (S_C, S_C))
            a("s_cmp_lt_u32 s%d, %d" % (S_C, NCH))
            a("s_cbranch_scc1 %db" % top)
            a("s_branch %df" % end)
    a("%d:" % end)


def epilogue(a):
    """2x2 blend of the 8x8 correlations to the 49 taps (k = ix*7 + iy: transposed, nets/pips.py:379-381) and the
    store of X[row][128 + 49 l + k].  Temporaries: v[24:35] (d values, weights, addresses), v[112:127] (neighbours)."""
    if "epi" in ABL:
        return
    d = [FB + i for i in range(4)]
    w = [FB + 4 + i for i in range(4)]           # lane-parallel blend weights (lane 4k+l = unit)
    bp, l4, t, o = FB + 8, FB + 9, FB + 10, FB + 11
    a("v_mbcnt_lo_u32_b32 v%d, -1, 0" % t)
    a("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (t, t))
    a("v_lshlrev_b32 v%d, 2, v%d" % (l4, t))                                  # lane * 4: store offset
    a("v_mul_u32_u24 v%d, 37, v%d" % (o, t))
    a("v_lshrrev_b32 v%d, 8, v%d" % (o, o))                                   # ti = t / 7 (t < 64)
    a("v_mad_i32_i24 v%d, v%d, -7, v%d" % (bp, o, t))                         # tj = t - 7 ti
    a("v_lshl_add_u32 v%d, v%d, 3, v%d" % (bp, bp, o))                        # source lane tj*8 + ti
    a("v_min_u32 v%d, 54, v%d" % (bp, bp))                                    # (lanes >= 49 are not stored)
    a("v_lshlrev_b32 v%d, 2, v%d" % (bp, bp))
    a("v_sub_f32 v%d, 1.0, %%[wx]" % t)                                       # e = 1 - wx
    a("v_sub_f32 v%d, 1.0, %%[wy]" % o)                                       # so = 1 - wy
    a("v_mul_f32 v%d, v%d, v%d" % (w[0], o, t))                               # nw: so * e
    a("v_mul_f32 v%d, v%d, %%[wx]" % (w[1], o))                               # ne: so * wx
    a("v_mul_f32 v%d, %%[wy], v%d" % (w[2], t))                               # sw: wy * e
    a("v_mul_f32 v%d, %%[wy], %%[wx]" % w[3])                                 # se: wy * wx
    for i in range(4):
        a("v_mul_f32 v%d, 0x3db504f3, v%d" % (w[i], w[i]))                    # the 1/sqrt(128) of :397 rides on the weights
    a("s_mov_b64 s[%d:%d], exec" % (S_EX, S_EX + 1))
    end = a.label()
    for k in range(SLOTS):
        a("s_cmp_lt_u32 %d, %%[nslot]" % k)
        a("s_cbranch_scc0 %df" % end)
        for l in range(LEVELS):
            a("v_add_f32 v%d, v%d, v%d" % (d[l], acc(l, k), acc(l, k) + 1))
            inside = a.label()
            a("s_bitcmp1_b32 s%d, %d" % (S_ALLIN, 4 * k + l))
            a("s_cbranch_scc1 %df" % inside)
            a("v_and_b32 v%d, %s, v%d" % (t, hex(1 << (4 * k + l)), INM))
            a("v_cmp_ne_u32 vcc, 0, v%d" % t)
            G.emit_sgpr_to_valu_guard(a)
            a("v_cndmask_b32 v%d, 0, v%d, vcc" % (d[l], d[l]))                 # zeros padding outside the map
            a("%d:" % inside)
            for i, off in enumerate((0, 4, 32, 36)):                          # nw, ne, sw, se
                a("ds_bpermute_b32 v%d, v%d, v%d offset:%d" % (FA + 4 * l + i, bp, d[l], off))
        a("v_readlane_b32 s%d, %%[row], %d" % (S_T, 4 * k))
        a("s_mul_i32 s%d, s%d, %d" % (S_T, S_T, KIN_PAD * 4))
        a("s_waitcnt lgkmcnt(0)")
        # the 16 blend weights of the slot (4 levels x nw / ne / sw / se) go to SGPRs FIRST -- the feature sets are free in the
        # epilogue -- and are used afterwards: an SGPR written by v_readlane needs two wait states before a VALU instruction
        # reads it (gfx940+; rounds 2-5 used one SGPR per weight, read by the very next instruction)
        for l in range(LEVELS):
            for i in range(4):
                a("v_readlane_b32 s%d, v%d, %d" % (SET[0] + 4 * l + i, w[i], 4 * k + l))
        for l in range(LEVELS):
            for i in range(4):
                if i == 0:
                    a("v_mul_f32 v%d, s%d, v%d" % (o + l, SET[0] + 4 * l, FA + 4 * l))
                else:
                    a("v_fmac_f32 v%d, s%d, v%d" % (o + l, SET[0] + 4 * l + i, FA + 4 * l + i))
        a("v_add_u32 v%d, s%d, v%d" % (t, S_T, l4))
        a("s_bfm_b64 exec, 49, 0")                                               # lanes 0..48
        for l in range(LEVELS):
            a("global_store_dword v%d, v%d, %%[xp] offset:%d" % (t, o + l, (C + NCORR * l) * 4))
        a("s_mov_b64 exec, s[%d:%d]" % (S_EX, S_EX + 1))
    a("%d:" % end)


def main():
    a = Asm()
    row_offsets(a)
    a("s_mov_b32 s%d, 0" % S_FOFF)
    feat_request(a, 0, 0, S_FOFF)                    # pair 0 of chunk 0: lands under the address set-up
    if TRACE:
        a("s_memtime s[%d:%d]" % (S_U, S_U + 1))
        a("s_waitcnt lgkmcnt(0)")
        a("v_writelane_b32 %%[tr], s%d, 0" % S_U)
    setup(a)
    warm_mask(a)
    wave_quarter(a)
    probe(a, 1)
    phases(a)
    probe(a, 7)
    epilogue(a)
    probe(a, 8)
    lines = ["// GENERATED by tools/gen_gather_asm.py -- do not edit; the design is documented in gather_tiled.hip.", "",
             "#define PIPS_ITEM_TEXT \\"]
    for i, ins in enumerate(a.lines):
        lines.append('    "%s\\n\\t"' % ins + (" \\" if i + 1 < len(a.lines) else ""))
    lines.append("")
    clob = ['"s%d"' % i for i in range(RO, 100) if i != 32] + ['"v%d"' % i for i in range(INM, 128)]
    lines.append('#define PIPS_ITEM_CLOBBER "memory", "scc", "vcc", ' + ", ".join(clob))
    lines.append("")
    with open(OUT, "w") as f:
        f.write("\n".join(lines))
    print("wrote", os.path.normpath(OUT), len(a.lines), "instructions")


if __name__ == "__main__":
    main()
