"""A software-wait-state lint for every instruction stream of libpips_hip.so (gfx950).

Why: hipcc pads the hazards of its OWN instructions, but an `asm volatile` statement is opaque to its hazard recognizer --
and 37 k lines of the product's hot loops are generated assembly (csrc/*_asm.inc) plus the hand-written statements of
gather_tiled.hip.  Round 5 found an MFMA -> DS-read hazard in such a statement by accident: it had passed every
test bit for bit until a register-allocation change moved two instructions.  Tests cannot see a hazard that today's schedule
happens to cover; this lint reads the DISASSEMBLY of the shipped code object and checks the distance between every producer and
consumer of the pairs below, compiler code and assembly statements alike.

What it does: `llvm-objdump --offloading` + `-d` on the shared library -> per kernel the instruction list and its control-flow
graph (branch targets are printed by objdump) -> a forward data-flow over "events" (register written by a producer of some
class, wait states elapsed since, minimum over all paths) -> at every instruction the events of the registers it touches are
compared with the wait states the pair needs.  One wait state = one issued instruction of the wave, `s_nop N` = N + 1 (the
counting of LLVM's GCNHazardRecognizer).

Where the numbers come from: gfx950 has no public hazard table in this image, so the table was PINNED AGAINST THE COMPILER'S OWN
RECOGNIZER: tools/asm_hazard_probe.hip holds one small kernel per pair, written with builtins and `sched_barrier`s so that
producer and consumer are adjacent; the `s_nop`s hipcc inserts between them are the requirement
(tests/test_boundary.py::test_hazard_table_matches_the_compilers_recognizer recompiles the probes and compares).  Pairs the
probes cannot express (write-after-write on an MFMA result) take the read number, the conservative choice of the LLVM tables
(AMDGPU/GCNHazardRecognizer.cpp, gfx940 rows).

    MFMA result (vDst)  ->  any other instruction reading OR writing it, incl. an MFMA taking it as A / B:
          v_mfma_f32_32x32x16_bf16 (8 passes) 12   v_mfma_f32_16x16x32_bf16 (4 passes) 8   v_mfma_f32_32x32x2_f32 (16 passes) 18
          v_mfma_f32_16x16x4_f32 (8 passes) 10
    MFMA result         ->  MFMA taking exactly the same registers as C: 0;  overlapping but different C: passes + 2
    VALU writes a VGPR  ->  MFMA reads it (A / B / C) 2;  DPP reads it 2;  v_readlane / v_readfirstlane reads it 1;
                            v_permlane*_swap reads it 2
    transcendental VALU (v_exp / log / rcp / rsq / sqrt / sin / cos) -> non-transcendental VALU reads the result 1
    VALU writes an SGPR / VCC (v_readlane, v_readfirstlane, v_cmp, carry-out)
                        ->  vector-memory instruction reads it 5;  lane select of v_readlane / v_writelane 4;  v_div_fmas (VCC) 4;
                            VALU reads it as an operand 2
    VALU writes EXEC (v_cmpx) -> DPP 5
    SALU writes M0      ->  LDS-DMA (`... lds`), ds_*_addtid, s_movrel 1
    vector-memory store of more than 8 bytes -> VALU overwrites its data registers 2

usage: python tools/asm_hazard_lint.py [libpips_hip.so | file.o | file.s ...] [-v]      exit code 1 when a hazard is found
       (a .s argument is a saved `llvm-objdump -d` text: tests/golden/hazard_negative_*.s)
       python tools/asm_hazard_lint.py --probes                                         the table against hipcc's recognizer
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("PIPS_LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
MAXW = 24                                  # events older than this many wait states are dropped (largest requirement: 18)

# ---- the table (see the docstring) -------------------------------------------------------------------------------
MFMA_PASSES = {"v_mfma_f32_32x32x16_bf16": 8, "v_mfma_f32_16x16x32_bf16": 4, "v_mfma_f32_32x32x2_f32": 16,
               "v_mfma_f32_16x16x4_f32": 8, "v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_16x16x32_f16": 4}
MFMA_F32_INPUT = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x4_f32"}        # "SMFMA" rows: passes + 2; the bf16 / f16 double-K forms: passes + 4


def mfma_read_ws(op):
    p = MFMA_PASSES[op]
    return p + 2 if op in MFMA_F32_INPUT else p + 4


# An MFMA issued between an MFMA and the consumer of its result: LLVM's recognizer counts it as ONE wait state inside a basic
# block; physically the matrix pipe is in order and takes one MFMA per `passes` quad-cycles, so by the time the later MFMA has
# issued the earlier one has spent at least its own passes.  Counted as the shortest pass count in use (4): hipcc's own code
# relies on it across basic blocks (gemm_x3_kernel reads an accumulator 9 instructions, 3 of them MFMAs, behind its MFMA).
MFMA_BETWEEN = 4


def mfma_srcc_overlap_ws(op):
    return MFMA_PASSES[op] + 2


WS_VALU_VGPR_TO_MFMA = 2
WS_VALU_VGPR_TO_DPP = 2
WS_VALU_VGPR_TO_READLANE = 1
WS_VALU_VGPR_TO_PERMLANE_SWAP = 2
WS_TRANS_TO_VALU = 1
WS_VALU_SGPR_TO_VMEM = 5
WS_VALU_SGPR_TO_LANESEL = 4
WS_VALU_VCC_TO_DIV_FMAS = 4
WS_VALU_SGPR_TO_VALU = 2
WS_VALU_EXEC_TO_DPP = 5
WS_SALU_M0_TO_LDSDMA = 1
WS_STORE_DATA_TO_VALU_WRITE = 2

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
ACCUM_DST = ("v_fmac_", "v_mac_", "v_pk_fmac_", "v_dot2c_", "v_dot4c_", "v_dot8c_", "v_writelane_b32", "v_movreld", "v_fmamk_", "v_madmk_")
CARRY_DEF2 = ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32",
              "v_div_scale_f32", "v_div_scale_f64", "v_mad_u64_u32", "v_mad_i64_i32")
DPP_MODS = ("quad_perm:", "row_shl:", "row_shr:", "row_ror:", "wave_shl:", "wave_shr:", "wave_rol:", "wave_ror:", "row_bcast:", "row_mirror",
            "row_half_mirror", "row_newbcast:")
VMEM_PREFIX = ("buffer_", "global_", "flat_", "scratch_", "tbuffer_")


# ---- parsing -------------------------------------------------------------------------------------------------------
class Ins:
    __slots__ = ("addr", "mnem", "ops", "mods", "text", "target", "regs", "defs", "uses")

    def __init__(self, addr, text, target):
        self.addr, self.text, self.target = addr, text, target
        parts = text.split(None, 1)
        self.mnem = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        pieces, depth, cur = [], 0, ""
        for ch in rest:
            if ch in "[(":
                depth += 1
            elif ch in "])":
                depth -= 1
            if ch == "," and depth == 0:
                pieces.append(cur.strip()); cur = ""
            else:
                cur += ch
        if cur.strip():
            pieces.append(cur.strip())
        self.ops, mods = [], []
        for i, p in enumerate(pieces):
            toks = p.split()
            if not toks:
                continue
            if i == len(pieces) - 1 and len(toks) > 1:
                # operand followed by modifiers (offset:16 lds, quad_perm:[..] row_mask:0xf ...)
                self.ops.append(toks[0]); mods += toks[1:]
            elif ":" in toks[0] and not toks[0].startswith(("v[", "s[", "a[", "ttmp[")) and "[" in toks[0] and i > 0 and _regs(toks[0]) == []:
                mods += toks                                     # a modifier with a bracketed list that contained commas
            else:
                self.ops.append(toks[0]); mods += toks[1:]
        self.mods = " ".join(mods)
        self.regs = [_regs(o) for o in self.ops]
        self.defs, self.uses = _def_use(self)


_REG1 = re.compile(r"^(v|a|s|ttmp)(\d+)$")
_REGN = re.compile(r"^(v|a|s|ttmp)\[(\d+):(\d+)\]$")


def _regs(tok):
    t = tok.strip()
    for w in ("neg(", "abs(", "sext("):
        if t.startswith(w) and t.endswith(")"):
            t = t[len(w):-1]
    t = t.lstrip("-").strip("|")
    m = _REG1.match(t)
    if m:
        return [(m.group(1), int(m.group(2)))]
    m = _REGN.match(t)
    if m:
        return [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if t in ("vcc", "vcc_lo", "vcc_hi"):
        return [("vcc", 0)]
    if t in ("exec", "exec_lo", "exec_hi"):
        return [("exec", 0)]
    if t == "m0":
        return [("m0", 0)]
    return []


def _flat(lists):
    return [r for l in lists for r in l]


def _def_use(I):
    """(registers written, registers read) of an instruction, as far as the table's rules need them."""
    m, R = I.mnem, I.regs
    if m.startswith("v_mfma") or m.startswith("v_smfmac"):
        return R[0] if R else [], _flat(R[1:])
    if m.startswith("v_cmpx"):
        d = [("exec", 0)] + (R[0] if R and R[0] and R[0][0][0] in ("s", "vcc") else [])
        return d, _flat(R[1:] if len(d) > 1 else R)
    if m.startswith("v_"):
        if m.startswith(("v_swap_b32", "v_permlane16_swap", "v_permlane32_swap")):
            return _flat(R[:2]), _flat(R[:2])
        nd = 2 if m.startswith(CARRY_DEF2) else 1
        d = _flat(R[:nd])
        u = _flat(R[nd:])
        if m.startswith(ACCUM_DST) or any(k in I.mods for k in DPP_MODS):
            u = u + (R[0] if R else [])                         # accumulate forms and DPP's `old` read the destination
        if m.startswith("v_div_fmas"):
            u = u + [("vcc", 0)]
        if m.startswith("v_nop"):
            return [], []
        return d, u
    if m.startswith("ds_"):
        if m.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append", "ds_ordered")) or "_rtn" in m:
            return (R[0] if R else []), _flat(R[1:])
        return [], _flat(R)
    if m.startswith(VMEM_PREFIX):
        lds = " lds" in (" " + I.mods) or "_lds_" in m
        if ("_load" in m) and not lds:
            return (R[0] if R else []), _flat(R[1:])
        u = _flat(R)
        if lds:
            u = u + [("m0", 0)]
        if "atomic" in m and ("sc0" in I.mods or "glc" in I.mods):
            return (R[0] if R else []), u
        return [], u
    if m.startswith("s_"):
        if m.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_branch", "s_cbranch", "s_endpgm", "s_sleep", "s_setprio", "s_sendmsg", "s_icache",
                         "s_dcache", "s_trap", "s_sethalt", "s_ttrace", "s_code_end", "s_setreg", "s_decperflevel", "s_incperflevel")):
            return [], _flat(R)
        if m.startswith(("s_cmp", "s_bitcmp")):
            return [], _flat(R)
        if m.startswith(("s_store", "s_buffer_store", "s_scratch_store")):
            return [], _flat(R)
        u = _flat(R[1:])
        if m.startswith(("s_movrel", )):
            u = u + [("m0", 0)]
        return (R[0] if R else []), u
    return [], _flat(R)


_LINE = re.compile(r"^\t(\S.*?)\s*// ([0-9A-Fa-f]+): [0-9A-Fa-f ]+(?:<([^>]+)>)?\s*$")
_SYM = re.compile(r"^([0-9a-f]+) <(\S+)>:$")


def parse_objdump(text):
    """-> {symbol: [Ins]}"""
    out, cur, base = {}, None, {}
    for line in text.splitlines():
        m = _SYM.match(line)
        if m:
            cur = out.setdefault(m.group(2), [])
            base[m.group(2)] = int(m.group(1), 16)
            continue
        if cur is None:
            continue
        m = _LINE.match(line)
        if not m:
            continue
        body, addr, tgt = m.group(1), int(m.group(2), 16), m.group(3)
        target = None
        if tgt and body.startswith(("s_branch", "s_cbranch")):
            mm = re.match(r"^(\S+?)(?:\+0x([0-9a-f]+))?$", tgt)
            if mm and mm.group(1) in base:
                target = base[mm.group(1)] + (int(mm.group(2), 16) if mm.group(2) else 0)
        cur.append(Ins(addr, body, target))
    return out


def disassemble(path):
    """Every gfx950 code object bundled in a host shared library / object -> list of objdump texts."""
    if path.endswith(".s"):
        return [open(path).read()]
    tmp = tempfile.mkdtemp(prefix="pips_lint_")
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)                                  # --offloading writes the bundles beside its input
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        texts = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f and "gfx950" in f:
                texts.append(subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout)
        if not texts:                                             # a bare device ELF
            texts.append(subprocess.run([OBJDUMP, "-d", local], check=True, capture_output=True, text=True).stdout)
        return texts
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---- the data-flow ---------------------------------------------------------------------------------------------------
def wait_states(I):
    if I.mnem == "s_nop":
        try:
            return int(I.ops[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def is_valu(I):
    return I.mnem.startswith("v_") and not I.mnem.startswith(("v_mfma", "v_smfmac", "v_nop"))


def is_mfma(I):
    return I.mnem.startswith(("v_mfma", "v_smfmac"))


def is_trans(I):
    return I.mnem.startswith(TRANS)


def is_dpp(I):
    return any(k in I.mods for k in DPP_MODS) or I.mnem.endswith("_dpp")


def is_vmem(I):
    return I.mnem.startswith(VMEM_PREFIX)


def is_lds_dma(I):
    return is_vmem(I) and (" lds" in (" " + I.mods) or "_lds_" in I.mnem)


def events_of(I, idx):
    """Events an instruction raises: {(kind, reg): payload}."""
    ev = {}
    if is_mfma(I):
        if I.mnem not in MFMA_PASSES:
            raise ValueError(f"asm_hazard_lint: no pass count for {I.mnem}; add it to MFMA_PASSES (probe it first)")
        rng = tuple(I.defs)
        for r in I.defs:
            ev[("mfma", r)] = (idx, I.mnem, rng)
        return ev
    if is_valu(I):
        for r in I.defs:
            if r[0] in ("v", "a"):
                ev[("valu_vgpr", r)] = (idx,)
                if is_trans(I):
                    ev[("trans", r)] = (idx,)
            elif r[0] in ("s", "vcc", "ttmp"):
                ev[("valu_sgpr", r)] = (idx,)
            elif r[0] == "exec":
                ev[("valu_exec", r)] = (idx,)
    elif I.mnem.startswith("s_") and ("m0", 0) in I.defs:
        ev[("salu_m0", ("m0", 0))] = (idx,)
    elif is_vmem(I) and "store" in I.mnem and I.mnem.endswith(("x3", "x4")):
        data = I.regs[1] if I.mnem.startswith(("global_", "flat_", "scratch_")) and len(I.regs) > 1 else (I.regs[0] if I.regs else [])
        for r in data:
            ev[("store_data", r)] = (idx,)
    return ev


def requirements(I, state):
    """Yield (need, elapsed, producer index, rule) for every event the instruction must be some distance away from."""
    if I.mnem in ("s_nop", "s_waitcnt", "s_barrier"):
        return
    touched = set(I.defs) | set(I.uses)
    uses, defs = set(I.uses), set(I.defs)
    # -- MFMA results
    for r in touched:
        e = state.get(("mfma", r))
        if e is None:
            continue
        el, (p, op, rng) = e
        if is_mfma(I):
            c_regs = I.regs[3] if len(I.regs) > 3 else []
            ab = set(_flat(I.regs[1:3]))
            if r in ab:
                yield mfma_read_ws(op), el, p, f"{op} result -> MFMA reads it as A/B"
            elif r in c_regs or r in defs:
                if tuple(c_regs) == rng and tuple(I.defs) == rng:
                    continue                                       # accumulate chain on the same registers: back to back
                yield mfma_srcc_overlap_ws(op), el, p, f"{op} result -> MFMA C / vDst overlapping but not identical"
        else:
            yield mfma_read_ws(op), el, p, f"{op} result -> {'read' if r in uses else 'overwritten'} by {I.mnem}"
    # -- VALU-written VGPRs
    if is_mfma(I) or is_dpp(I) or I.mnem.startswith(("v_readlane", "v_readfirstlane", "v_permlane16_swap", "v_permlane32_swap")):
        need = (WS_VALU_VGPR_TO_MFMA if is_mfma(I) else WS_VALU_VGPR_TO_DPP if is_dpp(I) else
                WS_VALU_VGPR_TO_READLANE if I.mnem.startswith("v_read") else WS_VALU_VGPR_TO_PERMLANE_SWAP)
        for r in uses:
            e = state.get(("valu_vgpr", r))
            if e is not None:
                yield need, e[0], e[1][0], f"VALU-written VGPR -> {I.mnem}"
    if is_dpp(I):
        e = state.get(("valu_exec", ("exec", 0)))
        if e is not None:
            yield WS_VALU_EXEC_TO_DPP, e[0], e[1][0], "VALU-written EXEC -> DPP"
    # -- transcendental results into an ordinary VALU
    if is_valu(I) and not is_trans(I):
        for r in uses:
            e = state.get(("trans", r))
            if e is not None:
                yield WS_TRANS_TO_VALU, e[0], e[1][0], f"transcendental result -> {I.mnem}"
    # -- VALU-written SGPRs / VCC
    for r in uses:
        e = state.get(("valu_sgpr", r))
        if e is None:
            continue
        if is_vmem(I):
            yield WS_VALU_SGPR_TO_VMEM, e[0], e[1][0], f"VALU-written SGPR -> {I.mnem}"
        elif I.mnem.startswith(("v_readlane", "v_writelane")) and len(I.regs) > 2 and r in I.regs[2]:
            yield WS_VALU_SGPR_TO_LANESEL, e[0], e[1][0], f"VALU-written SGPR -> lane select of {I.mnem}"
        elif I.mnem.startswith("v_div_fmas"):
            yield WS_VALU_VCC_TO_DIV_FMAS, e[0], e[1][0], "VALU-written VCC -> v_div_fmas"
        elif is_valu(I) or is_mfma(I):
            yield WS_VALU_SGPR_TO_VALU, e[0], e[1][0], f"VALU-written SGPR / VCC -> {I.mnem} reads it"
    # -- M0
    if is_lds_dma(I) or "addtid" in I.mnem or I.mnem.startswith("s_movrel"):
        e = state.get(("salu_m0", ("m0", 0)))
        if e is not None:
            yield WS_SALU_M0_TO_LDSDMA, e[0], e[1][0], f"SALU-written M0 -> {I.mnem}"
    # -- store data
    if is_valu(I) or is_mfma(I):
        for r in defs:
            e = state.get(("store_data", r))
            if e is not None:
                yield WS_STORE_DATA_TO_VALU_WRITE, e[0], e[1][0], f"data register of a > 8-byte store -> overwritten by {I.mnem}"


def lint_kernel(name, ins):
    """-> list of findings (dicts)."""
    n = len(ins)
    if n == 0:
        return []
    at = {I.addr: i for i, I in enumerate(ins)}
    leaders = {0}
    for i, I in enumerate(ins):
        if I.mnem.startswith(("s_branch", "s_cbranch")):
            if I.target in at:
                leaders.add(at[I.target])
            if i + 1 < n:
                leaders.add(i + 1)
        elif I.mnem.startswith(("s_endpgm", "s_setpc", "s_swappc")) and i + 1 < n:
            leaders.add(i + 1)
    starts = sorted(leaders)
    blk_of = {s: b for b, s in enumerate(starts)}
    ends = starts[1:] + [n]
    succ = []
    for b, (s, e) in enumerate(zip(starts, ends)):
        last = ins[e - 1]
        out = []
        if last.mnem.startswith("s_branch"):
            if last.target in at:
                out.append(blk_of[at[last.target]])
        elif last.mnem.startswith("s_cbranch"):
            if last.target in at:
                out.append(blk_of[at[last.target]])
            if e < n:
                out.append(blk_of[e])
        elif last.mnem.startswith(("s_endpgm", "s_setpc", "s_swappc")):
            pass
        elif e < n:
            out.append(blk_of[e])
        succ.append(out)
    evs = [events_of(I, i) for i, I in enumerate(ins)]
    wss = [wait_states(I) for I in ins]
    # for a pending MFMA result an MFMA in between counts MFMA_BETWEEN: the matrix pipe is in order, the later MFMA cannot
    # issue before the earlier one has gone through its passes
    wss_m = [MFMA_BETWEEN if is_mfma(I) else w for I, w in zip(ins, wss)]

    def run_block(b, state, report):
        s, e = starts[b], ends[b]
        for i in range(s, e):
            I = ins[i]
            if state and report is not None:
                for need, el, p, rule in requirements(I, state):
                    if el < need:
                        report[(p, i, rule)] = (need, el)
            w, wm = wss[i], wss_m[i]
            if state:
                state = {k: (v[0] + (wm if k[0] == "mfma" else w), v[1]) for k, v in state.items()
                         if v[0] + (wm if k[0] == "mfma" else w) <= MAXW}
            for k, payload in evs[i].items():
                state[k] = (0, payload)
        return state

    in_state = [None] * len(starts)
    in_state[0] = {}
    work = [0]
    while work:
        b = work.pop()
        out = run_block(b, dict(in_state[b]), None)
        for t in succ[b]:
            cur = in_state[t]
            if cur is None:
                in_state[t] = dict(out); work.append(t)
                continue
            changed = False
            for k, v in out.items():
                c = cur.get(k)
                if c is None or v[0] < c[0]:
                    cur[k] = v; changed = True
            if changed:
                work.append(t)
    report = {}
    for b in range(len(starts)):
        if in_state[b] is not None:
            run_block(b, dict(in_state[b]), report)
    res = []
    for (p, i, rule), (need, el) in sorted(report.items(), key=lambda kv: kv[0][1]):
        res.append({"kernel": name, "rule": rule, "need": need, "have": el, "producer": f"{ins[p].addr:x}: {ins[p].text}",
                    "consumer": f"{ins[i].addr:x}: {ins[i].text}"})
    return res


def lint_text(text):
    findings, stats = [], {"kernels": 0, "instructions": 0, "mfma": 0}
    for name, ins in parse_objdump(text).items():
        stats["kernels"] += 1
        stats["instructions"] += len(ins)
        stats["mfma"] += sum(1 for I in ins if is_mfma(I))
        findings += lint_kernel(name, ins)
    return findings, stats


def lint_path(path):
    findings, total = [], {"kernels": 0, "instructions": 0, "mfma": 0}
    for text in disassemble(path):
        f, s = lint_text(text)
        findings += f
        for k in total:
            total[k] += s[k]
    return findings, total


# ---- the table against the compiler's recognizer -----------------------------------------------------------------------
def probe_expectations():
    """(kernel-name substring, producer regex, consumer regex, wait states the lint's table asks for, what it is)"""
    ops = ["v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x2_f32"]
    cons = [r"v_add_f32", r"ds_write_b32", r"global_store_dword", r"v_mfma_f32_32x32x2_f32 .*", r"v_readlane_b32"]
    what = ["VALU read", "DS read", "vector-memory read", "MFMA A operand", "v_readlane"]
    rows = []
    for o, op in enumerate(ops):
        for k in range(5):
            # (kind 3 of OP 2: producer and consumer are the same opcode -- first and second occurrence)
            rows.append((f"probe_mfmaILi{o}ELi{k}E", op, cons[k], mfma_read_ws(op), f"{op} -> {what[k]}"))
    rows += [
        ("probe_agpr_read", ops[0], r"v_accvgpr_read_b32", mfma_read_ws(ops[0]), "MFMA result in AGPRs -> v_accvgpr_read"),
        ("probe_mfma_c_overlap", ops[0], ops[1], mfma_srcc_overlap_ws(ops[0]), "MFMA result -> MFMA C, overlapping"),
        ("probe_mfma_c_same", ops[0], ops[0], 0, "MFMA result -> MFMA C, same registers"),
        ("probe_betweenILi2E", ops[0], r"v_add_f32", mfma_read_ws(ops[0]), "two independent MFMAs in between count one state each (LLVM)"),
        ("probe_valu_to_mfma", r"v_add_f32|v_mul_f32", ops[2], WS_VALU_VGPR_TO_MFMA, "VALU-written VGPR -> MFMA A operand"),
        ("probe_sgpr_to_valu", r"v_readlane_b32", r"v_add_f32|v_mul_f32|v_mov_b32", WS_VALU_SGPR_TO_VALU, "VALU-written SGPR -> VALU reads it"),
        ("probe_vcc_and_readfirstlane", r"v_cmp_", r"v_cndmask_b32", WS_VALU_SGPR_TO_VALU, "VALU-written VCC -> v_cndmask"),
        ("probe_vcc_and_readfirstlane", r"v_cndmask_b32", r"v_readfirstlane_b32", WS_VALU_VGPR_TO_READLANE, "VALU-written VGPR -> v_readfirstlane"),
        ("probe_sgpr_to_lanesel", r"v_readfirstlane_b32 s\d+, v\d+$", r"v_readlane_b32", WS_VALU_SGPR_TO_LANESEL, "VALU-written SGPR -> lane select"),
        ("probe_trans_to_valu", r"v_exp_f32", r"v_add_f32", WS_TRANS_TO_VALU, "transcendental -> VALU"),
        ("probe_valu_to_dpp", r"v_mov_b32_e32 v\d+, 0", r"v_mov_b32_dpp", WS_VALU_VGPR_TO_DPP, "VALU-written VGPR -> DPP (old)"),
        ("probe_store_data", r"global_store_dwordx4", r"v_pk_mov_b32|v_mul_f32|v_pk_mul_f32", WS_STORE_DATA_TO_VALU_WRITE, "16-byte store data -> VALU overwrites it"),
        ("probe_sgpr_to_vmem", r"v_readfirstlane_b32", r"buffer_load_dword", WS_VALU_SGPR_TO_VMEM, "VALU-written SGPR -> buffer load's scalar offset"),
    ]
    return rows


def measure_probes(workdir=None, hipcc=None):
    """Compile tools/asm_hazard_probe.hip with -S and read off, per probe, the wait states hipcc left between producer and consumer.
    -> list of (what, expected by the table, measured)."""
    here = os.path.dirname(os.path.abspath(__file__))
    tmp = workdir or tempfile.mkdtemp(prefix="pips_probe_")
    out = os.path.join(tmp, "probe.s")
    exe = hipcc or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([exe, "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-o", out, os.path.join(here, "asm_hazard_probe.hip")],
                   check=True, capture_output=True)
    text = open(out).read()
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        kernels[m.group(1)] = [l.strip() for l in m.group(2).splitlines() if l.strip() and not l.strip().startswith((";", ".", "//"))]
    res = []
    for name, prod, cons, want, what in probe_expectations():
        body = next((b for k, b in kernels.items() if name in k), None)
        if body is None:
            res.append((what, want, None)); continue
        pi = next((i for i, l in enumerate(body) if re.match(prod, l)), None)
        ci = None if pi is None else next((i for i in range(pi + 1, len(body)) if re.match(cons, body[i])), None)
        if pi is None or ci is None:
            res.append((what, want, None)); continue
        if prod != cons and "in between" not in what:             # the LAST producer in front of the consumer
            pi = max(i for i in range(ci) if re.match(prod, body[i]))
        ws = 0
        for l in body[pi + 1:ci]:
            mm = re.match(r"s_nop (\d+)", l)
            ws += int(mm.group(1)) + 1 if mm else (0 if l.endswith(":") else 1)
        res.append((what, want, ws))
    if workdir is None:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def main(argv):
    if "--probes" in argv:
        bad = 0
        for what, want, got in measure_probes():
            flag = "" if got == want else "   <-- differs"
            bad += bool(flag)
            print(f"{what:70s} table {want:3d}   hipcc {got}{flag}")
        return 1 if bad else 0
    verbose = "-v" in argv
    paths = [a for a in argv if not a.startswith("-")] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                        "pips_amd", "libpips_hip.so")]
    bad = 0
    for p in paths:
        findings, st = lint_path(p)
        print(f"{p}: {st['kernels']} kernels, {st['instructions']} instructions, {st['mfma']} MFMAs: {len(findings)} hazard(s)")
        for f in findings[: (len(findings) if verbose else 40)]:
            print(f"  {f['kernel'][:70]}: {f['rule']}: needs {f['need']} wait states, has {f['have']}\n      {f['producer']}\n      {f['consumer']}")
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
