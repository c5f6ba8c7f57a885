"""Wait-state guards shared by the assembly generators (tools/gen_*.py).

hipcc pads nothing inside an `asm` statement, so every producer / consumer pair of tools/asm_hazard_lint.py's table that a
generator emits closer together than the table allows gets its `s_nop` from here -- ONE place holds the numbers (the lint's
table, pinned against the compiler's own hazard recognizer), the generators say which pair they are guarding.  The lint then
checks the disassembly of the built library, whatever the generators did.
"""
from asm_hazard_lint import (MFMA_PASSES, WS_SALU_M0_TO_LDSDMA, WS_VALU_SGPR_TO_LANESEL, WS_VALU_SGPR_TO_VALU, WS_VALU_SGPR_TO_VMEM,
                             WS_VALU_VGPR_TO_MFMA, mfma_read_ws)


def nops(states):
    """`s_nop` lines that cover `states` wait states (s_nop N = N + 1 states, N <= 15)."""
    out = []
    while states > 0:
        k = min(states, 16)
        out.append("s_nop %d" % (k - 1))
        states -= k
    return out


def emit_mfma_result_guard(emit, opcode, already=0):
    """In front of the first instruction that is not an MFMA accumulating on the same registers and touches the result of
    `opcode` (v_accvgpr_read, a DS / memory instruction, a VALU overwriting it): the states the lint's table asks for, minus the
    `already` instructions known to sit in between.  (Rounds 4-5 pasted `s_nop 15; s_nop 15` = 32 states in five generators.)"""
    assert opcode in MFMA_PASSES, opcode
    for line in nops(mfma_read_ws(opcode) - already):
        emit(line)


def emit_sgpr_to_valu_guard(emit, already=0):
    """v_cmp / v_readlane / v_readfirstlane / carry-out wrote an SGPR or VCC, a VALU instruction reads it: 2 states (gfx940+)."""
    for line in nops(WS_VALU_SGPR_TO_VALU - already):
        emit(line)


def emit_sgpr_to_vmem_guard(emit, already=0):
    for line in nops(WS_VALU_SGPR_TO_VMEM - already):
        emit(line)


def emit_sgpr_to_lanesel_guard(emit, already=0):
    for line in nops(WS_VALU_SGPR_TO_LANESEL - already):
        emit(line)


def emit_m0_guard(emit, already=0):
    """SALU wrote M0, the next instruction is an LDS-DMA load (`buffer_load ... lds`): 1 state."""
    for line in nops(WS_SALU_M0_TO_LDSDMA - already):
        emit(line)


def emit_valu_to_mfma_guard(emit, already=0):
    """a VALU instruction (v_accvgpr_write included) wrote a register the next MFMA reads as A / B / C: 2 states."""
    for line in nops(WS_VALU_VGPR_TO_MFMA - already):
        emit(line)
