"""Post-pass over hipcc's gfx950 device assembly, run by build.py between `hipcc -S --cuda-device-only` and the assembler.

Measured on an MI355X (tools/valu_issue_ceiling.hip, profiles/r5_valu_issue_ceiling.txt): two VOP2 `v_cndmask_b32_e32 ..., vcc`
issued back to back by one wavefront cost 16 cycles of its SIMD EACH (at any number of wavefronts per SIMD: throughput, not
latency), while the VOP3 encoding of the very same select (`v_cndmask_b32_e64 ..., vcc`, or an SGPR-pair mask) costs 4.3, and a
VOP2 select between unrelated vector instructions 2.5 - 3.  hipcc shrinks every select to VOP2 when it can and emits exactly the
slow pattern for `x = c ? a : x; y = c ? b : y;` (v_cmp -> vcc, s_nop 1, v_cndmask_e32, v_cndmask_e32): 44 instead of 12 SIMD
cycles per group at two wavefronts per SIMD.  LLVM has no switch for the shrink of one opcode, so the rule is applied here:

  every run of >= 2 VOP2 selects on vcc with no other vector-ALU instruction between them is re-encoded as VOP3 (same operands,
  same semantics, 8 bytes instead of 4), member by member, where the VOP3 form is legal on gfx9: src0 a VGPR or an inline
  constant (no 32-bit literal, no second scalar operand beside vcc).

Nothing else is touched; the result is assembled by the same LLVM (clang -x assembler -mcpu=gfx950).
"""
from __future__ import annotations

import re

VERSION = "cndmask-e64-runs-v1"

_CND = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+)(v\d+)\s*,\s*([^,]+?)\s*,\s*(v\d+)\s*,\s*vcc\s*(;.*)?$")
_INLINE_FLOATS = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0", "0.15915494", "0.15915494309189532"}
_LABEL = re.compile(r"^[.\w$@]+:")
_VALU = re.compile(r"^\s*v_")
_CTRL = re.compile(r"^\s*(s_cbranch|s_branch|s_setpc|s_swappc|s_endpgm|s_barrier|s_call)")


def _src0_legal_in_vop3(op: str) -> bool:
    op = op.strip()
    if re.fullmatch(r"v\d+", op):
        return True
    if re.fullmatch(r"-?\d+", op):
        return -16 <= int(op) <= 64
    return op in _INLINE_FLOATS


def rewrite_text(text: str) -> tuple[str, dict]:
    lines = text.split("\n")
    run: list[int] = []          # indices of the VOP2 selects of the current run
    stats = {"selects_vop2": 0, "runs": 0, "rewritten": 0, "kept_literal_or_sgpr": 0}

    def flush():
        if len(run) >= 2:
            stats["runs"] += 1
            for i in run:
                m = _CND.match(lines[i])
                if _src0_legal_in_vop3(m.group(4)):
                    tail = f" {m.group(6)}" if m.group(6) else ""
                    lines[i] = f"{m.group(1)}v_cndmask_b32_e64{m.group(2)}{m.group(3)}, {m.group(4).strip()}, {m.group(5)}, vcc{tail}"
                    stats["rewritten"] += 1
                else:
                    stats["kept_literal_or_sgpr"] += 1
        run.clear()

    for i, ln in enumerate(lines):
        if _CND.match(ln):
            stats["selects_vop2"] += 1
            run.append(i)
        elif _VALU.match(ln) or _LABEL.match(ln) or _CTRL.match(ln):
            flush()              # another vector-ALU instruction, a label or a branch ends the run
        # scalar, LDS, memory instructions, s_nop / s_waitcnt, comments and directives do not separate two selects in the VALU
    flush()
    return "\n".join(lines), stats


def rewrite_file(path: str) -> dict:
    with open(path) as f:
        text = f.read()
    out, stats = rewrite_text(text)
    with open(path, "w") as f:
        f.write(out)
    return stats
