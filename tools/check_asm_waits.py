#!/usr/bin/env python3
"""Build-time check of the hand-counted waits behind inline-assembly loads (round-2/3 advisor item).

csrc/cnn.hip fetches a tap's weights with `asm volatile("global_load_dwordx4 %0, ...")` and waits with an inline `s_waitcnt vmcnt(n)`
whose n is counted by hand: the compiler believes the destination registers hold their value as soon as the asm statement has
"executed", so a register copy, a spill or any other use that it schedules between the load and the wait would read stale data --
silently, and only after some compiler update.  This script compiles the file to assembly (device code only) and, for every function
that contains such loads, walks the text: from an inline-asm `global_load` (between ;;#ASMSTART / ;;#ASMEND) to the next inline-asm
`s_waitcnt vmcnt`, no instruction OUTSIDE inline assembly may name the load's destination registers.  (Loads stay in flight across
later waits with n > 0; which wait releases which load is what the hand count asserts -- the check covers the failure a compiler can
introduce on its own: a use with no wait at all in between.)  The s_load prefetches of k_tiny_select (lopq_search.hip) carry their
wait inside the same asm statement or are never read; their registers are covered by the same rule with `s_waitcnt lgkmcnt`.

usage: check_asm_waits.py <file.s> [<file.s> ...]   (exit code 1 on a violation)"""
import re
import sys


def regs_of(operand):
    """'v[12:15]' -> {'v12',...}; 'v7' -> {'v7'}; same for s registers"""
    out = set()
    for m in re.finditer(r"\b([vs])\[(\d+):(\d+)\]", operand):
        out.update("%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    for m in re.finditer(r"\b([vs])(\d+)\b", operand):
        out.add("%s%d" % (m.group(1), int(m.group(2))))
    return out


def check(path):
    bad = 0
    func, in_asm, pending = None, False, []   # pending: (kind, regs, line_no, text)
    n_loads = 0
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func, pending, in_asm = m.group(1), [], False
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.startswith("s_endpgm") or t.startswith("s_branch") or t.startswith("s_setpc"):
            pending = []  # the text that follows is not reached from here (linear walk: straight-line code and fall-through paths only)
            continue
        if in_asm:
            if t.startswith("global_load") or t.startswith("buffer_load"):
                pending.append(("vm", regs_of(t.split(",")[0]), ln, t))
                n_loads += 1
            elif t.startswith("s_load"):
                pending.append(("lgkm", regs_of(t.split(",")[0]), ln, t))
                n_loads += 1
            elif t.startswith("s_waitcnt"):
                if "vmcnt" in t:
                    pending = [p for p in pending if p[0] != "vm"]
                if "lgkmcnt" in t:
                    pending = [p for p in pending if p[0] != "lgkm"]
            continue
        if not pending:
            continue
        # a compiler-placed full wait releases everything of its kind as well
        if t.startswith("s_waitcnt"):
            if "vmcnt(0)" in t:
                pending = [p for p in pending if p[0] != "vm"]
            if "lgkmcnt(0)" in t:
                pending = [p for p in pending if p[0] != "lgkm"]
            continue
        used = regs_of(t)
        for kind, regs, l0, txt in pending:
            hit = used & regs
            if hit:
                print("%s:%d: %s\n    touches %s of the inline-asm load at line %d (%s) before any inline wait [%s]"
                      % (path, ln, t, sorted(hit), l0, txt, func))
                bad += 1
    return bad, n_loads


if __name__ == "__main__":
    total_bad = total_loads = 0
    for p in sys.argv[1:]:
        b, n = check(p)
        total_bad += b
        total_loads += n
    print("check_asm_waits: %d inline-asm loads checked, %d violations" % (total_loads, total_bad))
    sys.exit(1 if total_bad else 0)
