#!/usr/bin/env python3
"""isa_lint.py — build-time check of the hand-placed ("untracked") prefetch idiom in the compiled gfx950 ISA.

The kernels issue some loads from inline assembly (`global_load_* vD, ...` inside ;;#ASMSTART / ;;#ASMEND) so that the
compiler does not know about their latency, and wait for them with a hand-placed, counted `s_waitcnt vmcnt(N)` (also
inline assembly).  Between the two the destination VGPR holds garbage: the compiler must not have placed ANY instruction
that reads or overwrites it there (a `v_mov` copy at a loop back edge, a duplicate-operand copy in front of the wait —
both happened in round 1).  This script proves the absence of such an instruction on every path of the control-flow
graph of every kernel in an assembly listing (`make asm`, `make asm-policy`):

  * forward may-analysis over basic blocks; state = the set of VGPRs with a hand-placed load in flight;
  * an inline-asm `global_load_*` adds its destination registers;
  * an inline-asm `s_waitcnt` whose vmcnt field is present, or ANY `s_waitcnt vmcnt(0)`, clears the set (the hand-placed
    counted waits are written for "everything issued before the stores since"; their counts are checked by the
    full-size determinism tests, not here);
  * any other instruction that names an in-flight register (source or destination) is a violation.

Exit status 0 = clean.  Used by tests/test_isa_lint.py (-m "not gpu").
"""
import re
import sys
from collections import defaultdict

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LOAD = re.compile(r"^\s*(global|buffer|flat)_load_(?!lds_)(\w+)\s+(v\d+|v\[\d+:\d+\])")  # (an LDS-DMA load has no register destination)
WIDTH = {"dwordx2": 2, "dwordx3": 3, "dwordx4": 4}


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class Inst:
    __slots__ = ("text", "line", "asm")

    def __init__(self, text, line, asm):
        self.text, self.line, self.asm = text, line, asm


def parse_functions(path):
    """-> {name: [(label or None, [Inst])]} : basic blocks in layout order."""
    funcs = {}
    cur = None
    in_asm = False
    is_code = False
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            line = raw.rstrip("\n")
            s = line.strip()
            if s.startswith(".type") and "@function" in s:
                is_code = True
                continue
            m = FUNC.match(line)
            if m and is_code and not line.startswith(".L"):
                cur = funcs.setdefault(m.group(1), [[None, []]])
                is_code = False
                continue
            if cur is None:
                continue
            if s.startswith(".Lfunc_end"):
                cur = None
                continue
            m = LABEL.match(line)
            if m:
                cur.append([m.group(1), []])
                continue
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            code = s.split(";")[0].strip()
            if code:
                cur[-1][1].append(Inst(code, ln, in_asm))
    return funcs


def successors(blocks):
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab}
    succ = defaultdict(set)
    for i, (_, insts) in enumerate(blocks):
        fall = True
        for ins in insts:
            op = ins.text.split()[0]
            if op == "s_branch":
                succ[i].add(index[ins.text.split()[1]])
                fall = False
            elif op.startswith("s_cbranch"):
                succ[i].add(index[ins.text.split()[1]])
            elif op in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and i + 1 < len(blocks):
            succ[i].add(i + 1)
    return succ


def is_clearing_wait(ins):
    if not ins.text.startswith("s_waitcnt"):
        return False
    if "vmcnt(0)" in ins.text:
        return True
    if ins.asm and "vmcnt(" in ins.text:
        return True
    # numeric immediates (__builtin_amdgcn_s_waitcnt): vmcnt is bits 3:0 and 15:14 on gfx9
    m = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)$", ins.text)
    if m:
        imm = int(m.group(1), 0)
        return ((imm & 0xF) | ((imm >> 14) & 0x3) << 4) == 0
    return False


def transfer(insts, state, report):
    st = set(state)
    for ins in insts:
        if is_clearing_wait(ins):
            st.clear()
            continue
        m = LOAD.match(ins.text) if ins.asm else None
        if m:
            dst = vregs(m.group(3))
            first = min(dst)
            dst = set(range(first, first + WIDTH.get(m.group(2), 1)))
            # the address operands are read at issue: they must not be in flight either
            rest = ins.text[m.end():]
            bad = vregs(rest) & st
            if bad and report is not None:
                report.append((ins.line, ins.text, sorted(bad)))
            st |= dst
            continue
        if st:
            bad = vregs(ins.text) & st
            if bad and report is not None:
                report.append((ins.line, ins.text, sorted(bad)))
    return st


def lint_function(blocks):
    succ = successors(blocks)
    n = len(blocks)
    state_in = [set() for _ in range(n)]
    work = list(range(n))
    n_loads = sum(1 for _, insts in blocks for i in insts if i.asm and LOAD.match(i.text))
    if n_loads == 0:
        return 0, []
    while work:
        i = work.pop()
        out = transfer(blocks[i][1], state_in[i], None)
        for j in succ[i]:
            if not out <= state_in[j]:
                state_in[j] |= out
                work.append(j)
    report = []
    for i in range(n):
        transfer(blocks[i][1], state_in[i], report)
    return n_loads, sorted(set((l, t, tuple(b)) for l, t, b in report))


def lint_file(path):
    """-> (summary {kernel: hand-placed loads}, violations [(kernel, line, text, regs)])"""
    summary, violations = {}, []
    for name, blocks in parse_functions(path).items():
        n_loads, rep = lint_function(blocks)
        if n_loads:
            summary[name] = n_loads
        violations += [(name, l, t, list(b)) for l, t, b in rep]
    return summary, violations


def main(argv):
    rc = 0
    for path in argv[1:]:
        summary, violations = lint_file(path)
        for k, n in summary.items():
            print(f"{path}: {k}: {n} hand-placed load(s)")
        for k, l, t, b in violations:
            print(f"{path}:{l}: {k}: `{t}` touches in-flight v{b}", file=sys.stderr)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
