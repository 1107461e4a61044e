"""Scans gfx950 assembly (hipcc -S / -save-temps) for the one hazard the hand-written DPP instructions of
fx_grouped.hip could hit without the compiler noticing (inline asm is opaque to its hazard recogniser):
a DPP instruction reading, as its DPP operand, a VGPR that a VALU instruction wrote fewer than two wait
states earlier. Every instruction counts as one wait state, `s_nop N` as N + 1. The search walks the
control-flow graph backwards from each DPP instruction: through the textual predecessor (unless that is an
unconditional branch or the end of the program) and, at a label, through every branch that targets it.
Prints the offending paths; exit status 1 if any.
   python tools/check_dpp_hazards.py file.s [file.s ...]"""
import re
import sys


def regs(tok):
    m = re.search(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r'\bv(\d+)\b', tok)
    return {int(m.group(1))} if m else set()


class Ins:
    __slots__ = ("ln", "text", "op", "ops", "states", "written", "labels", "target", "falls")


def parse(path):
    """Instructions of one file in textual order; `labels`: the labels directly in front of an instruction."""
    out, pending = [], []
    for ln, line in enumerate(open(path), 1):
        t = line.split(';')[0].strip()
        if not t or t.startswith('.') and not t.endswith(':'):
            continue
        if t.endswith(':'):
            pending.append(t[:-1])
            continue
        parts = t.split(None, 1)
        i = Ins()
        i.ln, i.text, i.op = ln, t, parts[0]
        i.ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
        i.states = 1
        i.written = set()
        if i.op == 's_nop' and i.ops:
            i.states = int(i.ops[0], 0) + 1
        elif i.op.startswith('v_') and not i.op.startswith('v_cmp') and i.ops:
            i.written = regs(i.ops[0])
        i.labels, pending = pending, []
        i.target = i.ops[0] if i.op.startswith(('s_branch', 's_cbranch')) and i.ops else None
        i.falls = i.op not in ('s_branch', 's_endpgm', 's_setpc_b64')  # control reaches the next instruction
        out.append(i)
    return out


def main(paths):
    bad = 0
    for path in paths:
        ins = parse(path)
        jumps = {}  # label -> indices of the branches that target it
        for k, i in enumerate(ins):
            if i.target:
                jumps.setdefault(i.target, []).append(k)
        for k, i in enumerate(ins):
            if '_dpp' not in i.op or len(i.ops) < 2:
                continue
            src = regs(i.ops[1])
            # backwards over the CFG: (index of the instruction to look at, wait states still needed)
            stack, seen = [], set()

            def preds(at, need):
                if at > 0 and ins[at - 1].falls:
                    stack.append((at - 1, need))
                for lab in ins[at].labels:
                    for j in jumps.get(lab, []):
                        stack.append((j, need))

            preds(k, 2)
            while stack:
                at, need = stack.pop()
                if need <= 0 or (at, need) in seen:
                    continue
                seen.add((at, need))
                p = ins[at]
                if p.written & src:
                    print(f"{path}:{i.ln}: {i.text}\n    reads {sorted(src)} through DPP {2 - need} wait state(s) after "
                          f"line {p.ln}: {p.text}")
                    bad += 1
                    break
                preds(at, need - p.states)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
