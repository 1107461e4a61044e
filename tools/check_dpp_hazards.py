"""Scans gfx950 assembly (hipcc -save-temps) for the one hazard the hand-written DPP instructions of
fx_grouped.hip could hit without the compiler noticing (inline asm is opaque to its hazard recogniser):
a DPP instruction reading, as its DPP operand, a VGPR that a VALU instruction wrote fewer than two wait
states earlier. Prints the offending lines; exit status 1 if any.
   python tools/check_dpp_hazards.py file.s [file.s ...]"""
import re, sys

REG = re.compile(r'(-?\|?)(v|a)\[(\d+):(\d+)\]|(-?\|?)(v|a)(\d+)\b')

def regs(tok):
    m = re.search(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r'\bv(\d+)\b', tok)
    return {int(m.group(1))} if m else set()

def main(paths):
    bad = 0
    for path in paths:
        window = []  # (wait states this instruction provides, VGPRs it writes as a VALU op, text)
        for ln, line in enumerate(open(path), 1):
            t = line.split(';')[0].strip()
            if not t or t.startswith('.') or t.startswith(';'):
                continue
            if t.endswith(':'):
                window = []  # a branch target: the predecessor is unknown, the compiler's own code ends blocks safely
                continue
            parts = t.split(None, 1)
            op = parts[0]
            ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
            if '_dpp' in op and len(ops) >= 2:
                src = regs(ops[1])
                need = 2
                for states, written, text in reversed(window):
                    if need <= 0:
                        break
                    if written & src:
                        print(f"{path}:{ln}: {t}\n    reads {sorted(src)} through DPP {2 - need} wait state(s) after: {text}")
                        bad += 1
                        break
                    need -= states
            states = 1
            written = set()
            if op == 's_nop' and ops:
                states = int(ops[0], 0) + 1
            elif op.startswith('v_') and not op.startswith('v_cmp') and ops:
                written = regs(ops[0])
            window.append((states, written, t))
            window = window[-4:]
    return 1 if bad else 0

if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
