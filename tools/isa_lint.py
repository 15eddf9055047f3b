#!/usr/bin/env python3
"""ISA lint of the kernels' final assembly (make -C lyra_amd/csrc ODIR=obj_asm asm):
   python tools/isa_lint.py lyra_amd/csrc/obj_asm/*.s
gfx950 does not interlock a vector read against an MFMA that is still writing its destination (tools/hazard_probe.hip).
The compiler pads its own instructions with the wait states; it knows nothing about the instructions inside an asm block.
Rule: no instruction inside an asm block reads a register that an MFMA wrote fewer than passes + 3 wait states earlier
(every instruction in between counts one, `s_nop n` counts n + 1; passes: 8 for the 16x16 shapes used here, 16 otherwise --
the compiler's own padding for a vector read is passes + 2 or + 3).
Prints one line per kernel; exit status 1 if a kernel breaks the rule."""
import re, sys

MAX_WAIT = 19


def need(mfma):
    return (8 if '_16x16x' in mfma else 16) + 3


def regs(tok):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok):
        out.add(int(a))
    return out


def lint(path):
    bad = []
    s = open(path).read()
    for m in re.finditer(r'^(_ZN4lyra\d+(\w+?_kernel)\w*):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
        name = m.group(2)
        lines = [l.strip() for l in m.group(3).split('\n')]
        ins = [l for l in lines if (l and not l.startswith(('.', ';', '//')) and not l.endswith(':')) or l.startswith(';;#ASM')]
        worst, sites = None, 0
        for i, l in enumerate(ins):
            if not l.startswith('v_mfma'):
                continue
            dst = regs(l.split(',')[0])
            inasm, wait = False, 0
            for t in ins[i + 1:i + 1 + 2 * MAX_WAIT]:
                if t.startswith(';;#ASMSTART'):
                    inasm = True
                    continue
                if t.startswith(';;#ASMEND'):
                    inasm = False
                    continue
                if wait >= need(l) or not dst:
                    break
                ops = t.split(',')
                if inasm and len(ops) > 1 and regs(','.join(ops[1:])) & dst:
                    sites += 1
                    if worst is None or wait < worst[0]:
                        worst = (wait, l, t)
                    break
                wait += 1 + (int(t.split()[1]) if t.startswith('s_nop') else 0)
                if not inasm:
                    dst -= regs(ops[0])       # overwritten: no longer that MFMA's result
        print("%-28s %-22s %s" % (path.split('/')[-1], name, "ok" if worst is None else
              "asm reads an MFMA result after %d wait states (%d sites): %s  ->  %s" % (worst[0], sites, worst[1], worst[2])))
        if worst is not None:
            bad.append(name)
    return bad


if __name__ == "__main__":
    bad = [b for p in sys.argv[1:] for b in lint(p)]
    sys.exit(1 if bad else 0)
