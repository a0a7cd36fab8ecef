#!/usr/bin/env python3
"""Where in a kernel's ISA the high VGPR indices are used: per window of instructions, the largest VGPR index touched and
the dominant opcodes.  Input: an assembly file from `hipcc -S --cuda-device-only`, and a substring of the kernel symbol.

usage: tools/vgpr_profile.py file.s _ZN3crf16mi_binned_kernelILi64ELb1ELi2ELi64EEE [window]
"""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    sym = sys.argv[2]
    window = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    start = next(i for i, l in enumerate(lines) if l.startswith(sym) and ": " in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith((".", ";"))]
    print(len(body), "instructions")
    for w in range(0, len(body), window):
        seg = body[w:w + window]
        mx = 0
        for l in seg:
            for m in re.finditer(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", l):
                mx = max(mx, int(m.group(1) or m.group(3)))
        ops = {}
        for l in seg:
            o = l.split()[0]
            ops[o] = ops.get(o, 0) + 1
        top = sorted(ops.items(), key=lambda x: -x[1])[:5]
        print(f"{w:6d} max v{mx:<4d}", " ".join(f"{o}:{n}" for o, n in top))


if __name__ == "__main__":
    main()
