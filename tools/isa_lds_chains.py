#!/usr/bin/env python3
"""Per kernel of an ISA listing (`hipcc -S --cuda-device-only`): how many times the code waits for LDS data
(`s_waitcnt lgkmcnt(n)` with at least one ds_read issued since the previous wait) against its vector instruction
count, and how many reads each wait covers.  Many waits covering one or two reads each = a chain of dependent LDS round
trips (~100+ cycles each) the other waves of the SIMD have to hide -- how the serial binary searches of the Kraskov and
split rank kernels were found (profiles/tuning_r02.md).

usage: tools/isa_lds_chains.py file.s [substring of the kernel symbol]
"""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    name, rows = None, []
    valu = reads = pending = waits = 0
    loops = 0
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, valu, reads, pending, waits, loops = m.group(1), 0, 0, 0, 0, 0
            continue
        if name is None:
            continue
        t = l.strip()
        if t.startswith("v_"):
            valu += 1
        elif t.startswith("ds_read") or t.startswith("ds_load"):
            reads += 1
            pending += 1
        elif t.startswith("s_waitcnt") and "lgkmcnt" in t:
            if pending:
                waits += 1
                pending = 0
        elif t.startswith("s_cbranch"):
            loops += 1
        elif t.startswith(".Lfunc_end"):  # (not s_endpgm: a kernel with an early return has several)
            if want in name:
                rows.append((name, valu, reads, waits, loops))
            name = None
    print(f"{'kernel':90s} {'VALU':>7s} {'ds_read':>8s} {'LDS waits':>9s} {'reads/wait':>10s} {'branches':>8s}   (static counts)")
    for name, valu, reads, waits, loops in rows:
        print(f"{name[:90]:90s} {valu:7d} {reads:8d} {waits:9d} {reads / max(waits, 1):10.1f} {loops:8d}")


if __name__ == "__main__":
    main()
