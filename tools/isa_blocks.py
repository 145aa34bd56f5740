#!/usr/bin/env python3
"""Per-basic-block instruction counts of one kernel in a hipcc --save-temps .s file.

    python tools/isa_blocks.py <file.s> <substring of the kernel's symbol> [min VALU]
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().endswith(l.split(":")[0][:8]) is False and ":" in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", 0, 0, 0, []]
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur)
            cur = [m.group(1), 0, 0, 0, []]
            continue
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            cur[1] += 1
        elif op.startswith("s_"):
            cur[2] += 1
        else:
            cur[3] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            cur[4].append(op[2:] + "->" + t.split()[1])
    blocks.append(cur)
    total = 0
    for b in blocks:
        total += b[1]
        if b[1] >= floor:
            print("%-12s V=%4d S=%4d M=%3d %s" % (b[0], b[1], b[2], b[3], " ".join(b[4])))
    print("total VALU", total)


main()
