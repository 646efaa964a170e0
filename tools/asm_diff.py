#!/usr/bin/env python3
"""Compare the kernels of two device listings (hipcc -save-temps .s files) instruction by instruction.

usage: asm_diff.py <base.s> <new.s> [substring of the mangled names to look at]
Labels are renumbered per kernel (.LBB<function index>_<n> changes when kernels are added or removed elsewhere in the code object);
comments, .loc / .file / .cfi directives and blank lines are ignored.  Prints one line per kernel: identical / differs (first
differing instruction) / only in one listing."""
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    for ln in open(path):
        if name is None:
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                name, body = m.group(1), []
            continue
        if ln.startswith(".Lfunc_end"):
            out[name] = body
            name = None
            continue
        t = ln.split(";")[0].strip()
        if not t or t.startswith((".loc", ".file", ".cfi", ".p2align 6,")):
            continue
        t = re.sub(r"\.LBB\d+_", ".LBB_", t)
        t = re.sub(r"\.Ltmp\d+", ".Ltmp", t)
        body.append(t)
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    key = sys.argv[3] if len(sys.argv) > 3 else ""
    same = diff = 0
    for k in sorted(set(a) | set(b)):
        if key not in k:
            continue
        if k not in a or k not in b:
            print(("only in new : " if k in b else "only in base: ") + k)
            continue
        if a[k] == b[k]:
            same += 1
            continue
        diff += 1
        i = next((i for i, (x, y) in enumerate(zip(a[k], b[k])) if x != y), min(len(a[k]), len(b[k])))
        print(f"DIFFERS {k}: {len(a[k])} vs {len(b[k])} lines, first at {i}: {a[k][i] if i < len(a[k]) else '-'}  |  {b[k][i] if i < len(b[k]) else '-'}")
    print(f"{same} kernels identical, {diff} differ")


if __name__ == "__main__":
    main()
