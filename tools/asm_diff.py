#!/usr/bin/env python3
"""Compare the kernels of two device listings (hipcc -save-temps .s files) instruction by instruction.

usage: asm_diff.py <base.s> <new.s> [substring of the mangled names to look at]
       asm_diff.py --loops <base.s> <new.s> [substring]     compare only each kernel's substep loop (the loop with the most multiply-adds)
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


def loops(path):
    """kernel name -> instructions of its substep loop (asm_hist.largest_loop on the raw listing), labels renumbered"""
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    from asm_hist import kernel_lines, largest_loop
    out = {}
    for k in kernels(path):
        if "qg_step_kernel" not in k:
            continue
        try:
            lp = largest_loop(kernel_lines(path, k))
        except Exception:
            continue
        body = []
        for ln in lp or []:
            t = ln.split(";")[0].strip()
            if not t or t.startswith("."):
                continue
            body.append(re.sub(r"\.LBB\d+_", ".LBB_", t))
        if body:
            out[k] = body
    return out


def main():
    only_loops = len(sys.argv) > 1 and sys.argv[1] == "--loops"
    if only_loops:
        sys.argv.pop(1)
    a, b = (loops if only_loops else kernels)(sys.argv[1]), (loops if only_loops else kernels)(sys.argv[2])
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
