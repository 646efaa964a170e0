#!/usr/bin/env python3
"""Instruction histogram of one kernel of the device listing (make -C quadruped-gym_amd/csrc asm).

usage: asm_hist.py <listing.s> <substring of the mangled kernel name> [--loop]

--loop: only the largest backward-branch loop body (the substep loop of the step kernels).
Prints totals per class (VALU / DPP / SALU / VMEM / LDS / cndmask / mov ...) and the top opcodes.
"""
import collections
import re
import sys


def kernel_lines(path, key):
    out, on = [], False
    for ln in open(path):
        if not on:
            if ln.startswith("_Z") and key in ln.split(":")[0] and ":" in ln:
                on = True
            continue
        if ln.startswith(".Lfunc_end"):
            break
        out.append(ln.rstrip("\n"))
    return out


def largest_loop(lines):
    """The loop (label .. backward branch) with the most floating-point multiply-adds: the substep loop of a step kernel (rare-path
    loops of the epilogue can span more instructions)."""
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    best, best_score = None, -1
    for i, ln in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            score = sum(1 for l in lines[a:i + 1] if re.match(r"\s+v_(pk_)?(fma|fmac|mul)_f32", l))
            if score > best_score:
                best, best_score = (a, i), score
    return lines[best[0]:best[1] + 1] if best else lines


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = kernel_lines(path, key)
    if not lines:
        sys.exit("kernel not found")
    if "--loop" in sys.argv:
        lines = largest_loop(lines)
    ops = collections.Counter()
    dpp = 0
    for ln in lines:
        m = re.match(r"\s+([a-z_0-9]+)\s", ln + " ")
        if not m or ln.lstrip().startswith((".", ";")):
            continue
        op = m.group(1)
        if not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", op):
            continue
        ops[op] += 1
        if "quad_perm" in ln or "row_" in ln or "_dpp" in op:
            dpp += 1
    tot = sum(ops.values())
    cls = collections.Counter()
    for op, c in ops.items():
        if op.startswith("v_"):
            cls["VALU"] += c
            if op.startswith("v_cndmask"):
                cls["  v_cndmask"] += c
            elif op.startswith("v_mov") or op.startswith("v_accvgpr"):
                cls["  v_mov/accvgpr"] += c
            elif op.startswith("v_pk_"):
                cls["  v_pk_*"] += c
            elif op.startswith("v_cmp"):
                cls["  v_cmp"] += c
            elif re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)", op):
                cls["  trans"] += c
        elif op.startswith("s_"):
            cls["SALU/branch/wait"] += c
            if op.startswith("s_waitcnt") or op.startswith("s_nop"):
                cls["  s_waitcnt/s_nop"] += c
        elif op.startswith("ds_"):
            cls["LDS"] += c
        else:
            cls["VMEM"] += c
    print(f"{key}: {tot} instructions" + (" (largest loop)" if "--loop" in sys.argv else ""))
    for k in ["VALU", "  v_cndmask", "  v_mov/accvgpr", "  v_cmp", "  v_pk_*", "  trans", "SALU/branch/wait", "  s_waitcnt/s_nop", "LDS", "VMEM"]:
        print(f"  {k:22s} {cls[k]}")
    print(f"  with DPP operand       {dpp}")
    print("  top:", ", ".join(f"{o} {c}" for o, c in ops.most_common(14)))


if __name__ == "__main__":
    main()
