#!/usr/bin/env python3
"""Issue classes of a kernel's substep loop, priced with the per-SIMD costs tools/ubench/vgpr_bank measured on gfx950
(profiles/r03/ubench_vgpr_bank.txt): VOP2 2.3 cycles, VOP3 / literal forms 2.7, anything that reads an SGPR (constant bus, VCC
included), DPP, v_cndmask and packed FP32 4.4, transcendentals 8.6.

usage: asm_classes.py <listing.s> <substring of the mangled kernel name>"""
import collections
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from asm_hist import kernel_lines, largest_loop  # noqa: E402

COST = {"vop2": 2.3, "vop3/literal": 2.7, "sgpr operand": 4.4, "dpp": 4.4, "cndmask": 4.4, "packed": 4.4, "trans": 8.6, "salu/other": 0.0}
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def classify(ln):
    ln = ln.split(";")[0].strip()
    if not ln or ln.startswith(".") or ln.endswith(":"):
        return None
    op = ln.split()[0]
    if not op.startswith("v_"):
        return "salu/other"
    args = ln[len(op):]
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "packed"
    if "dpp" in op or "quad_perm" in args or "row_" in args:
        return "dpp"
    if op.startswith("v_cndmask"):
        return "cndmask"
    srcs = args.split(",")[1:] if not op.startswith("v_cmp") else args.split(",")
    if any(re.match(r"\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec)", a) for a in srcs):
        return "sgpr operand"
    if op.endswith("_e32") and not re.search(r"0x[0-9a-f]+", args):
        return "vop2"
    return "vop3/literal"


def main():
    lines = kernel_lines(sys.argv[1], sys.argv[2])
    body = largest_loop(lines)
    cnt = collections.Counter(c for c in map(classify, body) if c)
    tot = sum(cnt.values())
    cyc = sum(COST[c] * n for c, n in cnt.items())
    print(f"{sys.argv[2]}: {tot} instructions in the substep loop, {cyc:.0f} SIMD cycles at saturation ({cyc / max(1, tot - cnt['salu/other']):.2f} per VALU instruction)")
    for c, n in cnt.most_common():
        print(f"  {c:14s} {n:5d}  {100.0 * n / tot:5.1f} %   {COST[c] * n:7.0f} cycles")


if __name__ == "__main__":
    main()
