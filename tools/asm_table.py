#!/usr/bin/env python3
"""One line per step kernel of a device listing: instructions in the substep loop (tools/asm_hist.largest_loop), of which packed,
VGPRs, AGPRs, scratch bytes, LDS bytes (a jump in LDS = the compiler promoted a per-lane array to LDS: round 4 lost 13 us per launch to one).  With two listings: the two tables side by side (what a source change did to EVERY instantiation).

usage: asm_table.py <listing.s> [<other listing.s>] [substring filter]"""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from asm_hist import largest_loop  # noqa: E402


def table(path):
    out, name, body, on = {}, None, [], False
    meta = {}
    for ln in open(path):
        ln = ln.rstrip("\n")
        if not on:
            m = re.match(r"^(_Z\w*qg_step_kernel\w*):", ln)
            if m:
                name, body, on = m.group(1), [], True
            m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize): (\d+)", ln)
            if m and name:
                meta.setdefault(name, {})[m.group(1)] = int(m.group(2))
            continue
        if ln.startswith(".Lfunc_end"):
            loop = largest_loop(body)
            ins = [l for l in loop if re.match(r"\s+[a-z]", l) and not l.lstrip().startswith((".", ";"))]
            out[name] = {"loop": len(ins), "packed": sum(1 for l in ins if re.match(r"\s+v_pk_", l)), "total": sum(
                1 for l in body if re.match(r"\s+[a-z]", l) and not l.lstrip().startswith((".", ";")))}
            on = False
            continue
        body.append(ln)
    for k, v in meta.items():
        if k in out:
            out[k].update(v)
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", open(path).read(), re.S):
        g = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", m.group(2))
        if m.group(1) in out and g:
            out[m.group(1)]["LDSByteSize"] = int(g.group(1))
    return out


def short(name):
    m = re.match(r"_Z\d+(qg_step_kernel\w*?)I(.*?)EvPK", name)
    return f"{m.group(1)}<{m.group(2)}>" if m else name[:60]


def main():
    paths = [a for a in sys.argv[1:] if a.endswith(".s")]
    filt = [a for a in sys.argv[1:] if not a.endswith(".s")]
    tabs = [table(p) for p in paths]
    for k in tabs[0]:
        if filt and filt[0] not in k:
            continue
        cells = []
        for t in tabs:
            r = t.get(k)
            cells.append("        (absent)" if r is None else
                         f"loop {r['loop']:5d} ({r['packed']:4d} pk) all {r['total']:5d}  V {r.get('NumVgprs', -1):3d} A {r.get('NumAgprs', -1):3d} scr {r.get('ScratchSize', -1):3d} lds {r.get('LDSByteSize', -1):5d}")
        extra = ""
        if len(tabs) == 2 and all(t.get(k) for t in tabs):
            extra = f"   {tabs[1][k]['loop'] - tabs[0][k]['loop']:+5d}"
        print(f"{short(k):58s} " + "  |  ".join(cells) + extra)


if __name__ == "__main__":
    main()
