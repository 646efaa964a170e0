"""Live-VGPR profile of one kernel's main loop from the compiler's assembly listing (make -C quadruped-gym_amd/csrc asm writes
/tmp/qg_capi-hip-amdgcn-amd-amdhsa-gfx950.s).  Approximate: the loop body is treated as straight-line code (forward branches
ignored), the loop-carried set is found by iterating.  Prints the live count every `stride` instructions with the instruction
there, and the peak -- enough to see WHERE in the substep the register pressure sits.
usage: python tools/asm_liveness.py <listing.s> <kernel-symbol-substring> [stride]"""
import re
import sys

NO_DST = ("global_store", "buffer_store", "ds_write", "flat_store", "s_", "v_cmp", "v_cmpx", "scratch_store", "ds_bpermute_NOPE")
RMW = ("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_movrel")


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    op, _, rest = line.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
    if op.startswith(("v_readlane", "v_readfirstlane")):
        return op, [], [r for o in ops[1:] for r in regs(o)]
    if op.startswith(NO_DST) and not op.startswith(("s_load", "s_buffer")):
        return op, [], [r for o in ops for r in regs(o)]
    if not ops:
        return op, [], []
    d = regs(ops[0])
    u = [r for o in ops[1:] for r in regs(o)]
    if op.startswith(RMW) or "dpp" in line and "bound_ctrl" not in line:
        u = u + d
    return op, d, u


def main(path, sym, stride=100):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # main loop: from the first "Inner Loop Header" label to the last branch back to a label at or before it
    hdr = next(i for i, l in enumerate(body) if "Loop Header" in l)
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    label = body[hdr].split(":")[0]
    # the loop is usually rotated: its latch block sits BEFORE the header and is reached by a branch from the end of the body
    back, latch = None, None
    for i in range(len(body) - 1, hdr, -1):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\w+)", body[i])
        if m and labels.get(m.group(1), len(body)) <= hdr:
            back, latch = i, labels[m.group(1)]
            break
    seq = [parse(l) for l in body[hdr:back + 1]]
    if latch < hdr:
        seq = seq + [parse(l) for l in body[latch:hdr]]
    seq = [s for s in seq if s]
    live_out = set()
    for _ in range(3):
        live = set(live_out)
        prof = [0] * len(seq)
        for i in range(len(seq) - 1, -1, -1):
            op, d, u = seq[i]
            live -= set(d)
            live |= set(u)
            prof[i] = len(live)
        live_out = set(live)
    peak = max(range(len(seq)), key=lambda i: prof[i])
    print(f"{sym}: loop of {len(seq)} instructions, {sum(1 for s in seq if s[0].startswith('v_'))} VALU; loop-carried live VGPRs {len(live_out)}; peak {prof[peak]} at #{peak} ({seq[peak][0]})")
    for i in range(0, len(seq), stride):
        j = max(range(i, min(i + stride, len(seq))), key=lambda t: prof[t])
        print(f"  #{i:5d}  live {prof[i]:3d}   max in block {prof[j]:3d} at #{j:5d}  {seq[j][0]}")
    return seq, prof


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 100)
