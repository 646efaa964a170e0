#!/usr/bin/env python3
"""Instructions of a kernel's substep loop attributed to source lines (listing compiled with -gline-tables-only, .loc directives).

usage: asm_lines.py <listing.s> <substring of the mangled kernel name> [file substring]
Counts the instructions of the largest floating-point loop per (file, line) of the innermost .loc and prints them in line order with
the source text, plus totals per 'section' comment is left to the reader."""
import collections
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from asm_hist import kernel_lines, largest_loop  # noqa: E402


def main():
    path, key = sys.argv[1], sys.argv[2]
    files = {}
    for ln in open(path):
        m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2))
    lines = kernel_lines(path, key)
    body = largest_loop(lines)
    cur = None
    cnt = collections.Counter()
    for ln in body:
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if re.match(r"\s+[vs]_|\s+ds_|\s+global_|\s+buffer_", ln):
            cnt[cur] += 1
    src_cache = {}
    total = 0
    for (f, l), c in sorted(cnt.items(), key=lambda kv: (files.get(kv[0][0], ""), kv[0][1])):
        name = files.get(f, "?")
        if name not in src_cache:
            try:
                src_cache[name] = open(name if name.startswith("/") else "/root/repo/quadruped-gym_amd/csrc/" + name).read().splitlines()
            except OSError:
                src_cache[name] = []
        text = src_cache[name][l - 1].strip()[:110] if 0 < l <= len(src_cache[name]) else ""
        total += c
        print(f"{c:5d}  {name.split('/')[-1]}:{l:<5d} {text}")
    print("total", total)


if __name__ == "__main__":
    main()
