"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_gpu.sh) into small committed files:
profiles/<round>/<tag>_kernel_stats.csv (kernel names shortened) and <tag>_pmc.json (per-launch means
for the step kernel, HBM bytes corrected as MI355X_MICROARCH.md prescribes)."""
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.split("(")[0]
    return name if len(name) < 80 else name[:77] + "..."


def main(src, dst_prefix):
    os.makedirs(os.path.dirname(dst_prefix), exist_ok=True)
    stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if stats:
        rows = list(csv.reader(open(stats[-1])))            # the newest run (gpurun merges runs into the same directory)
        with open(dst_prefix + "_kernel_stats.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(rows[0])
            for r in rows[1:]:
                r[0] = short(r[0])
                w.writerow(r)
    pmc = {}
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_flops"):
        paths = sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
        for path in paths[-1:]:                               # the newest run only
            acc = {}
            for row in csv.DictReader(open(path)):
                if "qg_step" in row["Kernel_Name"]:
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for k, v in acc.items():
                pmc[k] = sum(v) / len(v)
    out = {"per_launch_mean": pmc}
    bid = os.path.join(src, "build_id.txt")
    if os.path.exists(bid):                                   # hash of the library's sources at profiling time (tools/profile_gpu.sh)
        out["build_id"] = open(bid).read().strip()
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at
        # 64 B, i.e. reads half the bytes of a wide coalesced stream -> doubled (MI355X_MICROARCH.md, HBM)
        out["hbm_bytes_per_launch"] = {"read": 2 * pmc["FETCH_SIZE"] * 1024, "write": pmc["WRITE_SIZE"] * 1024,
                                       "total": (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024,
                                       "correction": "FETCH_SIZE x2 (gfx950), KiB -> bytes"}
    if "SQ_INSTS_VALU_FLOPS_FP32" in pmc:
        out["fp32"] = {k: pmc[k] for k in pmc if k.startswith("SQ_INSTS_VALU_") and ("F32" in k or "FP32" in k)}
    if "SQ_INSTS_VALU" in pmc and "SQ_WAVES" in pmc:
        out["valu_insts_per_wave"] = pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"]
    json.dump(out, open(dst_prefix + "_pmc.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
