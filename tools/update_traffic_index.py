#!/usr/bin/env python3
"""Fold a condensed profile (tools/summarize_profile.py -> profiles/<round>/<tag>_pmc.json) into profiles/traffic_index.json, the file
bench.py reads for roofline.traffic / valu_issue / fp32 -- stamped with the build id (hash of the library's sources) the profile
was taken on, so that bench.py can tell a stale entry (roofline.profile_stale).

usage: update_traffic_index.py <pmc.json> <mapping> <n_envs> <frame_skip> <obs_dim> [--flops] [--suffix walking|generic]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, mapping, n, fs, od = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    prof = json.load(open(src))
    rel = os.path.relpath(os.path.abspath(src), ROOT)
    path = os.path.join(ROOT, "profiles", "traffic_index.json")
    idx = json.load(open(path))
    bid = prof.get("build_id")
    ent = {"hbm_bytes_per_launch": None, "valu_insts_per_wave": prof.get("valu_insts_per_wave"), "source": rel, "build_id": bid}
    hb = prof.get("hbm_bytes_per_launch")
    if hb:
        ent.update(hbm_bytes_per_launch=hb["total"], read=hb["read"], write=hb["write"],
                   method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py (data.ctrl write-back on); "
                          "KiB -> bytes; FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B)")
    else:
        ent["method"] = "rocprofv3 --pmc SQ_* pass of bench.py; HBM counters not collected at this size"
    suffix = ""
    if "--suffix" in sys.argv:
        suffix = "_" + sys.argv[sys.argv.index("--suffix") + 1]
    idx[f"{mapping}_n{n}_fs{fs}_obs{od}{suffix}"] = ent
    if "--flops" in sys.argv and "fp32" in prof:
        flops = prof["fp32"]["SQ_INSTS_VALU_FLOPS_FP32"] * 64.0 / n
        idx[f"flops_{mapping}_fs{fs}{suffix}"] = {
            "flops_per_env_step": flops, "source": rel, "build_id": bid,
            "method": "SQ_INSTS_VALU_FLOPS_FP32 (= 2*FMA + ADD + MUL + TRANS wave-instructions, checked against the per-class counters) "
                      "x 64 lanes / envs; EVERY lane is counted, so work the mapping replicates across lanes is included"}
    json.dump(idx, open(path, "w"), indent=1)
    print("updated", f"{mapping}_n{n}_fs{fs}_obs{od}{suffix}", "build", bid)


if __name__ == "__main__":
    main()
