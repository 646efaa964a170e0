"""Summarise a rocprofv3 --pmc counter_collection.csv: per-kernel mean of each counter (per dispatch)."""
import csv
import collections
import sys


def main(path, kernel_filter="qg_step"):
    acc = collections.defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if kernel_filter in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, vals in sorted(acc.items()):
        print(f"{name:28s} n={len(vals):4d} mean={sum(vals) / len(vals):16.1f}")


if __name__ == "__main__":
    main(*sys.argv[1:])
