#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel name.

    python tools/pmc_summary.py <dir-or-csv> [<dir-or-csv> ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(paths):
    out = []
    for p in paths:
        if os.path.isdir(p):
            out += glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
        else:
            out.append(p)
    return out


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in find(sys.argv[1:]):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?")
                name = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                c = row.get("Counter_Name")
                v = float(row.get("Counter_Value", 0) or 0)
                a = acc[name][c]
                a[0] += v
                a[1] += 1
    counters = sorted({c for k in acc for c in acc[k]})
    print("kernel," + ",".join(["dispatches"] + counters))
    for k in sorted(acc, key=lambda k: -sum(v[0] for v in acc[k].values())):
        n = max(v[1] for v in acc[k].values())
        print(k + "," + ",".join([str(n)] + [f"{acc[k][c][0] / max(acc[k][c][1], 1):.6g}" for c in counters]))


if __name__ == "__main__":
    main()
