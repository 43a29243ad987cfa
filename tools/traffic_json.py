#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into the per-launch traffic record
bench.py reports as roofline.traffic.

    python tools/traffic_json.py <fetch-dir> <write-dir> <out.json>

Counters are KiB per dispatch.  On gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so the read side is doubled; WRITE_SIZE
is exact for the 16-byte-per-lane stores these kernels use.  Both count requests that leave the
XCD's L2, including those the 256 MiB Infinity Cache then serves.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                acc[name][0] += float(row["Counter_Value"])
                acc[name][1] += 1
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"unit": "bytes per launch", "read_correction": "2 x FETCH_SIZE", "kernels": {}}
    gemm_bytes = gemm_launches = 0.0
    for k in sorted(set(fetch) | set(write)):
        n = max(fetch[k][1], write[k][1])
        if n == 0:
            continue
        rd = 2.0 * 1024.0 * fetch[k][0] / max(fetch[k][1], 1)
        wr = 1024.0 * write[k][0] / max(write[k][1], 1)
        out["kernels"][k] = {"dispatches": n, "read": rd, "write": wr, "traffic": rd + wr}
        if k.startswith("gemm_bf16_tn"):
            gemm_bytes += (rd + wr) * n
            gemm_launches += n
    out["gemm_launches"] = gemm_launches
    out["gemm_traffic_per_launch"] = gemm_bytes / gemm_launches if gemm_launches else None
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("gemm_launches", "gemm_traffic_per_launch")}))


if __name__ == "__main__":
    main()
