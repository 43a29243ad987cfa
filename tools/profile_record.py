#!/usr/bin/env python3
"""Fold the rocprofv3 passes of ONE bench.py command into profiles/current.json (what bench.py prints as `from_profile`
and uses for `roofline.traffic`) and copy the per-kernel summaries next to it.

    python tools/profile_record.py <dir> <tag> [--crops 4096] [--steps 3]      (steps = timed + warm-up passes in the trace)

<dir> holds the passes, each written by `rocprofv3 ... -d <dir>/<pass> --output-format csv -- python3 bench.py --steps 2 --warmup 1
--headline-only`:
    kt      --kernel-trace --stats
    sq      --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
    grbm    --kernel-trace --pmc GRBM_GUI_ACTIVE
    fetch   --pmc FETCH_SIZE
    write   --pmc WRITE_SIZE
Counters: FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE tallies a 128-byte request as 64 bytes on gfx950, so
reads are doubled (MI355X_MICROARCH.md, HBM).  GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = value / 8 / duration.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): busy matrix-pipe cycles over the SIMD
cycles the chip actually ran at the clock it held -- NOT a fraction of the nominal 2.5 PFLOP/s (that is
`forward_mfma_frac` / `roofline.frac` in the bench line, measured in the run itself).
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_embeddings_amd.build import kernel_source_hash  # noqa: E402


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def counters(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"] or 0)
                a[1] += 1
    return {k: {c: (v[0] / max(v[1], 1), v[1]) for c, v in d.items()} for k, d in acc.items()}


def durations(path):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                a = acc[short(row["Kernel_Name"])]
                a[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                a[1] += 1
    return {k: (v[0] / max(v[1], 1), v[1]) for k, v in acc.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    crops = int(sys.argv[sys.argv.index("--crops") + 1]) if "--crops" in sys.argv else 4096
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3  # passes of the hot path traced: --steps 2 + --warmup 1
    out_dir = os.path.join(ROOT, "profiles")
    kt = durations(os.path.join(src, "kt"))
    sq, grbm = counters(os.path.join(src, "sq")), counters(os.path.join(src, "grbm"))
    grbm_dur = durations(os.path.join(src, "grbm"))
    fetch, write = counters(os.path.join(src, "fetch")), counters(os.path.join(src, "write"))
    kernels = {}
    gemm_bytes = gemm_n = 0.0
    busy_total = avail_total = 0.0
    for k in sorted(kt, key=lambda k: -kt[k][0] * kt[k][1]):
        rec = {"dispatches": kt[k][1], "avg_ms": kt[k][0] / 1e6}
        if k in fetch and k in write and "FETCH_SIZE" in fetch[k] and "WRITE_SIZE" in write[k]:
            rd, wr = 2.0 * 1024.0 * fetch[k]["FETCH_SIZE"][0], 1024.0 * write[k]["WRITE_SIZE"][0]
            rec.update(read_bytes=rd, write_bytes=wr, traffic_bytes=rd + wr)
            if k.startswith("gemm_bf16_tn"):
                gemm_bytes += (rd + wr) * kt[k][1]
                gemm_n += kt[k][1]
        if k in sq and k in grbm and "SQ_VALU_MFMA_BUSY_CYCLES" in sq[k] and "GRBM_GUI_ACTIVE" in grbm[k]:
            busy, gui = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"][0], grbm[k]["GRBM_GUI_ACTIVE"][0]
            rec["mfma_util"] = busy / (gui / 8.0 * 1024.0) if gui else None
            if k in grbm_dur and grbm_dur[k][0] > 0:
                rec["clock_ghz_in_counter_pass"] = gui / 8.0 / grbm_dur[k][0]
            n = kt[k][1]
            busy_total += busy * n
            avail_total += gui / 8.0 * 1024.0 * n
        kernels[k] = rec
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        sha = ""
    gemm_ms_step = sum(v["avg_ms"] * v["dispatches"] for k, v in kernels.items() if k.startswith("gemm_bf16_tn")) / steps
    rec = {
        "what": "separate rocprofv3 passes of `python3 bench.py --steps %d --warmup 1 --headline-only` (kernel trace; SQ; GRBM; FETCH_SIZE; WRITE_SIZE), folded by tools/profile_record.py" % (steps - 1),
        "tag": tag, "git": sha, "source_hash": kernel_source_hash(),  # of csrc/ + mme.h: bench.py marks the record stale when it differs
        "crops_per_gpu": crops, "steps_traced": steps,
        "gemm_traffic_per_launch": gemm_bytes / gemm_n if gemm_n else None,
        "gemm_launches_traced": gemm_n,
        "gemm_ms_per_step_kernel_trace": gemm_ms_step,
        "forward_mfma_util": busy_total / avail_total if avail_total else None,
        "forward_mfma_util_definition": "sum over the forward's kernels of SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): matrix-pipe busy share of the "
                                         "SIMD cycles at the clock the chip HELD in the counter passes (see clock_ghz_in_counter_pass per kernel), not of the nominal 2.5 PFLOP/s",
        "kernels": kernels,
    }
    os.makedirs(out_dir, exist_ok=True)
    json.dump(rec, open(os.path.join(out_dir, "current.json"), "w"), indent=1)
    json.dump(rec, open(os.path.join(out_dir, f"{tag}_profile.json"), "w"), indent=1)
    for f in glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copyfile(f, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    print(json.dumps({k: rec[k] for k in ("tag", "git", "gemm_traffic_per_launch", "gemm_ms_per_step_kernel_trace", "forward_mfma_util")}))
    for k, v in list(kernels.items())[:8]:
        print(f"  {k}: {v}")


if __name__ == "__main__":
    main()
