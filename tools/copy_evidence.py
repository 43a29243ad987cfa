#!/usr/bin/env python3
"""Copy the summaries tools/collect_evidence.sh produced (scratch, under gpurun_out/) into profiles/ (tracked).

    python tools/copy_evidence.py gpurun_out/ev_r2 round2
"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "profiles")
    for tool in ("bench_c3", "bench_c5", "bench_neighbours", "bench_small", "bench_tilevit", "bench_regions"):
        for f in glob.glob(os.path.join(src, tool, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copyfile(f, os.path.join(dst, f"{tag}_{tool}_kernel_stats.csv"))
        log = os.path.join(src, tool + ".log")
        if os.path.exists(log):
            keep = [l for l in open(log, errors="replace") if not any(s in l for s in ("amdgpu.ids", "rocprofv3]", "output_stream.cpp", "simple_timer.cpp"))]
            open(os.path.join(dst, f"{tag}_{tool}.txt"), "w").writelines(keep)
    for name in ("bench_c2.json", "bench_c3.json", "bench_c4_1gpu.json", "bench_c5.json", "bench_tilevit.json", "bench_c4_nccl_world1.json",
                 "bench_c5_nccl_world1.json", "bench_gloo_2ranks_1gpu.json", "bench_gloo_2ranks_1gpu_c5.json",
                 "bench_from_host.log", "gemm_stamps.log", "attn_stamps.log", "attn_ab.log", "k1_ab.log", "tattn_ablations.log", "host_copy_probe.log"):
        f = os.path.join(src, name)
        if os.path.exists(f):
            lines = [l for l in open(f, errors="replace") if "amdgpu.ids" not in l]
            if name.endswith(".json"):
                lines = [l for l in lines if l.lstrip().startswith("{")]
            open(os.path.join(dst, f"{tag}_{name.replace('.log', '.txt')}"), "w").writelines(lines)
    print(sorted(f for f in os.listdir(dst) if f.startswith(tag)))


if __name__ == "__main__":
    main()
