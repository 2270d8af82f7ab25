#!/usr/bin/env python
"""Copy the summaries written by scripts/final_artifacts.sh (gpurun_out/final) into profiles/ under this round's prefix.

  python scripts/collect_profiles.py r04
"""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1]


def cp(src, name):
    src = os.path.join(SRC, src)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(DST, "%s_%s" % (tag, name)))
        print("%-40s -> profiles/%s_%s" % (os.path.relpath(src, ROOT), tag, name))
    else:
        print("missing: %s" % src)


def biggest_stats(sub):
    """The newest kernel_stats.csv under gpurun_out/final/<sub> (gpurun merges runs into one directory: older runs' files
    stay behind; one profiled process per run)."""
    files = glob.glob(os.path.join(SRC, sub, "**", "*_kernel_stats.csv"), recursive=True)
    return os.path.relpath(max(files, key=os.path.getmtime), SRC) if files else "none"


for src, name in (
        ("bench.json", "bench.json"), ("prof_bench.json", "bench_profiled_run.json"),
        (biggest_stats("prof"), "bench_kernel_stats.csv"), ("timeline.txt", "step_timeline.txt"),
        ("pmc_summary.txt", "pmc_dominant_kernel.txt"), ("pmc816_summary.txt", "pmc_halo_8x16_patches.txt"),
        ("box_bench.jsonl", "box_elementwise_bench.jsonl"), (biggest_stats("box"), "box_kernel_stats.csv"),
        ("bench_nograph.json", "bench_nograph_executor.json"), ("bench_c5.json", "bench_c5_r101_f16.json"),
        ("bench_nohalo.json", "bench_generic_kernel_only.json"),
        ("convbench_fwd.log", "convbench_fwd.log"), ("convbench_dgrad.log", "convbench_dgrad.log"),
        ("convbench_fwd_b1.log", "convbench_fwd_b1.log"), ("ksweep_l3.log", "ksweep_l3.log"),
        ("ksweep_p2.log", "ksweep_p2.log"), ("halo_ablate.log", "halo_ablations.log"),
        ("wgrad_group_bench.log", "wgrad_group_bench.log"), ("block_bench.log", "block_bench.log")):
    cp(src, name)
# one-line files gathered into jsonl tables
for name, files in (("batch_dependence.jsonl", ["bench_b1.json", "bench_b2.json", "bench_b4.json"]),
                    ("block_fusion_ab.jsonl", ["bench_fuse0.json", "bench_fuse64.json", "bench_fuse1.json", "bench_head0.json",
                                               "bench_bits0.json"])):
    lines = []
    for f in files:
        p = os.path.join(SRC, f)
        if os.path.exists(p):
            txt = open(p).read().strip().splitlines()
            if txt:
                lines.append('{"run": "%s", "line": %s}' % (f[:-5], txt[-1]))
    if lines:
        open(os.path.join(DST, "%s_%s" % (tag, name)), "w").write("\n".join(lines) + "\n")
        print("%d lines -> profiles/%s_%s" % (len(lines), tag, name))
