#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes written by scripts/pmc_bench.sh (newest run of each pass) per kernel.

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md) -> HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
"""
import collections
import csv
import glob
import os
import sys

ROOT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_bench"
import re


def family(name):
    """Kernel symbol without return type / argument list; template arguments kept (they name the tile)."""
    return re.sub(r"^void ", "", name).split("(")[0]


# tagged dominant launch first (bench.py brackets it with HIP events; TDN_TAG_DOMINANT gives it a symbol of its own:
# conv_halo_kernel<..., 9, 0> or conv_gemm_kernel<192, 256, 64, 2, 4, 2, 6, 1, ...>), then every kernel by device time
def is_tagged(fam):
    return bool(re.match(r"conv_halo_kernel<.*, 9, 0>$", fam) or re.match(r"conv_gemm_kernel<192, 256, 64, 2, 4, 2, 6, 1,", fam))


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(ROOT, "*", ""))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    f = max(files, key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        agg[family(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))


def weight(fam):
    v = agg[fam].get("GRBM_GUI_ACTIVE", [])
    return sum(v)


WANT = sorted(agg, key=lambda f_: (not is_tagged(f_), -weight(f_)))[:14]

print("# rocprofv3 --pmc passes (separate runs per counter group, kernel-trace only) of:")
print("#   python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph   (scripts/pmc_bench.sh)")
print("# one MI355X, R50-FPN fwd+bwd, 2 x 3x800x1344 per step. Values are per launch (mean over launches).")
print("# The first entry is the dominant launch (3x3 256->256, M = 134400: neck.fpn_convs.0 forward and its dgrad):")
print("# the launches bench.py brackets with HIP events run under their own symbol (TDN_TAG_DOMINANT).  Then every")
print("# kernel symbol by summed device time (GRBM_GUI_ACTIVE), pooled over its launches.")
print("# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads")
print("# (MI355X_MICROARCH.md) -> HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.")
for w in WANT:
    d = agg.get(w)
    if not d:
        continue
    print()
    print(w)
    for c, v in sorted(d.items()):
        print("   %-28s mean %.6g  min %.6g  max %.6g  launches %d" % (c, sum(v) / len(v), min(v), max(v), len(v)))
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        print("   -> HBM bytes per launch: corrected %.4g (2*FETCH + WRITE), uncorrected %.4g" %
              ((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024, (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        print("   -> MFMA pipe utilisation: %.3f  (MFMA_BUSY / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs))" %
              (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        print("   -> L2 hit rate: %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
