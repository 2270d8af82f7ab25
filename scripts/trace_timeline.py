#!/usr/bin/env python
"""Timeline of ONE step from a rocprofv3 --kernel-trace CSV of `python bench.py`: where the step's wall time goes.

A step starts at a `stage_image_kernel` dispatch and ends before the next one.  Printed for the chosen step
(default: the last complete one):
  * wall span, time with >= 1 kernel running, idle time, mean number of concurrently running kernels
  * forward / backward split (backward starts at the first dgrad-side kernel after the forward chain: taken as the
    first kernel that starts after the last forward-only kernel `subsample_fwd_kernel`)
  * per kernel symbol: launches, summed duration, share of the step's busy time
  * with --list: every dispatch as (start offset us, duration us, queue, grid, name)
"""
import argparse
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def load(path):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]),
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    return rows


def union_busy(iv):
    iv = sorted(iv)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--step", type=int, default=-2, help="index of the step (python indexing over complete steps)")
    ap.add_argument("--list", action="store_true")
    args = ap.parse_args()
    rows = load(args.trace)
    starts = [i for i, r in enumerate(rows) if "stage_image_kernel" in r[2]]
    if len(starts) < 2:
        sys.exit("need at least two steps in the trace")
    steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    st = steps[args.step]
    t0 = st[0][0]
    t1 = max(r[1] for r in st)
    wall = t1 - t0
    busy = union_busy([(r[0], r[1]) for r in st])
    ksum = sum(r[1] - r[0] for r in st)
    print("step %d of %d: %d dispatches, wall %.1f us, busy %.1f us, idle %.1f us, summed kernel time %.1f us "
          "(mean concurrency %.2f)" % (args.step, len(steps), len(st), wall / 1e3, busy / 1e3, (wall - busy) / 1e3,
                                       ksum / 1e3, ksum / max(1, busy)))
    fwd_end = None
    for r in st:
        if "subsample_fwd_kernel" in r[2]:
            fwd_end = r[1]
    if fwd_end:
        print("forward %.1f us, backward %.1f us" % ((fwd_end - t0) / 1e3, (t1 - fwd_end) / 1e3))
        for tag, sel in (("fwd", [r for r in st if r[1] <= fwd_end]), ("bwd", [r for r in st if r[1] > fwd_end])):
            if sel:
                a, b = min(r[0] for r in sel), max(r[1] for r in sel)
                bz = union_busy([(r[0], r[1]) for r in sel])
                print("  %s: %d dispatches, span %.1f us, busy %.1f, idle %.1f, kernel sum %.1f" %
                      (tag, len(sel), (b - a) / 1e3, bz / 1e3, (b - a - bz) / 1e3, sum(r[1] - r[0] for r in sel) / 1e3))
    agg = {}
    for s, e, n, q, g in st:
        k = short(n)
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    print("%-72s %5s %10s %6s" % ("kernel", "n", "sum us", "share"))
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %5d %10.1f %5.1f%%" % (k, n, d / 1e3, 100.0 * d / ksum))
    if args.list:
        for s, e, n, q, g in st:
            print("%9.1f %8.1f q%-3d wg%-6d %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, g, short(n)))


if __name__ == "__main__":
    main()
