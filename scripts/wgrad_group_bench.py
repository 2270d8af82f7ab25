#!/usr/bin/env python
"""Grouped weight-gradient launches (tdn_wgrad_group) over the R50-FPN layers, stage by stage, as the backward
schedule issues them: per stage one group of the nine-tap-eligible 3x3 convs and one group of the rest.

  --plan        host only (no GPU): print the decomposition per stage (splits, direct members, workgroups, slab bytes)
  default       time every stage group (HIP events, `--iters` launches) and the sum per step; with --check the
                gradients are compared with the TDN_WGRAD_T=<huge> (no split-K) run of the same kernels (rel-L2)
Environment knobs of the planner (csrc/conv_wgrad.hip: plan_group) can be swept with --sweep "NAME=v1,v2;NAME2=..".
"""
import argparse
import ctypes
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

STAGE_OF = {"fpn": "fpn", "l4": "l4", "l3": "l3", "l2": "l2", "l1": "l1"}
ORDER = ["fpn", "l4", "l3", "l2", "l1", "stem"]


def stage_shapes(depth=50):
    from conv_bench import SHAPES
    st = {k: [] for k in ORDER}
    for name, cin, cout, k, s, H, W, cnt in SHAPES:
        if depth == 101 and name.startswith("l3.c"):
            cnt += 17
        st[name.split(".")[0]] += [(name, cin, cout, k, s, H, W)] * cnt
    st["stem"] = [("stem 7x7", 3, 64, 7, 2, 800, 1344)]
    return st


def t9_eligible(cin, cout, k, s, W):
    return k == 3 and s == 1 and cout % 64 == 0 and cin % 64 == 0 and W >= 8 and \
        (cout % 128 == 0 or os.environ.get('TDN_WGRAD9_64', '1') != '0')


def fake_items(shapes, B, bn=True):
    from torch_detection_amd import ops, _lib
    items = []
    for name, cin, cout, k, s, H, W in shapes:
        it = _lib.WgradItem()
        for f in ("x", "g", "w_fwd", "dw", "dbeta"):
            setattr(it, f, 256)
        if bn and not name.startswith("fpn"):
            it.scale = it.mean = it.invstd = it.dgamma = 256
        it.kind = ops.WGRAD_STEM if k == 7 else ops.WGRAD_CONV
        it.N, it.H, it.W, it.Cin, it.Cout, it.k, it.stride, it.pad, it.groups = B, H, W, cin, cout, k, s, k // 2, 1
        items.append(it)
    return items


def split_groups(shapes):
    heavy = [sh for sh in shapes if t9_eligible(sh[1], sh[2], sh[3], sh[4], sh[6])]
    rest = [sh for sh in shapes if not t9_eligible(sh[1], sh[2], sh[3], sh[4], sh[6])]
    return [g for g in (heavy, rest) if g]


def show_plan(B, depth):
    from torch_detection_amd import ops
    st = stage_shapes(depth)
    tot_slab = tot_wg = tot_l = tot_f = 0
    for stage in ORDER:
        for grp in split_groups(st[stage]):
            per, tot = ops.wgrad_group_plan(fake_items(grp, B))
            tot_l += tot[0]; tot_f += tot[1]; tot_wg += tot[2]; tot_slab += tot[3]
            print("== %s: %d members, %d gradient launches + %d finalize, %d workgroups, %.1f MB slabs" %
                  (stage, len(grp), tot[0], tot[1], tot[2], tot[3] / 1024.0))
            seen = set()
            for sh, p in zip(grp, per):
                key = (sh[0], tuple(p))
                if key in seen:
                    continue
                seen.add(key)
                print("   %-20s %s %3dx%-3d splits %3d x %6d px  direct %d  wgs %5d  slab %7.1f MB" %
                      (sh[0], "T9 " if p[0] else "tap", p[1], p[2], p[3], p[4], p[5], p[6], p[7] / 1024.0))
    print("TOTAL per step: %d gradient launches, %d finalize launches, %d workgroups, %.1f MB of fp32 slabs "
          "(written once, read once)" % (tot_l, tot_f, tot_wg, tot_slab / 1024.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--plan", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--stages", default=",".join(ORDER))
    ap.add_argument("--sweep", default="")
    args = ap.parse_args()
    if args.plan:
        show_plan(args.batch, args.depth)
        return
    import torch
    from torch_detection_amd import ops
    from conv_bench import timeit
    B = args.batch
    st = stage_shapes(args.depth)
    dev = torch.device("cuda")
    torch.manual_seed(0)

    def build(shapes):
        items, keep, outs = [], [], []
        for name, cin, cout, k, s, H, W in shapes:
            bn = not name.startswith("fpn")
            if k == 7:
                x = (torch.randn(B, H + 6, W + 8, 4, device=dev)).bfloat16()
                g = (torch.randn(B, H // 2, W // 2, cout, device=dev) * 0.1).bfloat16()
                w = (torch.randn(cout, 7, 8, 4, device=dev) * 0.05).bfloat16()
            else:
                x = torch.randn(B, H, W, cin, device=dev).bfloat16()
                Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
                g = (torch.randn(B, Ho, Wo, cout, device=dev) * 0.1).bfloat16()
                w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).bfloat16()
            sc = (torch.rand(cout, device=dev) + 0.5) if bn else None
            mean = torch.randn(cout, device=dev) * 0.1 if bn else None
            inv = (torch.rand(cout, device=dev) + 0.5) if bn else None
            if k == 7:
                it, dw, dg, db = ops.stem_conv_wgrad_item(x, g, w, (H, W), sc, mean, inv)
            else:
                it, dw, dg, db = ops.conv2d_wgrad_item(x, g, w, k, s, k // 2, sc, mean, inv)
            items.append(it)
            keep += [x, g, w, sc, mean, inv]
            outs.append((dw, dg, db))
        return items, keep, outs

    def snapshot(outs):
        return [tuple(t.clone() if t is not None else None for t in o) for o in outs]

    def rel(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / (b.norm() + 1e-30))

    combos = [{}]
    if args.sweep:
        axes = []
        for part in args.sweep.split(";"):
            name, vals = part.split("=")
            axes.append([(name, v) for v in vals.split(",")])
        combos = [dict(c) for c in itertools.product(*axes)]
    built = {}
    for stage in args.stages.split(","):
        built[stage] = [build(grp) + (grp,) for grp in split_groups(st[stage])]
    refs = {}
    if args.check:
        os.environ["TDN_WGRAD_T"] = "100000"
        for stage, groups in built.items():
            for gi, (items, keep, outs, grp) in enumerate(groups):
                ops.wgrad_group(items, torch.bfloat16, dev)
                torch.cuda.synchronize()
                refs[(stage, gi)] = snapshot(outs)
        os.environ.pop("TDN_WGRAD_T")
    for combo in combos:
        for k_, v_ in combo.items():
            os.environ[k_] = v_
        total = 0.0
        cells = []
        for stage, groups in built.items():
            for gi, (items, keep, outs, grp) in enumerate(groups):
                us = timeit(lambda: ops.wgrad_group(items, torch.bfloat16, dev), args.iters)
                gflop = sum(2.0 * B * ops.conv_out_size(H, k, s, k // 2) * ops.conv_out_size(W, k, s, k // 2) * cout *
                            cin * k * k for _, cin, cout, k, s, H, W in grp) / 1e9
                total += us
                tag = "%s.%s" % (stage, "T9" if t9_eligible(*[grp[0][i] for i in (1, 2, 3, 4, 6)]) else "tap")
                cell = "%s %.0fus %.0fTF" % (tag, us, gflop / us * 1e3)
                if args.check:
                    worst = 0.0
                    for o, r in zip(outs, refs[(stage, gi)]):
                        for a, b in zip(o, r):
                            if a is not None:
                                worst = max(worst, rel(a, b))
                    cell += " err%.1e" % worst
                    # run-to-run reproducibility
                    s1 = snapshot(outs)
                    ops.wgrad_group(items, torch.bfloat16, dev)
                    torch.cuda.synchronize()
                    same = all(torch.equal(a, b) for o, r in zip(outs, s1) for a, b in zip(o, r) if a is not None)
                    cell += " repro" if same else " NONREPRO"
                cells.append(cell)
        print("%s -> %.0f us/step | %s" % (combo or "default", total, " | ".join(cells)), flush=True)
        for k_ in combo:
            os.environ.pop(k_)


if __name__ == "__main__":
    main()
