#!/usr/bin/env python
"""Graph-replay timing of the generic conv GEMM kernel's tile / load-path configurations (TDN_GEMM_CFG, trace build)
on the per-image 1x1 layers of R50-FPN (the launches of the layer2-4 chains), with a bit-for-bit check against
configuration 0 where the K order is the same (all but the K-group configurations)."""
import argparse
import os

os.environ.setdefault("TDN_LIB", "libtdn_trace.so")
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch_detection_amd import ops  # noqa: E402

SHAPES = [  # name, Cin, Cout, H, W (one image), launches per step and image (fwd + dgrad of the transposed shape)
    ("l2.c1 512>128", 512, 128, 100, 168), ("l2.c3 128>512", 128, 512, 100, 168),
    ("l3.0.c1 512>256", 512, 256, 100, 168),
    ("l3.c1 1024>256", 1024, 256, 50, 84), ("l3.c3 256>1024", 256, 1024, 50, 84),
    ("l4.0.c1 1024>512", 1024, 512, 50, 84),
    ("l4.c1 2048>512", 2048, 512, 25, 42), ("l4.c3 512>2048", 512, 2048, 25, 42),
    ("fpn.lat3 2048>256", 2048, 256, 25, 42),
]


def graph_time(fn, iters):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="0,25,1,46")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--filter", default="")
    args = ap.parse_args()
    dev = "cuda"
    cfgs = [int(c) for c in args.cfgs.split(",")]
    tot = {}
    for name, cin, cout, H, W in SHAPES:
        if args.filter and args.filter not in name:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(args.batch, H, W, cin, device=dev, generator=g).bfloat16()
        w = (torch.randn(cout, 1, 1, cin, device=dev, generator=g) * 0.05).bfloat16()
        scale = torch.rand(cout, device=dev, generator=g) + 0.5
        shift = torch.randn(cout, device=dev, generator=g) * 0.1
        add = torch.randn(args.batch, H, W, cout, device=dev, generator=g).bfloat16()
        fn = lambda: ops.conv2d_fwd(x, w, 1, 1, 0, relu=True, scale=scale, shift=shift, addend=add,  # noqa: E731
                                    addend_mode=ops.ADD_SAME)
        os.environ["TDN_HALO"] = "0"
        os.environ["TDN_GEMM_CFG"] = "0"
        ref = fn().clone()
        cells = []
        for c in cfgs:
            os.environ["TDN_GEMM_CFG"] = str(c)
            try:
                y = fn()
            except RuntimeError as e:
                cells.append("%d:n/a" % c)
                continue
            same = torch.equal(y, ref)
            us = graph_time(fn, args.iters)
            tot[c] = tot.get(c, 0.0) + us
            cells.append("%d:%.1f%s" % (c, us, "" if same else "~"))
        os.environ.pop("TDN_GEMM_CFG")
        us = graph_time(fn, args.iters)
        tot["auto"] = tot.get("auto", 0.0) + us
        print("%-20s M=%5d | %s | auto:%.1f" % (name, args.batch * H * W, "  ".join(cells), us), flush=True)
    print("sum us:", {k: round(v, 1) for k, v in tot.items()}, " (~ = not bit-identical to configuration 0)")


if __name__ == "__main__":
    main()
