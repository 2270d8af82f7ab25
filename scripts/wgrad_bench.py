#!/usr/bin/env python
"""Per-layer micro-benchmark of the weight-gradient path (wgrad GEMM + finalize) over tile / split-K settings."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from torch_detection_amd import ops  # noqa: E402
from conv_bench import SHAPES, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--tiles", default="128x64,128x128,64x64,64x128")
    ap.add_argument("--filter", default="")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    B = args.batch
    tot = {}
    for name, cin, cout, k, s, H, W, cnt in SHAPES:
        if args.filter and args.filter not in name:
            continue
        x = torch.randn(B, H, W, cin, device="cuda").bfloat16()
        Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
        g = torch.randn(B, Ho, Wo, cout, device="cuda").bfloat16()
        w = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).bfloat16()
        gflop = 2.0 * B * Ho * Wo * cout * cin * k * k / 1e9
        cells = []
        best = 1e30
        for tile in args.tiles.split(","):
            os.environ["TDN_WGRAD_TILE"] = tile
            fn = lambda: ops.conv2d_wgrad(x, g, w, k, s, k // 2)  # noqa: E731
            us = timeit(fn, args.iters)
            cells.append("%s:%.0f(%.0f)" % (tile, us, gflop / us * 1e3))
            best = min(best, us)
        os.environ.pop("TDN_WGRAD_TILE", None)
        os.environ["TDN_WGRAD9"] = "0"
        us_t1 = timeit(lambda: ops.conv2d_wgrad(x, g, w, k, s, k // 2), args.iters)
        os.environ["TDN_WGRAD9"] = "1"
        us_t9 = timeit(lambda: ops.conv2d_wgrad(x, g, w, k, s, k // 2), args.iters)
        os.environ.pop("TDN_WGRAD9", None)
        cells.append("tap-per-tile:%.0f nine-tap:%.0f" % (us_t1, us_t9))
        tot["tap-per-tile"] = tot.get("tap-per-tile", 0) + us_t1 * cnt
        us_auto = timeit(lambda: ops.conv2d_wgrad(x, g, w, k, s, k // 2), args.iters)
        tot["auto"] = tot.get("auto", 0) + us_auto * cnt
        tot["best"] = tot.get("best", 0) + best * cnt
        print("%-22s %8.2f | %s | auto:%.0f" % (name, gflop, "  ".join(cells), us_auto))
    print("weighted totals (us per step):", {k: round(v) for k, v in tot.items()})


if __name__ == "__main__":
    main()
