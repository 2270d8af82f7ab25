#!/usr/bin/env python
"""Per-layer micro-benchmark of the conv GEMM kernel over tile configurations (TDN_GEMM_CFG), R50-FPN shapes of
SURVEY Appendix A at per-GPU batch B.  Prints TFLOP/s per (shape, config) and checks every config against
config 0 bit for bit (same K order => identical results; a mismatch means a pipeline race)."""
import argparse
import os

# alternate tiles / ablation and cycle-stamp builds live in libtdn_trace.so (make -C torch_detection_amd/csrc TRACE=1)
os.environ.setdefault("TDN_LIB", "libtdn_trace.so")
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch_detection_amd import ops  # noqa: E402

SHAPES = [  # name, Cin, Cout, k, s, Hin, Win, count in R50-FPN
    ("l1.c1 64>64 1x1", 64, 64, 1, 1, 200, 336, 1), ("l1.c2 64>64 3x3", 64, 64, 3, 1, 200, 336, 3),
    ("l1.c3 64>256 1x1", 64, 256, 1, 1, 200, 336, 4), ("l1.c1 256>64 1x1", 256, 64, 1, 1, 200, 336, 2),
    ("l2.0.c1 256>128", 256, 128, 1, 1, 200, 336, 1), ("l2.0.c2 3x3s2", 128, 128, 3, 2, 200, 336, 1),
    ("l2.c3 128>512", 128, 512, 1, 1, 100, 168, 4), ("l2.ds 256>512 s2", 256, 512, 1, 2, 200, 336, 1),
    ("l2.c1 512>128", 512, 128, 1, 1, 100, 168, 3), ("l2.c2 128 3x3", 128, 128, 3, 1, 100, 168, 3),
    ("l3.0.c1 512>256", 512, 256, 1, 1, 100, 168, 1), ("l3.0.c2 3x3s2", 256, 256, 3, 2, 100, 168, 1),
    ("l3.c3 256>1024", 256, 1024, 1, 1, 50, 84, 6), ("l3.ds 512>1024 s2", 512, 1024, 1, 2, 100, 168, 1),
    ("l3.c1 1024>256", 1024, 256, 1, 1, 50, 84, 5), ("l3.c2 256 3x3", 256, 256, 3, 1, 50, 84, 5),
    ("l4.0.c1 1024>512", 1024, 512, 1, 1, 50, 84, 1), ("l4.0.c2 3x3s2", 512, 512, 3, 2, 50, 84, 1),
    ("l4.c3 512>2048", 512, 2048, 1, 1, 25, 42, 3), ("l4.ds 1024>2048 s2", 1024, 2048, 1, 2, 50, 84, 1),
    ("l4.c1 2048>512", 2048, 512, 1, 1, 25, 42, 2), ("l4.c2 512 3x3", 512, 512, 3, 1, 25, 42, 2),
    ("fpn.lat0 256>256", 256, 256, 1, 1, 200, 336, 1), ("fpn.lat1 512>256", 512, 256, 1, 1, 100, 168, 1),
    ("fpn.lat2 1024>256", 1024, 256, 1, 1, 50, 84, 1), ("fpn.lat3 2048>256", 2048, 256, 1, 1, 25, 42, 1),
    ("fpn.out0 3x3", 256, 256, 3, 1, 200, 336, 1), ("fpn.out1 3x3", 256, 256, 3, 1, 100, 168, 1),
    ("fpn.out2 3x3", 256, 256, 3, 1, 50, 84, 1), ("fpn.out3 3x3", 256, 256, 3, 1, 25, 42, 1),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--cfgs", default="0,1,2,3,4,5,7")
    ap.add_argument("--cfgs64", default="0,6")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--mode", default="fwd", choices=["fwd", "dgrad"])
    ap.add_argument("--filter", default="")
    ap.add_argument("--allow-mismatch", action="store_true", help="time ablation configs that compute garbage")
    args = ap.parse_args()
    B = args.batch
    dev = "cuda"
    tot = {}
    print("%-22s %9s | cfg:us(TF/s) ..." % ("shape", "GFLOP"))
    for name, cin, cout, k, s, H, W, cnt in SHAPES:
        if args.filter and args.filter not in name:
            continue
        x = (torch.randn(B, H, W, cin, device=dev)).bfloat16()
        w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).bfloat16()
        Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
        gflop = 2.0 * B * Ho * Wo * cout * cin * k * k / 1e9
        ngemm = cout if args.mode == "fwd" else cin
        cfgs = [int(c) for c in (args.cfgs if ngemm % 128 == 0 else args.cfgs64).split(",")]
        if args.mode == "dgrad":
            g = torch.randn(B, Ho, Wo, cout, device=dev).bfloat16()
            wd = (torch.randn(cin, k, k, cout, device=dev) * 0.05).bfloat16()
            fn = lambda: ops.conv2d_dgrad(g, wd, (H, W), k, s, k // 2)  # noqa: E731
        else:
            fn = lambda: ops.conv2d_fwd(x, w, k, s, k // 2, relu=True)  # noqa: E731
        ref = None
        cells = []
        best = (1e30, -1)
        for c in cfgs:
            os.environ["TDN_GEMM_CFG"] = str(c)
            y = fn()
            if ref is None:
                ref = y.clone()
            elif not torch.equal(ref, y):
                cells.append("%d:MISMATCH" % c)
                if not args.allow_mismatch:
                    continue
            us = timeit(fn, args.iters)
            cells.append("%d:%.0f(%.0f)" % (c, us, gflop / us * 1e3))
            if us < best[0]:
                best = (us, c)
        os.environ.pop("TDN_GEMM_CFG", None)
        us_auto = timeit(fn, args.iters)
        tot["auto"] = tot.get("auto", 0) + us_auto * cnt
        tot["best"] = tot.get("best", 0) + best[0] * cnt
        print("%-22s %9.2f | %s | auto:%.0f best:%d" % (name, gflop, "  ".join(cells), us_auto, best[1]))
    print("weighted totals (us per step, x layer count):", {k: round(v) for k, v in tot.items()})


if __name__ == "__main__":
    main()
