#!/usr/bin/env python
"""In-kernel timeline of the conv GEMM kernel: runs one conv with a tracing build (TDN_GEMM_CFG 16..19) and prints,
per workgroup population, the shader-clock cost of set-up, prologue issue, first-data latency, steady-state K-step,
loop tail and epilogue, plus the launch ramp (spread of workgroup start / end stamps)."""
import argparse
import os

# alternate tiles / ablation and cycle-stamp builds live in libtdn_trace.so (make -C torch_detection_amd/csrc TRACE=1)
os.environ.setdefault("TDN_LIB", "libtdn_trace.so")
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from torch_detection_amd import _lib, ops  # noqa: E402
from conv_bench import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--cfgs", default="16,17,18")
    ap.add_argument("--filter", default="l4.c2")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "dgrad"])
    ap.add_argument("--shape", default="", help="cin,cout,k,stride,H,W instead of a named R50-FPN shape")
    args = ap.parse_args()
    lib = _lib.load()
    B = args.batch
    dev = "cuda"
    buf = torch.zeros(1 << 20, 32, dtype=torch.int64, device=dev)
    _lib.check(lib.tdn_debug_trace(buf.data_ptr(), buf.numel() * 8), "tdn_debug_trace")
    shapes = SHAPES
    if args.shape:
        v = [int(t) for t in args.shape.split(",")]
        shapes = [("custom %s" % args.shape, v[0], v[1], v[2], v[3], v[4], v[5], 1)]
    for name, cin, cout, k, s, H, W, cnt in shapes:
        if not args.shape and args.filter not in name:
            continue
        x = torch.randn(B, H, W, cin, device=dev).bfloat16()
        w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).bfloat16()
        Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
        if args.mode == "dgrad":
            g = torch.randn(B, Ho, Wo, cout, device=dev).bfloat16()
            wd = (torch.randn(cin, k, k, cout, device=dev) * 0.05).bfloat16()
            fn = lambda: ops.conv2d_dgrad(g, wd, (H, W), k, s, k // 2)  # noqa: E731
        else:
            fn = lambda: ops.conv2d_fwd(x, w, k, s, k // 2, relu=True)  # noqa: E731
        for cfg in [int(c) for c in args.cfgs.split(",")]:
            os.environ["TDN_GEMM_CFG"] = str(cfg)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            buf.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3
            t = buf.cpu().numpy().astype(np.int64)
            live = t[:, 28] != 0
            t = t[live]
            if len(t) == 0:
                print(name, cfg, "no trace rows (config not a tracing build?)")
                continue
            T = int(t[0, 30])
            nrec = min(T, 24)
            t0 = t[:, 0].min()
            start = t[:, 0] - t0
            end = t[:, 28] - t0
            setup = t[:, 1] - t[:, 0]
            kernarg = t[:, 31] - t[:, 0]
            pro = t[:, 2] - t[:, 1]
            first = t[:, 3] - t[:, 2]
            steps = np.diff(t[:, 3:3 + nrec], axis=1) if nrec > 1 else np.zeros((len(t), 1))
            tail = t[:, 27] - t[:, 3 + nrec - 1]
            epi = t[:, 28] - t[:, 27]
            total = t[:, 28] - t[:, 0]
            xcc = (t[:, 29] >> 32) & 0xf
            span = end.max()
            med = lambda a: float(np.median(a))  # noqa: E731
            print("%s cfg %d: %.1f us by events | %d workgroups, T=%d K-steps | kernel span %d cyc "
                  "(=> %.2f GHz if span == event time)" % (name, cfg, us, len(t), T, span, span / us / 1e3))
            print("   start stamp: p50 %d p90 %d max %d | end stamp: p10 %d p50 %d max %d" % (
                med(start), np.percentile(start, 90), start.max(), np.percentile(end, 10), med(end), end.max()))
            print("   per workgroup (median cycles): kernarg %d | setup %d | prologue issue %d | first data %d | K-step %d "
                  "(p10 %d p90 %d; first 4: %s) | tail(%d steps) %d | epilogue %d | total %d" % (
                      med(kernarg), med(setup), med(pro), med(first), med(steps), np.percentile(steps, 10),
                      np.percentile(steps, 90), np.median(steps[:, :4], axis=0).astype(int).tolist(), T - nrec,
                      med(tail), med(epi), med(total)))
            print("   workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
    os.environ.pop("TDN_GEMM_CFG", None)
    lib.tdn_debug_trace(None, 0)


if __name__ == "__main__":
    main()
