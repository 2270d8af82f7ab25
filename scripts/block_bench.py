#!/usr/bin/env python
"""One-launch Bottleneck kernel (csrc/conv_block.hip) against the three per-conv launches it replaces, forward and
input-gradient chain, graph-replay timing (device time; eager Python launches cost ~10 us of host time each).

  python scripts/block_bench.py [--batch 1] [--H 200 --W 336] [--C 64] [--iters 20]
  python scripts/block_bench.py --head        the head block (layer1.0: 64 input channels, 1x1 downsample branch)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch_detection_amd import ops  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--H", type=int, default=200)
    ap.add_argument("--W", type=int, default=336)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--head", action="store_true")
    a = ap.parse_args()
    if a.head:
        return head(a)
    dev = torch.device("cuda")
    N, H, W, C = a.batch, a.H, a.W, a.C
    C4 = 4 * C
    gen = torch.Generator(device=dev).manual_seed(1)
    dt = torch.bfloat16
    x = torch.relu(torch.randn(N, H, W, C4, device=dev, generator=gen)).to(dt)
    w1 = (torch.randn(C, 1, 1, C4, device=dev, generator=gen) * (2.0 / C4) ** 0.5).to(dt)
    w2 = (torch.randn(C, 3, 3, C, device=dev, generator=gen) * (2.0 / (9 * C)) ** 0.5).to(dt)
    w3 = (torch.randn(C4, 1, 1, C, device=dev, generator=gen) * (2.0 / C) ** 0.5).to(dt)
    aff = [torch.rand(n, device=dev, generator=gen) + 0.5 if i % 2 == 0 else torch.randn(n, device=dev, generator=gen) * 0.1
           for i, n in enumerate((C, C, C, C, C4, C4))]
    w1d = w1.permute(3, 1, 2, 0).contiguous()
    w2d = w2.permute(3, 1, 2, 0).contiguous()
    w3d = w3.permute(3, 1, 2, 0).contiguous()
    h1, h2, out = ops.bottleneck_fwd(x, w1, w2, w3, aff)
    g = torch.where(out > 0, torch.randn(N, H, W, C4, device=dev, generator=gen).to(dt) * 0.1, torch.zeros((), device=dev, dtype=dt))
    g = g.contiguous()
    o1 = torch.empty_like(h1); o2 = torch.empty_like(h2); o3 = torch.empty_like(out)

    def fused_f():
        ops.bottleneck_fwd(x, w1, w2, w3, aff, outs=(o1, o2, o3))

    def sep_f():
        r1 = ops.conv2d_fwd(x, w1, 1, 1, 0, aff[0], aff[1], relu=True, out=o1)
        r2 = ops.conv2d_fwd(r1, w2, 3, 1, 1, aff[2], aff[3], relu=True, out=o2)
        ops.conv2d_fwd(r2, w3, 1, 1, 0, aff[4], aff[5], x, ops.ADD_SAME, True, out=o3)

    bits = ops.bottleneck_bit_planes(N, H, W, C, dev)
    ops.bottleneck_fwd(x, w1, w2, w3, aff, bits=bits)

    def fused_fb():
        ops.bottleneck_fwd(x, w1, w2, w3, aff, outs=(o1, o2, o3), bits=bits)

    def fused_b():
        ops.bottleneck_dgrad(g, w3d, w2d, w1d, (h2, h1, x), outs=(o1, o2, o3))

    def fused_bb():
        ops.bottleneck_dgrad(g, w3d, w2d, w1d, None, outs=(o1, o2, o3), bits=bits)

    def sep_b():
        r2 = ops.conv2d_dgrad(g, w3d, (H, W), 1, 1, 0, mask_src=h2, out=o1)
        r1 = ops.conv2d_dgrad(r2, w2d, (H, W), 3, 1, 1, mask_src=h1, out=o2)
        ops.conv2d_dgrad(r1, w1d, (H, W), 1, 1, 0, g, ops.ADD_SAME, x, out=o3)

    gflop = 2.0 * N * H * W * (C4 * C + 9 * C * C + C * C4) / 1e9
    mb_fused = N * H * W * (C4 + C + C + C4) * 2 / 1e6
    print("bottleneck C=%d, %d x %dx%d: %.2f GFLOP, %.0f MB (x once, h1 / h2 written, out written)" % (C, N, H, W, gflop, mb_fused))
    for name, f, s in (("forward", fused_f, sep_f), ("fwd+bits", fused_fb, sep_f), ("dgrad", fused_b, sep_b),
                       ("dgrad/bits", fused_bb, sep_b)):
        tf = timeit(f, a.iters)
        ts = timeit(s, a.iters)
        print("%-10s one launch %7.1f us (%5.0f TF/s, %4.2f TB/s)   three launches %7.1f us   x%.2f" %
              (name, tf, gflop / tf * 1e3, mb_fused / tf, ts, ts / tf))


def head(a):
    """layer1.0: four per-conv launches (downsample, conv1, conv2, conv3 + residual) against the head-block launch with the
    downsample branch as a launch of its own (1 + 1) and inside the launch (1)."""
    dev = torch.device("cuda")
    N, H, W, C, C4 = a.batch, a.H, a.W, 64, 256
    gen = torch.Generator(device=dev).manual_seed(1)
    dt = torch.bfloat16
    x = torch.relu(torch.randn(N, H, W, C, device=dev, generator=gen)).to(dt)
    w1 = (torch.randn(C, 1, 1, C, device=dev, generator=gen) * (2.0 / C) ** 0.5).to(dt)
    w2 = (torch.randn(C, 3, 3, C, device=dev, generator=gen) * (2.0 / (9 * C)) ** 0.5).to(dt)
    w3 = (torch.randn(C4, 1, 1, C, device=dev, generator=gen) * (2.0 / C) ** 0.5).to(dt)
    wd = (torch.randn(C4, 1, 1, C, device=dev, generator=gen) * (2.0 / C) ** 0.5).to(dt)
    aff = [torch.rand(n, device=dev, generator=gen) + 0.5 if i % 2 == 0 else torch.randn(n, device=dev, generator=gen) * 0.1
           for i, n in enumerate((C, C, C, C, C4, C4, C4, C4))]
    w1d, w2d, w3d, wdd = (w.permute(3, 1, 2, 0).contiguous() for w in (w1, w2, w3, wd))
    h1, h2, out = ops.bottleneck_head_fwd(x, w1, w2, w3, aff[:6], None, down=(wd, aff[6], aff[7]))
    g = torch.where(out > 0, torch.randn(N, H, W, C4, device=dev, generator=gen).to(dt) * 0.1,
                    torch.zeros((), device=dev, dtype=dt)).contiguous()
    o1 = torch.empty_like(h1); o2 = torch.empty_like(h2); o3 = torch.empty_like(out)
    res = torch.empty_like(out); dx = torch.empty_like(x); t = torch.empty_like(x)
    bits = ops.bottleneck_bit_planes(N, H, W, C, dev)[:2]
    ops.bottleneck_head_fwd(x, w1, w2, w3, aff[:6], None, bits=bits, down=(wd, aff[6], aff[7]))

    def sep_f():
        r = ops.conv2d_fwd(x, wd, 1, 1, 0, aff[6], aff[7], relu=False, out=res)
        r1 = ops.conv2d_fwd(x, w1, 1, 1, 0, aff[0], aff[1], relu=True, out=o1)
        r2 = ops.conv2d_fwd(r1, w2, 3, 1, 1, aff[2], aff[3], relu=True, out=o2)
        ops.conv2d_fwd(r2, w3, 1, 1, 0, aff[4], aff[5], r, ops.ADD_SAME, True, out=o3)

    def two_f():
        r = ops.conv2d_fwd(x, wd, 1, 1, 0, aff[6], aff[7], relu=False, out=res)
        ops.bottleneck_head_fwd(x, w1, w2, w3, aff[:6], r, outs=(o1, o2, o3), bits=bits)

    def one_f():
        ops.bottleneck_head_fwd(x, w1, w2, w3, aff[:6], None, outs=(o1, o2, o3), bits=bits, down=(wd, aff[6], aff[7]))

    def sep_b():
        tt = ops.conv2d_dgrad(g, wdd, (H, W), 1, 1, 0, out=t)
        r2 = ops.conv2d_dgrad(g, w3d, (H, W), 1, 1, 0, mask_src=h2, out=o1)
        r1 = ops.conv2d_dgrad(r2, w2d, (H, W), 3, 1, 1, mask_src=h1, out=o2)
        ops.conv2d_dgrad(r1, w1d, (H, W), 1, 1, 0, tt, ops.ADD_SAME, None, out=dx)

    def two_b():
        tt = ops.conv2d_dgrad(g, wdd, (H, W), 1, 1, 0, out=t)
        ops.bottleneck_head_dgrad(g, w3d, w2d, w1d, None, tt, outs=(o1, o2, dx), bits=bits)

    def one_b():
        ops.bottleneck_head_dgrad(g, w3d, w2d, w1d, None, None, outs=(o1, o2, dx), bits=bits, down=wdd)

    gflop = 2.0 * N * H * W * (C * C + 9 * C * C + 2 * C * C4) / 1e9
    print("head block C=64, %d x %dx%d: %.2f GFLOP per pass" % (N, H, W, gflop))
    for name, fs in (("forward", (sep_f, two_f, one_f)), ("dgrad", (sep_b, two_b, one_b))):
        ts = [timeit(f, a.iters) for f in fs]
        print("%-8s four launches %7.1f us   downsample + head launch %7.1f us   one launch %7.1f us (%4.0f TF/s)   x%.2f" %
              (name, ts[0], ts[1], ts[2], gflop / ts[2] * 1e3, ts[0] / ts[2]))


if __name__ == "__main__":
    main()
