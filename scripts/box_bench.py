#!/usr/bin/env python
"""Roofline measurement of the HBM-bound kernels of the path (SURVEY §8(d), config C3 and the element-wise passes):
anchor grid, pairwise IoU, NMS, max pool, GroupNorm passes, image batch staging — HIP-event time per launch,
algorithmic bytes / time against 8 TB/s (spec) and 6.3 TB/s (achievable), with the CPU oracle (oracle/box_ref.c,
scalar C, 1 core) timed beside the box ops.  Prints one JSON object per kernel."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_detection_amd as T  # noqa: E402
from torch_detection_amd import ops  # noqa: E402

HBM_SPEC, HBM_ACH = 8.0e12, 6.3e12


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, secs, nbytes, cpu=None, note=""):
    d = {"kernel": name, "us": round(secs * 1e6, 1), "algorithmic_MB": round(nbytes / 1e6, 2),
         "GB_per_s": round(nbytes / secs / 1e9, 1), "frac_of_8TBps": round(nbytes / secs / HBM_SPEC, 3),
         "frac_of_6.3TBps": round(nbytes / secs / HBM_ACH, 3)}
    if cpu is not None:
        d["cpu_oracle_ms"] = round(cpu * 1e3, 2)
        d["cpu_kind"] = "oracle/box_ref.c, 1 core"
    if note:
        d["note"] = note
    print(json.dumps(d), flush=True)


def main():
    from oracle import box_ref as B
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    # ---- C3: 10,000 boxes on the 800 x 1344 canvas ----
    N = 10000
    wh = torch.rand(N, 2, generator=g) * 248 + 8
    x1 = torch.rand(N, generator=g) * (1344 - wh[:, 0])
    y1 = torch.rand(N, generator=g) * (800 - wh[:, 1])
    boxes = torch.stack([x1, y1, x1 + wh[:, 0], y1 + wh[:, 1]], 1).float()
    scores = torch.rand(N, generator=g)
    bg, sg = boxes.to(dev), scores.to(dev)
    t = timeit(lambda: ops.bbox_iou_pairwise(bg, bg), 10)
    bn = boxes.numpy()
    t0 = time.perf_counter(); B.iou_pairwise(bn, bn); cpu = time.perf_counter() - t0
    report("bbox_iou_pairwise 10k x 10k", t, 16 * 2 * N + 4 * N * N, cpu)
    t = timeit(lambda: ops.nms(bg, sg, 0.5), 10)
    t0 = time.perf_counter(); B.nms(bn, scores.numpy(), 0.5); cpu = time.perf_counter() - t0
    words = (N + 63) // 64
    report("nms 10k boxes, thr 0.5 (rank + scatter + mask + scan)", t, 20 * N + 2 * 8 * N * words + N, cpu,
           "latency-bound by the single-wave keep scan, not by bytes")
    ag = T.AnchorGenerator(8, [8], [0.5, 1.0, 2.0])
    levels = [((200, 336), 4), ((100, 168), 8), ((50, 84), 16), ((25, 42), 32), ((13, 21), 64)]

    def anchors():
        for fs, st in levels:
            ag.grid_anchors(fs, st, dev)
    t = timeit(anchors, 20)
    nanch = sum(h * w * 3 for (h, w), _ in levels)
    base = B.base_anchors(8, [8], [0.5, 1.0, 2.0])
    t0 = time.perf_counter()
    for fs, st in levels:
        B.anchor_grid(base, fs, st)
    cpu = time.perf_counter() - t0
    report("anchor_grid 5 levels, %d anchors (5 launches)" % nanch, t, 17 * nanch, cpu, "launch-bound: 5 tiny launches")
    gens = [ag] * 5
    t = timeit(lambda: T.anchor_pyramid(gens, [fs for fs, _ in levels], [st for _, st in levels], dev), 20)
    report("anchor_pyramid 5 levels, %d anchors (1 launch, incl. the operator layer's allocations)" % nanch, t,
           17 * nanch, cpu)
    # ---- element-wise passes at bench size ----
    x = torch.randn(2, 400, 672, 64, device=dev).bfloat16()
    t = timeit(lambda: ops.maxpool3x3s2_fwd(x))
    report("maxpool3x3s2_fwd 2x400x672x64", t, x.numel() * 2 + x.numel() // 4 * 3)
    y, idx = ops.maxpool3x3s2_fwd(x)
    dy = torch.randn_like(y)
    t = timeit(lambda: ops.maxpool3x3s2_bwd(dy, idx, (400, 672), x))
    report("maxpool3x3s2_bwd (+ ReLU mask from the pool's input)", t, dy.numel() * 3 + x.numel() * 4)
    t = timeit(lambda: ops.maxpool3x3s2_bwd(dy, idx, (400, 672), pooled=y))
    report("maxpool3x3s2_bwd (+ ReLU mask from the pool's output: what the backbone uses)", t,
           dy.numel() * 5 + x.numel() * 2)
    z = torch.randn(2, 200, 336, 256, device=dev).bfloat16()
    gam, bet = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    t = timeit(lambda: ops.gn_fwd(z, gam, bet, 32, 1e-5, z, True))
    report("gn_fwd 2x200x336x256 (+ residual, ReLU; 3 launches)", t, z.numel() * 2 * 4)
    yy, st = ops.gn_fwd(z, gam, bet, 32)
    t = timeit(lambda: ops.gn_bwd(z, z, st, gam, 32))
    report("gn_bwd 2x200x336x256 (3 launches)", t, z.numel() * 2 * 5)
    imgs = [torch.randint(0, 256, (800, 1333, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    tr = T.ImageTransforms((123.675, 116.28, 103.53), (58.395, 57.12, 57.375), 32)
    t = timeit(lambda: tr(imgs))
    report("collate_images 2 x 800x1333 uint8 -> fp32 NCHW 800x1344", t, 2 * 800 * 1333 * 3 + 2 * 3 * 800 * 1344 * 4)
    t = timeit(lambda: tr(imgs, staged=True))
    report("collate_images ... -> staged bf16 NHWC4", t, 2 * 800 * 1333 * 3 + 2 * 806 * 1352 * 8)
    xim = torch.randn(2, 3, 800, 1344, device=dev)
    t = timeit(lambda: ops.stage_image(xim))
    report("stage_image 2x3x800x1344 fp32 -> bf16 NHWC4", t, xim.numel() * 4 + 2 * 806 * 1352 * 8)


if __name__ == "__main__":
    main()
