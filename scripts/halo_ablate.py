#!/usr/bin/env python
"""Timing-only ablations of the halo kernel (library built with `make ABL=1`): full kernel vs no MFMA (1) vs no
LDS-DMA in the K loop (2) vs no fragment reads (3), configurations 0 (128x128, 4 waves) and 1 (256x128, 8 waves)."""
import os

# alternate tiles / ablation and cycle-stamp builds live in libtdn_trace.so (make -C torch_detection_amd/csrc TRACE=1)
os.environ.setdefault("TDN_LIB", "libtdn_trace.so")
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from torch_detection_amd import ops  # noqa: E402
from conv_bench import timeit  # noqa: E402

def graph_time(fn, iters):
    """Graph replay of `iters` launches: device time per launch without the host's launch cost."""
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


CASES = [("l3.c2 256 3x3", 256, 256, 50, 84), ("l2.c2 128 3x3", 128, 128, 100, 168), ("fpn.out0", 256, 256, 200, 336),
         ("fpn.out1", 256, 256, 100, 168)]
def main():
    for name, cin, cout, H, W in CASES:
        x = torch.randn(2, H, W, cin, device="cuda").bfloat16()
        w = (torch.randn(cout, 3, 3, cin, device="cuda") * 0.05).bfloat16()
        fn = lambda: ops.conv2d_fwd(x, w, 3, 1, 1, relu=True)  # noqa: E731
        for cfg in ("11", "1"):
            cells = []
            for abl in ("0", "1", "2", "3"):
                os.environ["TDN_HALO_CFG3"] = cfg
                os.environ["TDN_HALO_ABL"] = abl
                cells.append("abl%s:%.1f" % (abl, graph_time(fn, 20)))
            print("%-16s cfg %s | %s" % (name, cfg, "  ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
