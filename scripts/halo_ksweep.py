#!/usr/bin/env python
"""Fixed cost vs per-K-step cost of a conv launch: one 3x3 layer shape (N x H x W pixels, Cout channels) timed over
Cin = 64 .. 1024 (9 K-steps per 64 input channels) — the intercept of the line is what a launch costs before / after
its K loop, the slope what a K-step costs.  Graph replay (no host launch cost), halo configuration vs generic kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from torch_detection_amd import ops  # noqa: E402
from halo_ablate import graph_time  # noqa: E402  (imports only: halo_ablate guards its main)


def main():
    N, H, W, cout = 2, 50, 84, 256
    if len(sys.argv) > 1:
        N, H, W, cout = [int(v) for v in sys.argv[1:5]]
    cfgs = [("generic", {"TDN_HALO": "0"}), ("halo c11", {"TDN_HALO_CFG3": "11"}), ("halo c1", {"TDN_HALO_CFG3": "1"}),
            ("halo c10", {"TDN_HALO_CFG3": "10"}), ("halo c5", {"TDN_HALO_CFG3": "5"})]
    print("3x3 conv %dx%dx%d -> %d channels; us per launch by Cin" % (N, H, W, cout))
    for label, env in cfgs:
        for k in ("TDN_HALO", "TDN_HALO_CFG3"):
            os.environ.pop(k, None)
        os.environ.update(env)
        cells = []
        pts = []
        for cin in (64, 128, 256, 512, 1024):
            x = torch.randn(N, H, W, cin, device="cuda").bfloat16()
            w = (torch.randn(cout, 3, 3, cin, device="cuda") * 0.05).bfloat16()
            us = graph_time(lambda: ops.conv2d_fwd(x, w, 3, 1, 1, relu=True), 20)
            cells.append("%4d:%6.1f" % (cin, us))
            pts.append((cin / 64 * 9, us))
        (s0, t0), (s1, t1) = pts[1], pts[-1]
        slope = (t1 - t0) / (s1 - s0)
        print("%-10s %s | per K-step %.3f us, intercept %.1f us" % (label, "  ".join(cells), slope, t0 - slope * s0))


if __name__ == "__main__":
    main()
