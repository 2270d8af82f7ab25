#!/usr/bin/env python
"""Debug aid for csrc/conv_block.hip: repeated launches at the layer1 geometry against the per-conv reference;
prints where mismatches sit (tile, pixel inside the tile) — a hazard shows up as run-to-run differences."""
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TDN_GEMM_CFG"] = "0"
os.environ["TDN_HALO"] = "0"
from torch_detection_amd import ops  # noqa: E402
from test_gpu_block import _case  # noqa: E402

C = int(os.environ.get("C", 64))
N, H, W = int(os.environ.get("N", 1)), (200 if C == 64 else 100), (336 if C == 64 else 168)
dt = torch.bfloat16
dev = torch.device("cuda")
x, w1, w2, w3, aff = _case(N, H, W, C, dt, 4242)
xg, w1g, w2g, w3g = (t.contiguous().to(dev) for t in (x, w1, w2, w3))
affg = [a.to(dev) for a in aff]
# a few small launches first, like the test session does
xs, a1, a2, a3, afs = _case(1, 13, 21, C, dt, 7)
ops.bottleneck_fwd(xs.to(dev), a1.to(dev), a2.to(dev), a3.to(dev), [a.to(dev) for a in afs])
r1 = ops.conv2d_fwd(xg, w1g, 1, 1, 0, affg[0], affg[1], relu=True)
r2 = ops.conv2d_fwd(r1, w2g, 3, 1, 1, affg[2], affg[3], relu=True)
r3 = ops.conv2d_fwd(r2, w3g, 1, 1, 0, affg[4], affg[5], xg, ops.ADD_SAME, True)
torch.cuda.synchronize()
bad_runs = 0
for rep in range(int(os.environ.get("REPS", 30))):
    bits = ops.bottleneck_bit_planes(N, H, W, C, dev)
    h1, h2, out = ops.bottleneck_fwd(xg, w1g, w2g, w3g, affg, bits=bits)
    torch.cuda.synchronize()
    msg = []
    for name, a, b in (("h1", h1, r1), ("h2", h2, r2), ("out", out, r3)):
        ne = (a.view(torch.int16) != b.view(torch.int16)).any(dim=3)     # per pixel
        if bool(ne.any()):
            idx = ne.nonzero().cpu()
            tiles = collections.Counter((int(n), int(y) // 8, int(xx) // 16) for n, y, xx in idx.tolist())
            local = collections.Counter((int(y) % 8, int(xx) % 16) for n, y, xx in idx.tolist())
            msg.append("%s: %d pixels in %d tiles %s ; local (y,x) top: %s" % (
                name, idx.shape[0], len(tiles), list(tiles.items())[:6], local.most_common(8)))
    if msg:
        bad_runs += 1
        print("rep %d: " % rep + " | ".join(msg))
print("bad runs: %d" % bad_runs)
