#!/usr/bin/env python
"""Where a workgroup of the one-launch Bottleneck kernel spends its cycles (libtdn_trace.so: cycle stamps at the
phase boundaries, wave 0 of every workgroup).  python scripts/block_trace.py [--batch 1] [--bwd]"""
import argparse
import ctypes
import os
import sys

os.environ.setdefault("TDN_LIB", "libtdn_trace.so")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch_detection_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--bwd", action="store_true")
ap.add_argument("--C", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda")
C = a.C
N, H, W = a.batch, (200 if C == 64 else 100), (336 if C == 64 else 168)
C4 = 4 * C
dt = torch.bfloat16
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.relu(torch.randn(N, H, W, C4, device=dev, generator=gen)).to(dt)
w1 = (torch.randn(C, 1, 1, C4, device=dev, generator=gen) * (2.0 / C4) ** 0.5).to(dt)
w2 = (torch.randn(C, 3, 3, C, device=dev, generator=gen) * (2.0 / (9 * C)) ** 0.5).to(dt)
w3 = (torch.randn(C4, 1, 1, C, device=dev, generator=gen) * (2.0 / C) ** 0.5).to(dt)
aff = [torch.rand(n, device=dev, generator=gen) + 0.5 for n in (C, C, C, C, C4, C4)]
lib = ctypes.CDLL(_lib.LIB_PATH)
ntiles = N * ((H + 7) // 8) * ((W + 15) // 16)
nwg = (ntiles + 7) & ~7
buf = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
h1, h2, out = ops.bottleneck_fwd(x, w1, w2, w3, aff)
g = torch.where(out > 0, torch.randn(N, H, W, C4, device=dev, generator=gen).to(dt) * 0.1, torch.zeros((), device=dev, dtype=dt)).contiguous()
w1d, w2d, w3d = (w.permute(3, 1, 2, 0).contiguous() for w in (w1, w2, w3))


def run():
    if a.bwd:
        ops.bottleneck_dgrad(g, w3d, w2d, w1d, (h2, h1, x))
    else:
        ops.bottleneck_fwd(x, w1, w2, w3, aff)


for _ in range(5):
    run()
torch.cuda.synchronize()
lib.tdn_debug_block_trace(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.tdn_debug_block_trace(ctypes.c_void_p(0))
t = buf.cpu().numpy().reshape(nwg, 16).astype(np.int64)
t = t[t[:, 0] != 0]
if C == 64:
    names = ["entry->first K-step landed", "K-steps 0-3", "K-steps 4-7 (b0)", "epilogue 1 + b1 (taps 2-6 landed)", "taps 0,1 + b2",
             "taps 2-6 + b3", "taps 7,8 + b4", "epilogue 2 + b5 (W3 landed)", "pass 0 MFMA + epilogue math", "pass 1 MFMA + epilogue math",
             "last stores issued", "stores drained"]
    cols = list(range(13))
else:
    names = ["entry->first K-step landed", "K-steps 0-7", "K-steps 8-15 (b0)", "epilogue 1, stores, unit 0 wait", "units 0-8",
             "units 9-17 + b4", "addend loads, epilogue 2, b5", "pass 0 MFMA + b6 + epilogue math", "stores, b7, pass 1 + epilogue math",
             "last stores issued", "stores drained"]
    cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12]
d = np.diff(t[:, cols], axis=1)
tot = t[:, 12] - t[:, 0]
print("%d workgroups; cycles per segment: median / p10 / p90; share of the median lifetime %d cycles" % (t.shape[0], np.median(tot)))
for i, nm in enumerate(names):
    print("  %-36s %7d %7d %7d   %4.1f%%" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90),
                                           100.0 * np.median(d[:, i]) / np.median(tot)))
rt0, rt1 = t[:, 14], t[:, 15]
base = rt0.min()
print("wall (100 MHz clock): first entry -> last exit %.1f us; entries: p50 %.1f us p90 %.1f us max %.1f us; lifetime median %.1f us"
      % ((rt1.max() - base) / 100.0, np.median(rt0 - base) / 100.0, np.percentile(rt0 - base, 90) / 100.0, (rt0.max() - base) / 100.0,
         np.median(rt1 - rt0) / 100.0))
clk = np.median(tot) / (np.median(rt1 - rt0) / 100.0) / 1e3
print("shader clock while running: %.2f GHz" % clk)
