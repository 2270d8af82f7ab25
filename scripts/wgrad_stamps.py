#!/usr/bin/env python
"""In-kernel stamps of the tap-per-tile weight-gradient kernel (libtdn_trace.so, `make TRACE=1`): for one stage group
of R50-FPN, where a workgroup's time goes (setup + first data, K loop per step, tile store), how many workgroups a CU
holds at a time and how the launch fills the chip over time."""
import argparse
import os
import sys

os.environ.setdefault("TDN_LIB", "libtdn_trace.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="l3")
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    from torch_detection_amd import ops, _lib
    import wgrad_group_bench as wb
    import ctypes
    lib = _lib.load()
    lib.tdn_debug_wgrad_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]   # a bare int would be cut to 32 bits
    lib.tdn_debug_wgrad_trace.restype = ctypes.c_int
    dev = torch.device("cuda")
    st = wb.stage_shapes(50)
    groups = wb.split_groups(st[args.stage])
    grp = groups[-1]     # the tap-per-tile group of the stage
    # build tensors the way the bench does
    B = args.batch
    items, keep = [], []
    for name, cin, cout, k, s, H, W in grp:
        x = torch.randn(B, H, W, cin, device=dev).bfloat16()
        Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
        g = (torch.randn(B, Ho, Wo, cout, device=dev) * 0.1).bfloat16()
        w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).bfloat16()
        bn = not name.startswith("fpn")
        sc = (torch.rand(cout, device=dev) + 0.5) if bn else None
        mean = torch.randn(cout, device=dev) * 0.1 if bn else None
        inv = (torch.rand(cout, device=dev) + 0.5) if bn else None
        it, dw, dg, db = ops.conv2d_wgrad_item(x, g, w, k, s, k // 2, sc, mean, inv)
        items.append(it)
        keep += [x, g, w, sc, mean, inv, dw, dg, db]
    cap = 16384
    buf = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
    for _ in range(20):    # warm: clocks, plan cache
        ops.wgrad_group(items, torch.bfloat16, dev)
    torch.cuda.synchronize()
    _lib.check(lib.tdn_debug_wgrad_trace(buf.data_ptr(), cap), "tdn_debug_wgrad_trace")
    ops.wgrad_group(items, torch.bfloat16, dev)
    torch.cuda.synchronize()
    lib.tdn_debug_wgrad_trace(None, 0)
    t = buf.cpu().numpy().astype(np.uint64).reshape(cap, 8)
    live = t[:, 0] != 0
    t = t[live]
    print("workgroups stamped: %d (several launches of different tile shapes share block ids: the LAST writer wins)" % len(t))
    c0, c1, c2, c3 = (t[:, i].astype(np.int64) for i in range(4))
    r0, r1 = t[:, 4].astype(np.int64), t[:, 5].astype(np.int64)
    T = (t[:, 6] >> np.uint64(32)).astype(np.int64)
    member = (t[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
    hw = (t[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (t[:, 7] >> np.uint64(32)).astype(np.int64) & 0xf
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 0x7
    ok = (c3 > c0) & (T > 0)
    wall_us = (r1 - r0) / 100.0
    clk = (c3 - c0) / np.maximum(wall_us, 1e-9) / 1e3     # GHz
    print("in-kernel clock (cycle counter / 100 MHz wall), median over workgroups: %.2f GHz" % np.median(clk[ok]))
    launch_us = (r1.max() - r0.min()) / 100.0
    print("launch: first entry -> last exit %.1f us; workgroup wall median %.1f us (min %.1f, max %.1f)" %
          (launch_us, np.median(wall_us[ok]), wall_us[ok].min(), wall_us[ok].max()))
    print("%-8s %5s %6s | %10s %12s %10s | %9s" % ("member", "wgs", "Ksteps", "setup+1st", "cyc/K-step", "store", "wall us"))
    for m in sorted(set(member[ok].tolist())):
        s_ = ok & (member == m)
        print("%-8d %5d %6d | %10.0f %12.0f %10.0f | %9.1f" %
              (m, s_.sum(), np.median(T[s_]), np.median((c1 - c0)[s_]), np.median(((c2 - c1) / np.maximum(T, 1))[s_]),
               np.median((c3 - c2)[s_]), np.median(wall_us[s_])))
    # occupancy over time: workgroups alive at 10 sample times, and per-CU co-residency
    t0 = r0.min()
    for frac in (0.1, 0.25, 0.5, 0.75, 0.9):
        at = t0 + frac * (r1.max() - t0)
        alive = ok & (r0 <= at) & (r1 > at)
        cus = set(zip(xcc[alive].tolist(), se[alive].tolist(), sh[alive].tolist(), cu[alive].tolist()))
        print("at %3.0f%% of the launch: %4d workgroups alive on %3d distinct (xcc, se, sh, cu)" % (frac * 100, alive.sum(), len(cus)))


if __name__ == "__main__":
    main()
