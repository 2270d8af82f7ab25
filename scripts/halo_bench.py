#!/usr/bin/env python
"""Halo-tile conv kernel (csrc/conv_halo.hip) against the generic implicit-GEMM kernel (csrc/conv_igemm.hip) over the
R50-FPN shapes of SURVEY Appendix A: results must be bit-identical (same K order, same MFMA sequence per output
element), then both are timed.  --sweep tries the halo configurations / patch shapes given on the command line.

  python scripts/halo_bench.py --mode fwd --batch 2 [--filter 3x3] [--cfg3 0,1 --cfg1 4,0] [--check-only]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from torch_detection_amd import ops  # noqa: E402
from conv_bench import SHAPES  # noqa: E402
from conv_bench import timeit as timeit_eager  # noqa: E402

GRAPH = [False]


def timeit(fn, iters):
    """Device time per launch.  --graph: the launches are captured into one hipGraph and replayed — eager launches
    from Python cost ~10-12 us of host time each, which is what an event-bracketed loop of kernels shorter than that
    measures (every figure under ~12 us from the eager loop is the host's, not the kernel's)."""
    if not GRAPH[0]:
        return timeit_eager(fn, iters)
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

HALO_KEYS = ("TDN_GEMM_CFG", "TDN_HALO", "TDN_HALO_CFG3", "TDN_HALO_CFG1", "TDN_HALO_TH", "TDN_HALO_TW", "TDN_HALO_NT",
             "TDN_HALO_XBUF")


def set_env(**kw):
    for k in HALO_KEYS:
        os.environ.pop(k, None)
    for k, v in kw.items():
        if v is not None:
            os.environ[k] = str(v)


def make_case(name, cin, cout, k, s, H, W, B, mode, dev, epi):
    pad = k // 2
    Ho, Wo = ops.conv_out_size(H, k, s, pad), ops.conv_out_size(W, k, s, pad)
    gen = torch.Generator(device=dev).manual_seed(hash(name) & 0xffff)
    if mode == "fwd":
        x = torch.randn(B, H, W, cin, device=dev, generator=gen).bfloat16()
        w = (torch.randn(cout, k, k, cin, device=dev, generator=gen) * 0.05).bfloat16()
        kw = {}
        if epi:
            kw["scale"] = torch.rand(cout, device=dev, generator=gen) + 0.5
            kw["shift"] = torch.randn(cout, device=dev, generator=gen) * 0.1
            kw["addend"] = torch.randn(B, Ho, Wo, cout, device=dev, generator=gen).bfloat16()
            kw["addend_mode"] = ops.ADD_SAME
        return lambda: ops.conv2d_fwd(x, w, k, s, pad, relu=True, **kw)
    g = torch.randn(B, Ho, Wo, cout, device=dev, generator=gen).bfloat16()
    wd = (torch.randn(cin, k, k, cout, device=dev, generator=gen) * 0.05).bfloat16()
    kw = {}
    if epi:
        kw["addend"] = torch.randn(B, H, W, cin, device=dev, generator=gen).bfloat16()
        kw["addend_mode"] = ops.ADD_SAME
        kw["mask_src"] = torch.randn(B, H, W, cin, device=dev, generator=gen).bfloat16()
    return lambda: ops.conv2d_dgrad(g, wd, (H, W), k, s, pad, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--mode", default="fwd", choices=["fwd", "dgrad"])
    ap.add_argument("--filter", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cfg3", default="", help="halo configs to sweep on 3x3 layers (comma list)")
    ap.add_argument("--cfg1", default="", help="halo configs to sweep on 1x1 layers")
    ap.add_argument("--patches", default="", help="THxTW list to sweep on 3x3 layers, e.g. 8x16,3x42")
    ap.add_argument("--nt", default="", help="passes per workgroup to sweep on 1x1 layers")
    ap.add_argument("--xbuf", default="", help="TDN_HALO_XBUF values to sweep (1 resident, 2 double buffer)")
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--graph", action="store_true", help="time graph replays of the launches (no host launch cost)")
    ap.add_argument("--no-epi", action="store_true")
    args = ap.parse_args()
    GRAPH[0] = args.graph
    dev = "cuda"
    lib_plan = ops._lib.load().tdn_conv2d_plan
    import ctypes
    tot = {"generic": 0.0, "halo": 0.0, "best": 0.0}
    bad = 0
    print("%-20s %7s | generic us(TF/s) | halo default | sweep ..." % ("shape", "GFLOP"))
    for name, cin, cout, k, s, H, W, cnt in SHAPES:
        if args.filter and args.filter not in name:
            continue
        pad = k // 2
        Ho, Wo = ops.conv_out_size(H, k, s, pad), ops.conv_out_size(W, k, s, pad)
        gflop = 2.0 * args.batch * Ho * Wo * cout * cin * k * k / 1e9
        fn = make_case(name, cin, cout, k, s, H, W, args.batch, args.mode, dev, not args.no_epi)
        set_env(TDN_GEMM_CFG=0)          # reference: generic 64x64 tile, K order (chunk, tap) like the halo kernel's
        ref = fn().clone()
        set_env(TDN_HALO=0)              # timed: whatever tile the generic path picks by itself
        us_g = 0.0 if args.check_only else timeit(fn, args.iters)
        o = (ctypes.c_int32 * 16)()
        set_env()
        lib_plan(0 if args.mode == "fwd" else 1, args.batch, H, W, cin, cout, k, s, pad, o)
        forced = args.cfg3 if k == 3 else args.cfg1
        if o[8] < 100 and not forced:
            print("%-20s %7.2f | %5.0f(%4.0f) | generic kernel (halo does not apply)" % (name, gflop, us_g, gflop / max(us_g, 1e-9) * 1e3))
            tot["generic"] += us_g * cnt
            tot["halo"] += us_g * cnt
            tot["best"] += us_g * cnt
            continue
        variants = [("dflt[c%d %dx%d xb%d]" % (o[8] - 100, o[11] // 1000, o[11] % 1000, o[12] // 100), {})] \
            if o[8] >= 100 else []
        cfgkey = "TDN_HALO_CFG3" if k == 3 else "TDN_HALO_CFG1"
        for c in [c for c in (args.cfg3 if k == 3 else args.cfg1).split(",") if c]:
            variants.append(("c" + c, {cfgkey: c, "TDN_HALO": 3}))
            if k == 3:
                for pt in [p_ for p_ in args.patches.split(",") if p_]:
                    th, tw = pt.split("x")
                    variants.append(("c%s/%s" % (c, pt), {cfgkey: c, "TDN_HALO_TH": th, "TDN_HALO_TW": tw}))
            else:
                for nt in [n_ for n_ in args.nt.split(",") if n_]:
                    variants.append(("c%s/nt%s" % (c, nt), {cfgkey: c, "TDN_HALO_NT": nt, "TDN_HALO": 3}))
            for xb in [x_ for x_ in args.xbuf.split(",") if x_]:
                variants.append(("c%s/xb%s" % (c, xb), {cfgkey: c, "TDN_HALO_XBUF": xb}))
        cells = []
        best = 1e30
        us_default = None
        for label, env in variants:
            set_env(**env)
            lib_plan(0 if args.mode == "fwd" else 1, args.batch, H, W, cin, cout, k, s, pad, o)
            if o[8] < 100:
                cells.append("%s:n/a" % label)
                continue
            y = fn()
            torch.cuda.synchronize()
            if not torch.equal(y, ref):
                d = (y.float() - ref.float()).abs()
                nbad = int((d > 0).sum())
                cells.append("%s:MISMATCH(%d el, max %.3g)" % (label, nbad, float(d.max())))
                bad += 1
                continue
            if args.check_only:
                cells.append("%s:ok" % label)
                continue
            us = timeit(fn, args.iters)
            if us_default is None:
                us_default = us
            best = min(best, us)
            cells.append("%s:%.0f(%.0f)" % (label, us, gflop / us * 1e3))
        set_env()
        if us_default is None:
            us_default = us_g
            best = min(best, us_g)
        tot["generic"] += us_g * cnt
        tot["halo"] += us_default * cnt
        tot["best"] += min(best, us_g) * cnt
        print("%-20s %7.2f | %5.0f(%4.0f) | %s" % (name, gflop, us_g, gflop / max(us_g, 1e-9) * 1e3, "  ".join(cells)))
    print("weighted totals (us per pass, x layer count):", {k_: round(v) for k_, v in tot.items()})
    print("MISMATCHES: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
