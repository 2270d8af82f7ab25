#!/usr/bin/env python
"""bench.py — images/sec of ResNet-50-FPN forward+backward at "1333x800" (zero-padded to 800x1344) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL); this file reads
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.
One step = one pass of the hot path over one synthetic batch: stage image -> ResNet-50 -> FPN (5 levels)
forward, backward from fixed cotangents g_l = randn_like(P_l)/numel(P_l) (the reference has no head / loss,
SURVEY §8(d) C4), all parameter gradients (conv weights, BN gamma/beta, FPN biases), and for N > 1 the bucketed
RCCL all-reduce of the 26.85 M gradients overlapped with backward.  Weak scaling: 2 images per GPU.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

H_PAD, W_PAD = 800, 1344           # "1333x800" padded to a multiple of 32 (image.py:326-347)
FWD_GFLOP = {50: 296.96, 101: 456.06}     # per image, SURVEY §8(d)
FWDBWD_GFLOP = {50: 885.8, 101: 1363.1}   # 3*fwd - dgrad(conv1)
MFMA_PEAK_TFLOPS = 2500.0          # bf16 dense, MI355X_MICROARCH.md
DOM = dict(Cin=256, Cout=256, k=3, stride=1, H=200, W=336)   # neck.fpn_convs.0: 79.27 GFLOP / image


def dominant_traffic():
    """HBM bytes of ONE launch of the dominant kernel at batch 2 from the newest committed PMC summary
    (profiles/rNN_pmc_dominant_kernel.txt, written by scripts/pmc_bench.sh + scripts/pmc_summarize.py: separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md).  Read from the file — never typed in
    here — so the figure cannot outlive the kernel it was measured on; (None, None) if there is no summary."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_dominant_kernel.txt")))
    for f in reversed(files):
        txt = open(f).read()
        blocks = re.split(r"\n\s*\n", txt)
        for b in blocks:      # the first kernel entry is the tagged dominant launch
            m = re.search(r"HBM bytes per launch: corrected ([0-9.eE+]+)", b)
            if m and ("conv_halo_kernel" in b or "conv_gemm_kernel" in b):
                return float(m.group(1)), os.path.relpath(f, ROOT)
    return None, None


def build_models(depth, device, seed=0):
    import torch_detection_amd as T
    torch.manual_seed(seed)
    backbone = T.BACKBONES.module_dict["ResNet"](depth)
    neck = T.NECKS.module_dict["FPN"]([256, 512, 1024, 2048], 256, 5)
    backbone.init_weights()
    neck.init_weights()
    g = torch.Generator().manual_seed(1)
    for m in backbone.modules():          # non-trivial BN statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    backbone.to(device).train()
    neck.to(device).train()
    return backbone, neck


def conv_inventory(depth):
    """(Cin, Cout, k, stride, Hin, Win, has_dgrad) of every conv of ResNet-`depth` + FPN(256, 5 levels) at 800x1344 —
    SURVEY Appendix A rebuilt from the architecture (resnet.py:178-184 arch_settings, :122-155 _make_res_layer,
    fpn.py:44-58): 61 convs for depth 50, 112 for depth 101."""
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
    convs = [(3, 64, 7, 2, H_PAD, W_PAD, False)]
    H, W, cin = H_PAD // 4, W_PAD // 4, 64
    for si, n in enumerate(blocks):
        planes = 64 * 2 ** si
        for bi in range(n):
            s_ = 2 if (bi == 0 and si > 0) else 1
            convs.append((cin, planes, 1, 1, H, W, True))                       # conv1
            convs.append((planes, planes, 3, s_, H, W, True))                   # conv2 carries the stride (resnet.py:75)
            Ho, Wo = H // s_, W // s_
            convs.append((planes, planes * 4, 1, 1, Ho, Wo, True))              # conv3
            if bi == 0:
                convs.append((cin, planes * 4, 1, s_, H, W, True))              # downsample
            cin, H, W = planes * 4, Ho, Wo
    h, w = H_PAD // 4, W_PAD // 4
    for li, c in enumerate((256, 512, 1024, 2048)):
        convs.append((c, 256, 1, 1, h >> li, w >> li, True))                    # lateral
        convs.append((256, 256, 3, 1, h >> li, w >> li, True))                  # output conv
    return convs


def per_layer_roofline_ms(depth, nimg):
    """the step's time if every conv pass ran at its own roofline,
    sum over layers and passes (forward, input gradient, weight gradient) of max(F / 2.5 PFLOP/s, bytes / 8 TB/s),
    each tensor moved once: 16-bit activations / gradients / weights, fp32 weight gradients."""
    tot = 0.0
    for cin, cout, k, s_, H, W, has_dgrad in conv_inventory(depth):
        Ho, Wo = H // s_, W // s_
        F_ = 2.0 * nimg * Ho * Wo * cout * cin * k * k
        x_b, y_b, w_n = nimg * H * W * cin * 2.0, nimg * Ho * Wo * cout * 2.0, cout * cin * k * k
        passes = [x_b + y_b + w_n * 2, x_b + y_b + w_n * 4] + ([x_b + y_b + w_n * 2] if has_dgrad else [])
        for by in passes:
            tot += max(F_ / (MFMA_PEAK_TFLOPS * 1e12), by / 8.0e12)
    return tot * 1e3


def dominant_by_time():
    """The kernel symbol with the largest share of summed kernel time in the newest committed rocprofv3 --stats
    summary of this command (profiles/rNN_bench_kernel_stats.csv) and, where the newest PMC summary has that symbol,
    its MFMA-pipe busy fraction.  Read from the files, never typed in here."""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_kernel_stats.csv")))
    if not files:
        return None
    rows = list(csv.DictReader(open(files[-1])))
    if not rows:
        return None
    fam = {}
    for r in rows:
        # pool the instantiations of one kernel template by tile shape: kernel name + its first two template arguments
        # (BM, BN of conv_gemm_kernel / FM, FN of conv_halo_kernel / BMW, BNW of the weight-gradient kernels), so that
        # e.g. the plain, the K-group and the BK = 128 builds of the 64 x 64 tile count as one family
        full = re.sub(r"^void ", "", r["Name"]).split("(")[0]
        m = re.match(r"([A-Za-z0-9_]+)<\s*([^,>]+)\s*,\s*([^,>]+)", full)
        name = "%s<%s, %s, ...>" % (m.group(1), m.group(2).strip(), m.group(3).strip()) if m else full
        fam[name] = fam.get(name, 0.0) + float(r["Percentage"])
    name, share = max(fam.items(), key=lambda kv: kv[1])
    out = {"symbol": name, "share": round(share / 100.0, 4), "source": os.path.relpath(files[-1], ROOT), "mfma_busy": None}
    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_*.txt")))
    for f in reversed(pmc):
        for b in re.split(r"\n\s*\n", open(f).read()):
            head = b.strip().split("\n")[0].strip()
            stem = name.split(", ...>")[0]
            if head and not head.startswith("#") and head.rstrip(".>") and head.startswith(stem):
                m = re.search(r"MFMA pipe utilisation: ([0-9.]+)", b)
                if m:
                    out["mfma_busy"] = float(m.group(1))
                    out["mfma_busy_source"] = os.path.relpath(f, ROOT)
                    return out
    return out


class KernelTimer(object):
    """HIP-event timing of the dominant launch (the 3x3 256->256 conv GEMM at M = N*200*336 — neck.fpn_convs.0
    forward and its dgrad, 2 launches per step; conv_halo_kernel's 256x128 tile, or conv_gemm_kernel<192,256,...> with
    TDN_HALO=0) on the stream it is launched on."""

    def __init__(self, ops, match):
        self.ops, self.match, self.pairs = ops, match, []
        self._fwd, self._dgrad = ops.conv2d_fwd, ops.conv2d_dgrad

    def _timed(self, fn, nimg, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        os.environ["TDN_TAG_DOMINANT"] = "1"   # same kernel code under its own symbol: rocprofv3 --stats then lists
        try:                                   # exactly the launches timed here on one line
            e0.record()
            y = fn(*a, **kw)
            e1.record()
        finally:
            os.environ.pop("TDN_TAG_DOMINANT", None)
        self.pairs.append((e0, e1, nimg))
        return y

    def __enter__(self):
        m = self.match

        def fwd(x, w_fwd, k, stride, pad, *a, **kw):
            hit = (tuple(x.shape[1:]) == (m["H"], m["W"], m["Cin"]) and w_fwd.shape[0] == m["Cout"] and k == m["k"]
                   and stride == m["stride"])
            if not hit:
                return self._fwd(x, w_fwd, k, stride, pad, *a, **kw)
            return self._timed(self._fwd, x.shape[0], x, w_fwd, k, stride, pad, *a, **kw)

        def dgrad(g, w_dgrad, in_hw, k, stride, pad, *a, **kw):
            hit = (tuple(g.shape[1:]) == (m["H"], m["W"], m["Cout"]) and w_dgrad.shape[0] == m["Cin"] and k == m["k"]
                   and stride == m["stride"] and tuple(in_hw) == (m["H"], m["W"]))
            if not hit:
                return self._dgrad(g, w_dgrad, in_hw, k, stride, pad, *a, **kw)
            return self._timed(self._dgrad, g.shape[0], g, w_dgrad, in_hw, k, stride, pad, *a, **kw)

        self.ops.conv2d_fwd, self.ops.conv2d_dgrad = fwd, dgrad
        return self

    def __exit__(self, *exc):
        self.ops.conv2d_fwd, self.ops.conv2d_dgrad = self._fwd, self._dgrad

    def summary(self):
        if not self.pairs:
            return None
        ms = [a.elapsed_time(b) for a, b, _ in self.pairs]
        n = self.pairs[0][2]
        return sum(ms) / len(ms), n, len(ms)


def usable_cores():
    """Host cores this process may actually use: affinity mask and cgroup CPU quota, not the machine total
    (a 1-GPU box exposes 256 logical CPUs but grants a 16-core share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    cap = int(os.environ.get("TDN_BENCH_CPU_THREADS", "16"))
    return max(1, min(n, cap))


def cpu_baseline(depth, iters=3):
    """The CPU oracle (restatement of the reference's PyTorch-CPU path, bit-equal to it: oracle/gen_golden.py)
    timed on this host's cores on a bounded sample: `iters` fwd+bwd passes of ONE 3x800x1344 image."""
    from golden_util import det_tensor
    from oracle import torch_ref as O
    import torch_detection_amd as T
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    rb, rf = T.ResNet(depth), T.FPN([256, 512, 1024, 2048], 256, 5)
    rb.init_weights()
    rf.init_weights()
    sdb, sdf = rb.state_dict(), rf.state_dict()
    x = det_tensor((1, 3, H_PAD, W_PAD), 1, -2, 2)
    sizes = [(H_PAD // s, W_PAD // s) for s in (4, 8, 16, 32)] + [((H_PAD // 32 + 1) // 2, (W_PAD // 32 + 1) // 2)]
    cots = [torch.randn(1, 256, h, w) / (256 * h * w) for h, w in sizes]
    times = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        O.resnet_fpn_fwd_bwd(sdb, sdf, x, depth, cots)
        times.append(time.perf_counter() - t0)
    timed = sorted(times[1:])
    best, med = timed[0], timed[len(timed) // 2]
    return {"value": round(1.0 / best, 4), "median": round(1.0 / med, 4), "unit": "images/sec", "cores": cores,
            "kind": "port",
            "sample": "%d timed fwd+bwd passes (after 1 warm-up) of one 3x%dx%d image through oracle/torch_ref.py "
                      "(R%d-FPN, fp32, torch CPU, %d threads); best %.2f s, median %.2f s" %
                      (iters, H_PAD, W_PAD, depth, cores, best, med)}


def box_secondary(device):
    """BASELINE config C3 beside the headline: pairwise IoU 10k x 10k, NMS of 10k boxes (thr 0.5) and the 5-level
    anchor pyramid (268,569 anchors) — HIP-event time per call on this process's stream, algorithmic bytes of
    SURVEY §8(d) / time.  (rocprofv3 figures of the same calls: profiles/rNN_box_kernel_stats.csv.)"""
    import torch_detection_amd as T
    from torch_detection_amd import ops

    def timeit(fn, iters):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    g = torch.Generator().manual_seed(0)
    N = 10000
    wh = torch.rand(N, 2, generator=g) * 248 + 8
    x1 = torch.rand(N, generator=g) * (W_PAD - wh[:, 0])
    y1 = torch.rand(N, generator=g) * (H_PAD - wh[:, 1])
    boxes = torch.stack([x1, y1, x1 + wh[:, 0], y1 + wh[:, 1]], 1).float().to(device)
    scores = torch.rand(N, generator=g).to(device)
    out = {}
    t = timeit(lambda: ops.bbox_iou_pairwise(boxes, boxes), 10)
    nb = 16 * 2 * N + 4 * N * N
    out["iou_10k_x_10k"] = {"us": round(t * 1e6, 1), "GB_per_s": round(nb / t / 1e9, 1), "algorithmic_MB": nb / 1e6}
    t = timeit(lambda: ops.nms(boxes, scores, 0.5), 10)
    nb = 20 * N + 2 * 8 * N * ((N + 63) // 64) + N
    out["nms_10k_thr0.5"] = {"us": round(t * 1e6, 1), "GB_per_s": round(nb / t / 1e9, 1), "algorithmic_MB": nb / 1e6,
                             "note": "bound by the serial keep scan, not by bytes"}
    ag = T.AnchorGenerator(8, [8], [0.5, 1.0, 2.0])
    sizes = [(H_PAD // s, W_PAD // s) for s in (4, 8, 16, 32)] + [((H_PAD // 32 + 1) // 2, (W_PAD // 32 + 1) // 2)]
    strides = [4, 8, 16, 32, 64]
    na = sum(h * w * 3 for h, w in sizes)
    # one launch for the whole pyramid, outputs preallocated and the C entry point called directly: the kernel takes
    # a few microseconds, so the operator layer's allocations would be what is timed otherwise
    import ctypes
    from torch_detection_amd import _lib
    lib = _lib.load()
    base = ag._base_on(torch.device(device))
    lv = (_lib.AnchorLevel * 5)()
    for l, ((fh, fw), st) in enumerate(zip(sizes, strides)):
        lv[l].base_anchors, lv[l].A = base.data_ptr(), 3
        lv[l].featH, lv[l].featW, lv[l].stride, lv[l].valid_h, lv[l].valid_w = fh, fw, st, fh, fw
    anchors = torch.empty(na, 4, dtype=torch.float32, device=device)
    valid = torch.empty(na, dtype=torch.uint8, device=device)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    t = timeit(lambda: lib.tdn_anchor_pyramid(lv, 5, ctypes.c_void_p(anchors.data_ptr()),
                                              ctypes.c_void_p(valid.data_ptr()), sp), 50)
    ref = T.anchor_pyramid([ag] * 5, sizes, strides, device)[0]
    assert torch.equal(torch.cat(ref), anchors)
    out["anchor_pyramid_5_levels"] = {"us": round(t * 1e6, 1), "GB_per_s": round(17 * na / t / 1e9, 1),
                                      "anchors": na, "algorithmic_MB": 17 * na / 1e6, "launches": 1}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--batch-per-gpu", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"],
                    help="operand type; f16 + --depth 101 --batch-per-gpu 4 = SURVEY §8(d) config C5 "
                         "(cotangents x1024: static loss scale)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the box-op and repack side measurements")
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--force-reducer", action="store_true",
                    help="world size 1 only: run the N>1 code path (RCCL process group, bucketed reducer inside the "
                         "captured step) on a single GPU — a rehearsal of what --gpus N executes")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "(--nproc-per-node %d)" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_reducer
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is one JSON line
        # on stdout, so file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            warm = torch.zeros(1, device=device)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from torch_detection_amd import dp, ops
    backbone, neck = build_models(args.depth, device)
    cdtype = torch.float16 if args.dtype == "f16" else torch.bfloat16
    backbone.compute_dtype = neck.compute_dtype = cdtype
    if world > 1:   # identical weights everywhere: broadcast rank 0's
        for p in list(backbone.state_dict().values()) + list(neck.state_dict().values()):
            dist.broadcast(p, 0)
    B = args.batch_per_gpu
    g = torch.Generator(device="cpu").manual_seed(rank)
    x = torch.zeros(B, 3, H_PAD, W_PAD)
    x[:, :, :, :1333] = torch.randn(B, 3, H_PAD, 1333, generator=g)   # logical 800x1333, zero right pad
    x = x.to(device)
    with torch.no_grad():
        outs = neck(backbone(x))
    g2 = torch.Generator(device="cpu").manual_seed(2)
    # cotangents in the layout the outputs have (NCHW-shaped, channels_last strides), as a head consuming the
    # pyramid on this path would hand them back
    loss_scale = 1024.0 if args.dtype == "f16" else 1.0
    cots = [(torch.randn(o.shape, generator=g2) * (loss_scale / o[0].numel())).to(device=device, dtype=o.dtype)
            .contiguous(memory_format=torch.channels_last) for o in outs]
    del outs

    reducer = dp.attach_reducer([neck, backbone], bucket_bytes=args.bucket_mb << 20, dtype=cdtype) \
        if use_dist else None
    params = list(backbone.parameters()) + list(neck.parameters())

    def step():
        if reducer is None:
            for p in params:
                p.grad = None
        outs = neck(backbone(x))
        torch.autograd.backward(outs, cots)
        if reducer is not None:
            reducer.finish()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- execution mode: the whole step — forward, backward and (N > 1) the bucketed RCCL all-reduces on their
    # comm stream — captured once into a hipGraph and replayed; eager launches if capture is refused ----
    graph, mode = None, "eager"
    repack_graph = None
    if not args.no_graph:
        from torch_detection_amd.graph import GraphedStep
        # headline: forward + backward with STATIC weights (the metric BASELINE.json names) — the packed 16-bit operands
        # are derived once; config.weights_static says so and ms_per_step_with_repack (below) is the same step with
        # the fold + pack launches of a real training step (weights change every iteration) inside the graph
        gs = GraphedStep(step, params=params, repack=False)
        ok = 1 if gs.captured else 0
        if use_dist:
            # every rank must run the same mode: a graph on some ranks and eager launches on others would still be
            # correct (same collectives in the same order) but is a configuration nobody has measured — agree on the
            # weakest outcome
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and ok:
                print("rank %d: another rank could not capture; running eager everywhere" % rank, file=sys.stderr)
                ok = 0
        if ok:
            graph = gs
            mode = "hipGraph replay (one captured fwd+bwd step; wgrad kernels on forked side streams%s)" % (
                "; bucketed RCCL all-reduce nodes on a comm stream inside the graph" if use_dist else "")
            if world == 1 and not args.no_secondary:
                rg = GraphedStep(step, params=params, repack=True)
                repack_graph = rg if rg.captured else None
        else:
            mode = "eager (graph capture failed%s)" % ("" if gs.captured else ": %s" % type(gs.error).__name__)

    plan = None
    if graph is None and not use_dist:
        # no hipGraph (refused, or --no-graph): the library's own executor — the launches of one recorded step,
        # enqueued again by one C call per step (torch_detection_amd.graph.PreparedStep, tdn_plan_* in include/tdn.h)
        from torch_detection_amd.graph import PreparedStep
        ps = PreparedStep(step, params=params, repack=False)
        if ps.prepared:
            plan = ps
            mode = "launch plan replay (libtdn executor: %d launches, %d events per step, one C call)" % ps.stats()[:2]
    if graph is None and plan is None:
        # eager launches: run autograd's backward on this thread — the hand-off to the engine's device thread costs
        # ~1.5 ms of the ~7.4 ms it takes to enqueue a step (scripts/host_profile.py), and the step is host-bound
        torch.autograd.set_multithreading_enabled(False)
        mode += ", single-threaded autograd"

    def run_step():
        if graph is not None:
            graph()
        elif plan is not None:
            plan()
        else:
            step()

    for _ in range(args.warmup):
        run_step()
    timer = None if (args.no_kernel_timer or graph is not None or plan is not None) else KernelTimer(ops, DOM)
    sync()
    t0 = time.perf_counter()
    if timer is not None:
        with timer:
            for _ in range(args.steps):
                run_step()
            sync()
            elapsed = time.perf_counter() - t0
    else:
        for _ in range(args.steps):
            run_step()
        sync()
        elapsed = time.perf_counter() - t0
    timed_in = "the timed region"
    if (graph is not None or plan is not None) and not args.no_kernel_timer:
        # HIP events cannot be read back from inside a replayed graph: time the dominant launch in eager steps of
        # the same process / tensors right after the timed region (same kernel, same stream, full-step context)
        timer = KernelTimer(ops, DOM)
        with timer:
            for _ in range(min(args.steps, 5)):
                step()
            torch.cuda.synchronize()
        timed_in = "eager steps right after the timed region (events cannot be read from a replayed graph)"
    ms_repack = None
    if repack_graph is not None:
        for _ in range(2):
            repack_graph()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            repack_graph()
        torch.cuda.synchronize()
        ms_repack = (time.perf_counter() - t0) / args.steps * 1e3
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        imgs = B * world * args.steps
        value = imgs / elapsed
        mfma_frac = value * FWDBWD_GFLOP[args.depth] / 1e3 / (world * MFMA_PEAK_TFLOPS)
        roof = None
        if timer is not None and timer.summary():
            ms, nimg, cnt = timer.summary()
            flop = 2.0 * nimg * DOM["H"] * DOM["W"] * DOM["Cout"] * DOM["Cin"] * DOM["k"] ** 2
            ach = flop / (ms * 1e-3) / 1e12
            traffic, traffic_src = dominant_traffic()
            import ctypes
            from torch_detection_amd import _lib
            plan = (ctypes.c_int32 * 16)()
            _lib.check(_lib.load().tdn_conv2d_plan(0, nimg, DOM["H"], DOM["W"], DOM["Cin"], DOM["Cout"], DOM["k"],
                                                   DOM["stride"], 1, plan), "tdn_conv2d_plan")
            roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic if nimg == 2 else None, "traffic_source": traffic_src,
                    "algorithmic_bytes": (2 * nimg * DOM["H"] * DOM["W"] * 256 + 256 * 2304) * 2,
                    "kernel": "%s, %dx%dx%d tile (symbol tagged for the timed launches) = 3x3 256->256 "
                              "conv GEMM, M=%d N=256 K=2304 (neck.fpn_convs.0 forward + its dgrad, 2 launches/step), "
                              "%.1f GFLOP per launch" %
                              ("conv_halo_kernel (LDS-resident %dx%d patch + halo, streamed weight ring)" %
                               (plan[11] // 1000, plan[11] % 1000) if plan[8] >= 100 else "conv_gemm_kernel",
                               plan[3], plan[4], plan[5], nimg * DOM["H"] * DOM["W"], flop / 1e9),
                    "avg_ms": round(ms, 4), "launches_timed": cnt, "timed_in": timed_in}
        line = {
            "metric": "images/sec ResNet-50-FPN fwd+bwd 1333x800" if args.depth == 50 else
                      "images/sec ResNet-%d-FPN fwd+bwd 1333x800" % args.depth,
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "ResNet-%d + FPN(256, 5 levels) forward+backward, %d x 3x800x1344 "
                                   "(1333x800 zero-padded to /32) per GPU, BN eval (reference default), fixed "
                                   "cotangents, all parameter grads%s" %
                                   (args.depth, B, ", bucketed RCCL all-reduce (sum/%d) overlapped with backward"
                                    % world if world > 1 else ""),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "execution": mode,
                       "weights_static": True,
                       "mfma_frac_whole_step": round(mfma_frac, 4),
                       "algorithmic_gflop_per_image": FWDBWD_GFLOP[args.depth]},
            "roofline": roof,
            # the whole step against the sum of its layers' own rooflines, and what dominates by time
            "roofline_step": {"per_layer_roofline_ms": round(per_layer_roofline_ms(args.depth, B), 4),
                              "frac": round(per_layer_roofline_ms(args.depth, B) / (elapsed / args.steps * 1e3), 4),
                              "note": "sum over the %d convs x (fwd, dgrad, wgrad) of max(F / 2.5 PFLOP/s, bytes / "
                                      "8 TB/s), each tensor once" % len(conv_inventory(args.depth))},
            "dominant_by_time": dominant_by_time(),
        }
        if ms_repack is not None:
            # the same step with every conv's BN fold + weight pack re-run inside the graph (grouped launches), i.e.
            # what one iteration of a training loop pays on top once an optimizer changes the weights
            line["ms_per_step_with_repack"] = round(ms_repack, 3)
            line["value_with_repack"] = round(B * world / (ms_repack * 1e-3), 2)
        # what torch.distributed reported after init (1 = no process group): lets the driver's N-rank check read the
        # world size the collectives actually ran over
        line["rccl_world"] = dist.get_world_size() if (use_dist and dist.is_initialized()) else 1
        if world == 1 and not args.no_secondary:
            line["secondary"] = box_secondary(device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.depth)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
