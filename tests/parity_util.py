"""Teacher-forced parity check of the HIP ResNet+FPN schedule against oracle/sched_ref.py (test infrastructure).

Deep bf16 pipelines cannot be compared end to end at rounding accuracy: a 1e-6 difference in fp32 accumulation
order flips a few bf16 roundings, each flip perturbs the next layer's roundings, and after a handful of layers
two bit-different-but-equally-valid implementations sit a full quantisation-noise apart (~1e-2 forward); through
~50 ReLU masks the gradients then differ by tens of percent.  So the comparison is made *in situ*:

  forward   every fused launch is recomputed on CPU from the GPU's own input tensors of that launch;
  backward  the CPU schedule is given the GPU's saved activations (identical ReLU masks / pool indices) and
            runs the whole backward itself; bf16 rounding noise of the activation-gradients then only adds up
            linearly (~sqrt(#layers) * 2^-9), while any routing / indexing / epilogue bug is O(1).
"""
import os

import torch
import torch.nn.functional as F

from golden_util import det_tensor, fill_state_dict, rel_l2


def nchw(t):
    return t.detach().float().cpu().permute(0, 3, 1, 2).contiguous()


def run_teacher_forced(T, depth, shape, dev="cuda", threads=None, dtype=torch.bfloat16, cot_scale=1.0,
                       res_gain=1.0, end_to_end=True):
    """Returns a dict of measured relative-L2 errors (forward in situ, backward teacher-forced, and the plain
    end-to-end distances to the fp32 autograd oracle for the record)."""
    from oracle import sched_ref as S
    from oracle import torch_ref as O
    from torch_detection_amd import functional as HF
    torch.set_num_threads(threads or min(16, os.cpu_count() or 1))
    chans = [64, 128, 256, 512] if depth < 50 else [256, 512, 1024, 2048]
    rb, rf = T.ResNet(depth), T.FPN(chans, 256, 5)
    sdb = fill_state_dict(rb.state_dict(), 50)
    sdf = fill_state_dict(rf.state_dict(), 51)
    if res_gain != 1.0:   # damp every residual branch (its last BN's gamma): keeps a deep net inside fp16's range
        last_bn = "bn2.weight" if depth < 50 else "bn3.weight"
        for k in sdb:
            if k.endswith(last_bn):
                sdb[k] = sdb[k] * res_gain
    rb.load_state_dict(sdb)
    rf.load_state_dict(sdf)
    rb.to(dev).train()
    rf.to(dev)
    rb.compute_dtype = rf.compute_dtype = dtype   # bf16 (default) or fp16 operands; parameters stay fp32
    x = det_tensor(shape, 700, -2, 2)
    cap = {}
    bwd_launches = []
    HF.DEBUG_CAPTURE = cap
    try:
        outs = rf(rb(x.to(dev)))
    finally:
        HF.DEBUG_CAPTURE = None
    # cot_scale: the loss scale of an fp16 run (a power of two, so the cotangents stay exactly representable)
    cots = [det_tensor(tuple(o.shape), 710 + i, -1, 1) * cot_scale for i, o in enumerate(outs)]
    HF.DEBUG_BWD = bwd_launches
    try:
        torch.autograd.backward(outs, [c.to(dev).to(o.dtype) for c, o in zip(cots, outs)])
        torch.cuda.synchronize()
    finally:
        HF.DEBUG_BWD = None
    assert all(o.dtype == dtype for o in outs), [o.dtype for o in outs]
    got = {}
    for prefix, mod in (("backbone.", rb), ("neck.", rf)):
        for k, p in mod.named_parameters():
            assert p.grad is not None, k
            got[prefix + k] = p.grad.detach().float().cpu()
    st, saved = cap["seq"]
    xs, lat = cap["fpn"]
    g_s = nchw(st["s"])
    g_saved = [tuple(nchw(t) if t is not None else None for t in sv) for sv in saved]
    g_lat = [nchw(t) for t in lat]
    g_outs = [o.detach().float().cpu() for o in outs]

    sch = S.Sched(sdb, sdf, depth, 5, quant=dtype)
    fwd = {}
    # ---- forward, launch by launch, from the GPU's own inputs ----
    fwd["stem"] = rel_l2(g_s, sch.stem.fwd(S.rnd(x, dtype), relu=True))
    fwd["maxpool"] = rel_l2(g_saved[0][0], F.max_pool2d(g_s, 3, 2, 1))
    worst_blk = 0.0
    for blk, (bx, h1, h2, out) in zip(sch.blocks, g_saved):
        worst_blk = max(worst_blk, rel_l2(h1, blk.u1.fwd(bx, relu=True)))
        res = bx if blk.ud is None else blk.ud.fwd(bx)   # not saved by the HIP path: one launch deep on CPU
        if blk.kind == "bottleneck":
            worst_blk = max(worst_blk, rel_l2(h2, blk.u2.fwd(h1, relu=True)))
            worst_blk = max(worst_blk, rel_l2(out, blk.u3.fwd(h2, res, "same", True)))
        else:
            worst_blk = max(worst_blk, rel_l2(out, blk.u2.fwd(h1, res, "same", True)))
    fwd["blocks_worst"] = worst_blk
    feats = [g_saved[i][3] for i in sch.stage_last]
    worst = 0.0
    for i in reversed(range(4)):
        ref = sch.lat_u[i].fwd(feats[i]) if i == 3 else sch.lat_u[i].fwd(feats[i], g_lat[i + 1], "up2x")
        worst = max(worst, rel_l2(g_lat[i], ref))
    for i in range(4):
        worst = max(worst, rel_l2(g_outs[i], sch.fpn_u[i].fwd(g_lat[i])))
    worst = max(worst, rel_l2(g_outs[4], g_outs[3][:, :, ::2, ::2]))
    fwd["fpn_worst"] = worst
    # ---- backward, launch by launch, from the GPU's own operands of that launch ----
    bwd = backward_in_situ(sch, rb, rf, bwd_launches, S.rnd(x, dtype))
    # ---- backward with the GPU's saved activations ----
    sch.x = S.rnd(x, dtype)
    sch.out_shapes = [tuple(o.shape) for o in g_outs]
    sch.load_saved(g_s, g_saved, g_lat)
    ref_grads = sch.backward(cots)
    assert set(ref_grads) == set(got)
    eg = {k: rel_l2(got[k], ref_grads[k]) for k in got}
    srt = sorted(eg.values())
    worst_g = max(eg.items(), key=lambda kv: kv[1])
    # ---- for the record: end-to-end distance to the fp32 autograd oracle (== the reference's arithmetic) ----
    if end_to_end:
        ref_outs, ref32 = O.resnet_fpn_fwd_bwd(sdb, sdf, x, depth, cots)
        eo32 = [rel_l2(a, b) for a, b in zip(g_outs, ref_outs)]
        eg32 = sorted(rel_l2(got[k], ref32[k]) for k in got)
    else:   # full-size runs skip the fp32 autograd pass of the whole net (the in-situ checks are the point there)
        eo32, eg32 = [0.0], [0.0]
    return {"forward_in_situ": fwd,
            "backward_in_situ": bwd,
            "backward_teacher_forced": {"grad_worst": list(worst_g), "grad_median": srt[len(srt) // 2]},
            "end_to_end_vs_fp32_autograd": {"out": eo32, "grad_median": eg32[len(eg32) // 2],
                                            "grad_worst": eg32[-1]}}


def backward_in_situ(sch, rb, rf, launches, x_img):
    """Every dgrad launch and every weight-gradient member of the GPU's backward pass, recomputed on the CPU by the
    schedule oracle's unit of the same layer from the GPU's OWN operands of that launch (the activation gradient g it
    read, the addend / ReLU-mask tensors of its epilogue, the saved forward input).  Identical inputs on both sides:
    what is left is one layer's arithmetic — fp32 accumulation order and one rounding to the 16-bit storage type —
    so the per-launch bound is the north star's 1e-3, for every layer, at any depth.
    Returns the worst relative-L2 error per kind and where it occurred."""
    from torch_detection_amd import ops
    units = {}
    conv_name = {}
    for prefix, mod in (("backbone.", rb), ("neck.", rf)):
        for name, m in mod.named_modules():
            if isinstance(m, torch.nn.Conv2d):
                conv_name[id(m)] = prefix + name
    units["backbone.conv1"] = sch.stem
    for blk in sch.blocks:
        units["backbone.%s.conv1" % blk.p] = blk.u1
        units["backbone.%s.conv2" % blk.p] = blk.u2
        if blk.u3 is not None:
            units["backbone.%s.conv3" % blk.p] = blk.u3
        if blk.ud is not None:
            units["backbone.%s.downsample.0" % blk.p] = blk.ud
    for i in range(sch.nlat):
        units["neck.lateral_convs.%d.conv" % i] = sch.lat_u[i]
        units["neck.fpn_convs.%d.conv" % i] = sch.fpn_u[i]
    worst = {"dgrad": [0.0, None], "dw": [0.0, None], "dgamma": [0.0, None], "dbeta_or_dbias": [0.0, None]}
    count = {"dgrad": 0, "wgrad": 0}

    def upd(kind, err, name):
        if err > worst[kind][0]:
            worst[kind] = [err, name]

    for rec in launches:
        name = conv_name[id(rec[1].conv)]
        ref_u = units[name]
        if rec[0] == 'dgrad':
            _, u, g, in_hw, addend, mode, mask_src, dx = rec
            a = nchw(addend) if addend is not None and mode != ops.ADD_NONE else None
            ref = ref_u.dgrad(nchw(g), in_hw, a, 'same' if mode == ops.ADD_SAME else 'sumpool',
                              nchw(mask_src) if mask_src is not None else None)
            upd("dgrad", rel_l2(nchw(dx), ref), name)
            count["dgrad"] += 1
        else:
            _, u, x_in, g, img_hw, (dw, dg, db) = rec
            xin = x_img if u.is_stem else nchw(x_in)
            ref = ref_u.wgrad(xin, nchw(g))
            upd("dw", rel_l2(dw.detach().float().cpu(), ref[0]), name)
            if len(ref) == 3:
                upd("dgamma", rel_l2(dg.float().cpu(), ref[1]), name)
                upd("dbeta_or_dbias", rel_l2(db.float().cpu(), ref[2]), name)
            elif len(ref) == 2:
                upd("dbeta_or_dbias", rel_l2(db.float().cpu(), ref[1]), name)
            count["wgrad"] += 1
    out = {k: v for k, v in worst.items()}
    out["launches"] = count
    return out


# bounds (relative L2); see module docstring
FWD_IN_SITU_TOL = 1e-3      # north-star figure for conv activations, per fused launch on identical inputs
                            # (measured ~3e-5: a ~2e-4 fraction of outputs round to the neighbouring bf16)
BWD_TEACHER_TOL = 6e-2      # every parameter gradient, whole backward, identical saved activations
                            # (measured: R18 ~1e-2, R50 ~2e-2, R101 ~3e-2 worst; medians ~1e-2): rounding noise of the
                            # 16-bit activation-gradients adding up over the depth.  This run checks the ROUTING of the
                            # schedule (a missing residual / FPN / stage gradient is O(1)); the arithmetic of every
                            # launch is bounded separately at 1e-3 on its own operands (backward_in_situ).
BWD_TEACHER_TOL_DEEP = 6e-2  # R101: the same bound since the per-launch backward check exists (was 1.2e-1)
FWD_END_TO_END_TOL = 2e-2   # bf16 activations vs the fp32 reference path (~1e-2 expected)


BWD_IN_SITU_TOL = 1e-3      # every dgrad launch / weight-gradient member on its own operands (fp32 outputs of the
                            # weight gradients: accumulation order only; dgrad: one 16-bit rounding, like the forward)


def check(res, depth=50):
    f = res["forward_in_situ"]
    assert max(f.values()) <= FWD_IN_SITU_TOL, f
    b = res["backward_in_situ"]
    assert b["launches"]["dgrad"] > 0 and b["launches"]["wgrad"] > 0, b
    for kind in ("dgrad", "dw", "dgamma", "dbeta_or_dbias"):
        assert b[kind][0] <= BWD_IN_SITU_TOL, (kind, b[kind])
    tol = BWD_TEACHER_TOL if depth <= 50 else BWD_TEACHER_TOL_DEEP
    assert res["backward_teacher_forced"]["grad_worst"][1] <= tol, res["backward_teacher_forced"]
    assert res["backward_teacher_forced"]["grad_median"] <= 3e-2, res["backward_teacher_forced"]
    assert max(res["end_to_end_vs_fp32_autograd"]["out"]) <= FWD_END_TO_END_TOL, res["end_to_end_vs_fp32_autograd"]
