"""oracle/gen_golden.py — generate tests/golden/* by importing the reference on PyTorch-CPU.

TEST INFRASTRUCTURE ONLY.  This is the ONLY file that imports /root/reference; it runs in the build
container (the reference never travels to the GPU box).  It
  1. imports the reference with third-party shims (none modifies reference code):
     sys.path first, collections.Sequence/Mapping aliases, stub cv2 / pycocotools modules;
  2. asserts that oracle/torch_ref.py is BIT-EQUAL to the imported reference (ResNet-18/50/101 + FPN,
     forward and parameter/input gradients) on fp32 CPU;
  3. writes small golden vectors (inputs are regenerated from tests/golden_util.py, outputs are stored).
Run:  python oracle/gen_golden.py
"""
import collections
import collections.abc
import contextlib
import json
import warnings
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.path.insert(0, "/root/reference")
    collections.Sequence = collections.abc.Sequence
    collections.Mapping = collections.abc.Mapping
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    pc = types.ModuleType("pycocotools")
    pcc = types.ModuleType("pycocotools.coco")
    pcc.COCO = object
    sys.modules.setdefault("pycocotools", pc)
    sys.modules.setdefault("pycocotools.coco", pcc)
    import models.backbone  # noqa: F401
    import models.necks  # noqa: F401
    from models.registry import BACKBONES, NECKS
    import models.backbone.resnet as ref_resnet
    return BACKBONES, NECKS, ref_resnet


def manifest_of(m):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    BACKBONES, NECKS, ref_resnet = import_reference()
    from oracle import torch_ref as O
    from golden_util import det_tensor, fill_state_dict

    RefResNet = BACKBONES.module_dict["ResNet"]
    RefFPN = NECKS.module_dict["FPN"]
    os.makedirs(GOLD, exist_ok=True)
    man = {"registry": {"backbone": sorted(BACKBONES.module_dict), "neck": sorted(NECKS.module_dict)}}

    # ---- (1) state_dict manifests -------------------------------------------------------------
    for d in (18, 50, 101):
        man["resnet%d" % d] = manifest_of(RefResNet(d))
    man["fpn_r50"] = manifest_of(RefFPN([256, 512, 1024, 2048], 256, 5))
    man["fpn_r18"] = manifest_of(RefFPN([64, 128, 256, 512], 256, 5))

    # ---- (5) train() semantics ---------------------------------------------------------------
    m = RefResNet(18)
    ret = m.train()
    man["train_semantics"] = {
        "train_returns_none": ret is None,
        "all_bn_eval_after_train": all(not x.training for x in m.modules() if isinstance(x, torch.nn.BatchNorm2d)),
        "all_params_require_grad": all(p.requires_grad for p in m.parameters()),
        "feat_dim": {str(d): RefResNet(d).feat_dim for d in (18, 50, 101)},
    }
    try:
        RefResNet(20)
        man["bad_depth_error"] = None
    except KeyError as e:
        man["bad_depth_error"] = "KeyError:" + str(e)
    try:
        RefResNet(18).init_weights(pretrained=3)
        man["bad_pretrained_error"] = None
    except TypeError as e:
        man["bad_pretrained_error"] = "TypeError:" + str(e)

    # ---- (2) config 1: ResNet-18, 1x3x224x224 -------------------------------------------------
    m = RefResNet(18)
    sd = fill_state_dict(m.state_dict(), 18)
    m.load_state_dict(sd)
    m.train()  # BN -> eval per reference default (resnet.py:270-276)
    x = det_tensor((1, 3, 224, 224), 1001, -2.0, 2.0)
    with torch.no_grad():
        ref = m(x)
        mine = O.resnet_forward(sd, x, 18)
    for a, b in zip(ref, mine):
        assert torch.equal(a, b), "oracle != reference (R18 forward)"
    np.savez_compressed(os.path.join(GOLD, "resnet18_c1.npz"), **{"c%d" % (i + 2): t.numpy() for i, t in enumerate(ref)})
    man["resnet18_c1"] = {"input": {"shape": [1, 3, 224, 224], "seed": 1001, "lo": -2.0, "hi": 2.0},
                          "state_seed": 18, "out_shapes": [list(t.shape) for t in ref]}

    # R50 / R101 bit-equality at a small padded size (not stored: weights too large; checksums only)
    for d, seed in ((50, 50), (101, 101)):
        m = RefResNet(d)
        sd = fill_state_dict(m.state_dict(), seed)
        m.load_state_dict(sd)
        m.train()
        x = det_tensor((1, 3, 64, 96), 2000 + d, -2.0, 2.0)
        with torch.no_grad():
            ref = m(x)
            mine = O.resnet_forward(sd, x, d)
        for a, b in zip(ref, mine):
            assert torch.equal(a, b), "oracle != reference (R%d forward)" % d
        man["resnet%d_small" % d] = {
            "input": {"shape": [1, 3, 64, 96], "seed": 2000 + d, "lo": -2.0, "hi": 2.0}, "state_seed": seed,
            "sum": [float(t.double().sum()) for t in ref], "abssum": [float(t.double().abs().sum()) for t in ref]}

    # ---- (3) residual blocks: fwd + all grads ----------------------------------------------------
    blocks = {}
    cases = {
        "bottleneck_s1_nodown": (ref_resnet.Bottleneck, 256, 64, 1, (2, 10, 12)),
        "bottleneck_s1_down": (ref_resnet.Bottleneck, 64, 64, 1, (2, 10, 12)),
        "bottleneck_s2_down": (ref_resnet.Bottleneck, 128, 64, 2, (2, 9, 12)),
        "basic_s1": (ref_resnet.BasicBlock, 64, 64, 1, (2, 10, 12)),
        "basic_s2_down": (ref_resnet.BasicBlock, 64, 128, 2, (2, 10, 11)),
    }
    for ci, (name, (cls, inpl, planes, stride, (n, h, w))) in enumerate(sorted(cases.items())):
        blk = ref_resnet._make_res_layer(cls, inpl, planes, 1, stride=stride)[0]
        sd = fill_state_dict(blk.state_dict(), 300 + ci)
        blk.load_state_dict(sd)
        blk.eval()
        x = det_tensor((n, inpl, h, w), 400 + ci, -1.0, 1.0).requires_grad_(True)
        y = blk(x)
        dy = det_tensor(tuple(y.shape), 500 + ci, -1.0, 1.0)
        y.backward(dy)
        # oracle check (fwd + grads, bit-equal)
        kind = _basic = cls is ref_resnet.BasicBlock
        fn = O._basic_block if kind else O._bottleneck
        ps = {("b." + k): v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k)
              for k, v in sd.items()}
        x2 = x.detach().clone().requires_grad_(True)
        y2 = fn(x2, ps, "b", stride, 1, blk.downsample is not None)
        y2.backward(dy)
        assert torch.equal(y2, y) and torch.equal(x2.grad, x.grad), "oracle != reference (%s)" % name
        for k, p in blk.named_parameters():
            assert torch.equal(ps["b." + k].grad, p.grad), "oracle grad != reference (%s %s)" % (name, k)
        blocks[name + "/y"] = y.detach().numpy()
        blocks[name + "/dx"] = x.grad.numpy()
        for k, p in blk.named_parameters():
            blocks[name + "/grad/" + k] = p.grad.numpy()
        man.setdefault("blocks", {})[name] = {
            "cls": cls.__name__, "inplanes": inpl, "planes": planes, "stride": stride, "x_shape": [n, inpl, h, w],
            "state_seed": 300 + ci, "x_seed": 400 + ci, "dy_seed": 500 + ci,
            "state_keys": manifest_of(blk)}
    np.savez_compressed(os.path.join(GOLD, "blocks.npz"), **blocks)

    # ---- (4) FPN: fwd + input grads + param grads; and the odd-size failure ----------------------
    chans = [64, 128, 256, 512]
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3)]
    fpn = RefFPN(chans, 64, 5)
    sd = fill_state_dict(fpn.state_dict(), 600)
    fpn.load_state_dict(sd)
    ins = [det_tensor((2, c, h, w), 610 + i, -1.0, 1.0).requires_grad_(True) for i, (c, (h, w)) in
           enumerate(zip(chans, sizes))]
    outs = fpn(ins)
    cots = [det_tensor(tuple(o.shape), 620 + i, -1.0, 1.0) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, cots)
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    ins2 = [t.detach().clone().requires_grad_(True) for t in ins]
    outs2 = O.fpn_forward(ps, ins2, 5)
    torch.autograd.backward(outs2, cots)
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b), "oracle != reference (FPN forward)"
    for a, b in zip(ins, ins2):
        assert torch.equal(a.grad, b.grad), "oracle != reference (FPN input grad)"
    for k, p in fpn.named_parameters():
        assert torch.equal(ps[k].grad, p.grad), "oracle != reference (FPN grad %s)" % k
    f = {}
    for i, o in enumerate(outs):
        f["out%d" % i] = o.detach().numpy()
    for i, t in enumerate(ins):
        f["din%d" % i] = t.grad.numpy()
    for k, p in fpn.named_parameters():
        f["grad/" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(GOLD, "fpn.npz"), **f)
    man["fpn_small"] = {"in_channels": chans, "out_channels": 64, "num_outs": 5, "sizes": [list(s) for s in sizes],
                        "N": 2, "state_seed": 600, "in_seed0": 610, "cot_seed0": 620,
                        "out_shapes": [list(o.shape) for o in outs]}
    # failure case: a level that is not exactly 2x the next (fpn.py:99-101) raises
    bad = [det_tensor((1, c, h, w), 1, -1, 1) for c, (h, w) in zip(chans, [(16, 24), (8, 11), (4, 6), (2, 3)])]
    try:
        fpn(bad)
        man["fpn_odd_size_error"] = None
    except RuntimeError as e:
        man["fpn_odd_size_error"] = "RuntimeError"
        man["fpn_odd_size_error_msg"] = str(e)[:120]
    try:
        fpn(ins[:3])
        man["fpn_wrong_len_error"] = None
    except AssertionError:
        man["fpn_wrong_len_error"] = "AssertionError"

    # ---- PAFPN: fwd + input grads + param grads, with and without ReLU -------------
    RefPAFPN = NECKS.module_dict["PAFPN"]
    man["pafpn_keys"] = manifest_of(RefPAFPN(chans, 64, 5))
    pf = {}
    for tag, actv in (("none", None), ("relu", "relu"), ("relu6", "relu6")):
        pa = RefPAFPN(chans, 64, 5, activation=actv)
        sdp = fill_state_dict(pa.state_dict(), 800)
        pa.load_state_dict(sdp)
        # relu6: inputs x4 so that a visible share of the PA activations saturates at 6 (where nn.ReLU6 passes no
        # gradient — the case a plain-ReLU backward mask would get wrong)
        amp = 4.0 if actv == "relu6" else 1.0
        pins = [det_tensor((2, c, h, w), 810 + i, -amp, amp).requires_grad_(True) for i, (c, (h, w)) in
                enumerate(zip(chans, sizes))]
        pouts = pa(pins)
        pcots = [det_tensor(tuple(o.shape), 820 + i, -1.0, 1.0) for i, o in enumerate(pouts)]
        torch.autograd.backward(pouts, pcots)
        pps = {k: v.detach().clone().requires_grad_(True) for k, v in sdp.items()}
        pins2 = [t.detach().clone().requires_grad_(True) for t in pins]
        pouts2 = O.pafpn_forward(pps, pins2, 5, actv)
        torch.autograd.backward(pouts2, pcots)
        for a, b in zip(pouts, pouts2):
            assert torch.equal(a, b), "oracle != reference (PAFPN forward %s)" % tag
        for a, b in zip(pins, pins2):
            assert torch.equal(a.grad, b.grad), "oracle != reference (PAFPN input grad %s)" % tag
        for k, p in pa.named_parameters():
            assert torch.equal(pps[k].grad, p.grad), "oracle != reference (PAFPN grad %s %s)" % (tag, k)
        for i, o in enumerate(pouts):
            pf["%s/out%d" % (tag, i)] = o.detach().numpy().astype(np.float16 if False else np.float32)
        for i, t in enumerate(pins):
            pf["%s/din%d" % (tag, i)] = t.grad.numpy()
        for k, p in pa.named_parameters():
            if k.startswith("pa_convs") or tag == "none":
                pf["%s/grad/%s" % (tag, k)] = p.grad.numpy()
        if actv == "relu6":
            sat = [float((o.detach() >= 6).float().mean()) for o in pouts[1:4]]
            assert max(sat) > 0.01, "relu6 golden case does not saturate anywhere: %s" % sat
            man["pafpn_relu6_saturated_fraction"] = sat
    # RetinaNet-style extra levels: stride-2 convs on the last backbone input (pafpn.py:139-147), 6 outputs
    pa = RefPAFPN(chans, 64, 6, add_extra_convs=True)
    sdp = fill_state_dict(pa.state_dict(), 830)
    pa.load_state_dict(sdp)
    pins = [det_tensor((2, c, h, w), 840 + i, -1.0, 1.0).requires_grad_(True) for i, (c, (h, w)) in
            enumerate(zip(chans, sizes))]
    pouts = pa(pins)
    pcots = [det_tensor(tuple(o.shape), 850 + i, -1.0, 1.0) for i, o in enumerate(pouts)]
    torch.autograd.backward(pouts, pcots)
    pps = {k: v.detach().clone().requires_grad_(True) for k, v in sdp.items()}
    pins2 = [t.detach().clone().requires_grad_(True) for t in pins]
    pouts2 = O.pafpn_forward(pps, pins2, 6, None, True)
    torch.autograd.backward(pouts2, pcots)
    for a, b in zip(pouts, pouts2):
        assert torch.equal(a, b), "oracle != reference (PAFPN extra convs forward)"
    for a, b in zip(pins, pins2):
        assert torch.equal(a.grad, b.grad), "oracle != reference (PAFPN extra convs input grad)"
    for k, p in pa.named_parameters():
        assert torch.equal(pps[k].grad, p.grad), "oracle != reference (PAFPN extra convs grad %s)" % k
    for i, o in enumerate(pouts):
        pf["extra/out%d" % i] = o.detach().numpy()
    for i, t in enumerate(pins):
        pf["extra/din%d" % i] = t.grad.numpy()
    for k in ("fpn_convs.5.conv.weight", "fpn_convs.5.conv.bias", "fpn_convs.4.conv.bias", "pa_convs2.2.conv.weight"):
        pf["extra/grad/" + k] = dict(pa.named_parameters())[k].grad.numpy()
    man["pafpn_extra"] = {"in_channels": chans, "out_channels": 64, "num_outs": 6,
                          "sizes": [list(s_) for s_ in sizes], "N": 2, "state_seed": 830, "in_seed0": 840,
                          "cot_seed0": 850, "state_keys": manifest_of(pa),
                          "grad_keys": ["fpn_convs.5.conv.weight", "fpn_convs.5.conv.bias", "fpn_convs.4.conv.bias",
                                        "pa_convs2.2.conv.weight"]}
    np.savez_compressed(os.path.join(GOLD, "pafpn.npz"), **pf)
    man["pafpn_small"] = {"in_channels": chans, "out_channels": 64, "num_outs": 5, "sizes": [list(s) for s in sizes],
                          "N": 2, "state_seed": 800, "in_seed0": 810, "cot_seed0": 820}

    # ---- box delta (de)normalisation: the reference's own functions ---------------
    from datasets.utils.bbox import bbox_denormalize as ref_denorm, bbox_normalize as ref_norm
    from oracle import box_ref as BR
    means, stds = [0.0, 0.0, 0.0, 0.0], [0.1, 0.1, 0.2, 0.2]
    means2, stds2 = [0.5, -0.25, 0.125, 1.0], [0.3, 0.7, 1.1, 0.9]
    b4 = det_tensor((257, 4), 900, -3.0, 3.0, bf16=False)
    b12 = det_tensor((65, 12), 901, -3.0, 3.0, bf16=False)
    bn = {}
    for tag, (m, s) in (("a", (means, stds)), ("b", (means2, stds2))):
        n = ref_norm(b4.clone(), m, s)
        d4 = ref_denorm(b4.clone(), m, s)
        d12 = ref_denorm(b12.clone(), m, s)
        assert np.array_equal(n.numpy(), BR.np_bbox_normalize(b4.numpy(), m, s)), "oracle != reference (bbox_normalize)"
        assert np.array_equal(d4.numpy(), BR.np_bbox_denormalize(b4.numpy(), m, s))
        assert np.array_equal(d12.numpy(), BR.np_bbox_denormalize(b12.numpy(), m, s))
        bn[tag + "/norm"], bn[tag + "/denorm4"], bn[tag + "/denorm12"] = n.numpy(), d4.numpy(), d12.numpy()
    t = b4.clone()
    assert ref_norm(t, means, stds) is t      # in-place contract (bbox.py:140)
    np.savez_compressed(os.path.join(GOLD, "bbox_norm.npz"), **bn)
    man["bbox_norm"] = {"seed4": 900, "seed12": 901, "lo": -3.0, "hi": 3.0,
                        "a": [means, stds], "b": [means2, stds2], "normalize_in_place": True}

    # ---- GroupNorm variants: use_gn blocks, GN FPN, GN ResNet-18 ------------------------
    gn = {}
    gcases = {
        "bottleneck_s1_nodown": (ref_resnet.Bottleneck, 256, 64, 1, (2, 10, 12)),
        "bottleneck_s2_down": (ref_resnet.Bottleneck, 128, 64, 2, (2, 9, 12)),
        "basic_s2_down": (ref_resnet.BasicBlock, 64, 128, 2, (2, 10, 11)),
    }
    for ci, (name, (cls, inpl, planes, stride, (n, h, w))) in enumerate(sorted(gcases.items())):
        blk = ref_resnet._make_res_layer(cls, inpl, planes, 1, stride=stride, use_gn=True)[0]
        sd = fill_state_dict(blk.state_dict(), 1300 + ci)
        blk.load_state_dict(sd)
        x = det_tensor((n, inpl, h, w), 1400 + ci, -1.0, 1.0).requires_grad_(True)
        y = blk(x)
        dy = det_tensor(tuple(y.shape), 1500 + ci, -1.0, 1.0)
        y.backward(dy)
        fn = O._basic_block if cls is ref_resnet.BasicBlock else O._bottleneck
        ps = {("b." + k): v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        x2 = x.detach().clone().requires_grad_(True)
        y2 = fn(x2, ps, "b", stride, 1, blk.downsample is not None)
        y2.backward(dy)
        assert torch.equal(y2, y) and torch.equal(x2.grad, x.grad), "oracle != reference (GN %s)" % name
        for k, p in blk.named_parameters():
            assert torch.equal(ps["b." + k].grad, p.grad), "oracle grad != reference (GN %s %s)" % (name, k)
        gn["blk/" + name + "/y"] = y.detach().numpy()
        gn["blk/" + name + "/dx"] = x.grad.numpy()
        for k, p in blk.named_parameters():
            gn["blk/" + name + "/grad/" + k] = p.grad.numpy()
        man.setdefault("gn_blocks", {})[name] = {
            "cls": cls.__name__, "inplanes": inpl, "planes": planes, "stride": stride, "x_shape": [n, inpl, h, w],
            "state_seed": 1300 + ci, "x_seed": 1400 + ci, "dy_seed": 1500 + ci, "state_keys": manifest_of(blk)}
    # FPN with GroupNorm ConvModules (normalize given, use_gn=True: fpn.py:18-19,40-58)
    gfpn = RefFPN(chans, 64, 5, normalize=dict(type="GN"), use_gn=True)
    sd = fill_state_dict(gfpn.state_dict(), 1600)
    gfpn.load_state_dict(sd)
    ins = [det_tensor((2, c, h, w), 1610 + i, -1.0, 1.0).requires_grad_(True) for i, (c, (h, w)) in
           enumerate(zip(chans, sizes))]
    outs = gfpn(ins)
    cots = [det_tensor(tuple(o.shape), 1620 + i, -1.0, 1.0) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, cots)
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    ins2 = [t.detach().clone().requires_grad_(True) for t in ins]
    outs2 = O.fpn_forward(ps, ins2, 5)
    torch.autograd.backward(outs2, cots)
    for a_, b_ in zip(outs, outs2):
        assert torch.equal(a_, b_), "oracle != reference (GN FPN forward)"
    for a_, b_ in zip(ins, ins2):
        assert torch.equal(a_.grad, b_.grad), "oracle != reference (GN FPN input grad)"
    for k, p in gfpn.named_parameters():
        assert torch.equal(ps[k].grad, p.grad), "oracle != reference (GN FPN grad %s)" % k
    for i, o in enumerate(outs):
        gn["fpn/out%d" % i] = o.detach().numpy()
    for i, t in enumerate(ins):
        gn["fpn/din%d" % i] = t.grad.numpy()
    for k, p in gfpn.named_parameters():
        gn["fpn/grad/" + k] = p.grad.numpy()
    man["gn_fpn_small"] = {"in_channels": chans, "out_channels": 64, "num_outs": 5, "sizes": [list(s_) for s_ in sizes],
                           "N": 2, "state_seed": 1600, "in_seed0": 1610, "cot_seed0": 1620,
                           "state_keys": manifest_of(gfpn)}
    # whole GN backbone forward (stem GN + max pool + BasicBlocks): outputs only
    m = RefResNet(18, use_gn=True)
    sd = fill_state_dict(m.state_dict(), 1700)
    m.load_state_dict(sd)
    m.train()
    x = det_tensor((2, 3, 64, 96), 1701, -2.0, 2.0)
    with torch.no_grad():
        ref = m(x)
        mine = O.resnet_forward(sd, x, 18)
    for a_, b_ in zip(ref, mine):
        assert torch.equal(a_, b_), "oracle != reference (GN R18 forward)"
    for i, t in enumerate(ref):
        gn["r18/c%d" % (i + 2)] = t.numpy()
    man["gn_resnet18"] = {"input": {"shape": [2, 3, 64, 96], "seed": 1701, "lo": -2.0, "hi": 2.0}, "state_seed": 1700,
                          "state_keys": manifest_of(m), "out_shapes": [list(t.shape) for t in ref]}
    np.savez_compressed(os.path.join(GOLD, "gn.npz"), **gn)

    # ---- ResNeXt: grouped-conv bottlenecks fwd + all grads, ResNeXt-50 32x4d forward ------
    import importlib
    ref_resnext = importlib.import_module("models.backbone.resnext")
    RefResNeXt = BACKBONES.module_dict["ResNeXt"]
    rx = {}
    xcases = {
        "x32x4d_s1_down": (64, 64, 4, 32, 1, (2, 10, 12)),      # layer1.0: 128 ch, 4 per group
        "x32x4d_s2_down": (256, 128, 4, 32, 2, (2, 9, 12)),     # layer2.0: 256 ch, 8 per group, stride 2
        "x32x16d_s1_nodown": (256, 64, 16, 32, 1, (1, 6, 8)),   # D = 16: 512 ch, 16 per group, identity residual
    }
    for ci, (name, (inpl, planes, bw, card, stride, (n, h, w))) in enumerate(sorted(xcases.items())):
        blk = ref_resnext._make_resX_layer(ref_resnext.ResNeXtBottleneck, inpl, planes, 1, bw, card, stride=stride)[0]
        sd = fill_state_dict(blk.state_dict(), 1800 + ci)
        blk.load_state_dict(sd)
        blk.eval()
        x = det_tensor((n, inpl, h, w), 1810 + ci, -1.0, 1.0).requires_grad_(True)
        y = blk(x)
        dy = det_tensor(tuple(y.shape), 1820 + ci, -1.0, 1.0)
        y.backward(dy)
        ps = {("b." + k): v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k)
              for k, v in sd.items()}
        x2 = x.detach().clone().requires_grad_(True)
        y2 = O._resnext_bottleneck(x2, ps, "b", stride, card, blk.downsample is not None)
        y2.backward(dy)
        assert torch.equal(y2, y) and torch.equal(x2.grad, x.grad), "oracle != reference (ResNeXt %s)" % name
        for k, p in blk.named_parameters():
            assert torch.equal(ps["b." + k].grad, p.grad), "oracle grad != reference (ResNeXt %s %s)" % (name, k)
        rx["blk/" + name + "/y"] = y.detach().numpy()
        rx["blk/" + name + "/dx"] = x.grad.numpy()
        for k, p in blk.named_parameters():
            rx["blk/" + name + "/grad/" + k] = p.grad.numpy()
        man.setdefault("resnext_blocks", {})[name] = {
            "inplanes": inpl, "planes": planes, "base_width": bw, "cardinality": card, "stride": stride,
            "x_shape": [n, inpl, h, w], "state_seed": 1800 + ci, "x_seed": 1810 + ci, "dy_seed": 1820 + ci,
            "state_keys": manifest_of(blk)}
    m = RefResNeXt(50, 4, 32)
    sd = fill_state_dict(m.state_dict(), 1900)
    m.load_state_dict(sd)
    m.train()
    x = det_tensor((1, 3, 64, 96), 1901, -2.0, 2.0)
    with torch.no_grad():
        ref = m(x)
        mine = O.resnext_forward(sd, x, 50, 32)
    for a_, b_ in zip(ref, mine):
        assert torch.equal(a_, b_), "oracle != reference (ResNeXt-50 forward)"
    for i, t in enumerate(ref):
        rx["x50/c%d" % (i + 2)] = t.numpy()
    man["resnext50_32x4d"] = {"input": {"shape": [1, 3, 64, 96], "seed": 1901, "lo": -2.0, "hi": 2.0},
                              "state_seed": 1900, "state_keys": manifest_of(m),
                              "out_shapes": [list(t.shape) for t in ref],
                              "all_bn_eval_after_train": all(not x_.training for x_ in m.modules()
                                                             if isinstance(x_, torch.nn.BatchNorm2d))}
    np.savez_compressed(os.path.join(GOLD, "resnext.npz"), **rx)

    # ---- dilated stages: ResNet(strides=(1,2,1,1), dilations=(1,1,2,4)) ("DC5"-style), forward ------------------
    dil = {}
    for d_, seed in ((18, 2100), (50, 2101)):
        m = RefResNet(d_, strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4))
        sd = fill_state_dict(m.state_dict(), seed)
        m.load_state_dict(sd)
        m.train()
        x = det_tensor((1, 3, 64, 96), seed + 10, -2.0, 2.0)
        with torch.no_grad():
            ref = m(x)
            mine = O.resnet_forward(sd, x, d_, strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4))
        for a_, b_ in zip(ref, mine):
            assert torch.equal(a_, b_), "oracle != reference (dilated R%d forward)" % d_
        if d_ == 18:
            for i, t in enumerate(ref):
                dil["r18/c%d" % (i + 2)] = t.numpy()
        man["resnet%d_dilated" % d_] = {
            "strides": [1, 2, 1, 1], "dilations": [1, 1, 2, 4], "state_seed": seed,
            "input": {"shape": [1, 3, 64, 96], "seed": seed + 10, "lo": -2.0, "hi": 2.0},
            "out_shapes": [list(t.shape) for t in ref],
            "sum": [float(t.double().sum()) for t in ref], "abssum": [float(t.double().abs().sum()) for t in ref]}
    np.savez_compressed(os.path.join(GOLD, "dilated.npz"), **dil)

    # ---- BatchNorm with batch statistics: ResNet(18, bn_eval=False).train(), forward + backward ----------------
    bt = {}
    m = RefResNet(18, bn_eval=False)
    sd = fill_state_dict(m.state_dict(), 2200)
    m.load_state_dict(sd)
    m.train()
    assert all(x_.training for x_ in m.modules() if isinstance(x_, torch.nn.BatchNorm2d))
    x = det_tensor((2, 3, 64, 96), 2210, -2.0, 2.0)
    outs = m(x)
    cots = [det_tensor(tuple(o.shape), 2220 + i, -1.0, 1.0) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, cots)
    ps = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    with O.bn_training():
        outs2 = O.resnet_forward(ps, x, 18)
    torch.autograd.backward(outs2, cots)
    for a_, b_ in zip(outs, outs2):
        assert torch.equal(a_, b_), "oracle != reference (BN-train R18 forward)"
    for k, p in m.named_parameters():
        assert torch.equal(ps[k].grad, p.grad), "oracle != reference (BN-train grad %s)" % k
    new_sd = m.state_dict()
    for k in new_sd:
        if "running" in k:
            assert torch.equal(ps[k], new_sd[k]), "oracle != reference (running stat %s)" % k
    for i, o in enumerate(outs):
        bt["c%d" % (i + 2)] = o.detach().numpy()
    keep = ["conv1.weight", "bn1.weight", "bn1.bias", "layer1.0.conv1.weight", "layer2.0.downsample.1.weight",
            "layer2.1.conv1.weight", "layer3.1.bn2.bias", "layer4.1.bn2.weight"]
    for k in keep:
        bt["grad/" + k] = dict(m.named_parameters())[k].grad.numpy()
    for k in ("bn1.running_mean", "bn1.running_var", "layer4.1.bn2.running_mean", "layer4.1.bn2.running_var"):
        bt["stat/" + k] = new_sd[k].numpy()
    np.savez_compressed(os.path.join(GOLD, "bn_train.npz"), **bt)
    man["bn_train_r18"] = {"state_seed": 2200, "input": {"shape": [2, 3, 64, 96], "seed": 2210, "lo": -2.0, "hi": 2.0},
                           "cot_seed0": 2220, "grad_keys": keep, "out_shapes": [list(o.shape) for o in outs],
                           "num_batches_tracked_after": int(new_sd["bn1.num_batches_tracked"])}

    # ---- ConvModule: ReLU6 and the pre-activation order (layers.py:57-135), forward + all gradients ---------------
    from models.utils.layers import ConvModule as RefConvModule
    cm = {}
    cm_cases = [
        # tag, kernel, bias, normalize, use_gn, activation, activate_last, training
        ("post_bn_relu6", 3, False, True, False, "relu6", True, False),
        ("post_bias_relu6", 3, True, False, False, "relu6", True, False),
        ("post_gn_relu6", 1, False, True, True, "relu6", True, False),
        ("pre_bn_relu", 3, True, True, False, "relu", False, False),
        ("pre_bn_relu6_train", 3, True, True, False, "relu6", False, True),
        ("pre_gn_relu6", 3, True, True, True, "relu6", False, False),
        ("pre_none_relu", 1, True, False, False, "relu", False, False),
        ("pre_bn_noact", 3, False, True, False, None, False, False),
        # a conv with a bias AND a norm behind it: the reference warns (layers.py:84-85) and computes it
        ("post_bn_bias_relu", 3, True, True, False, "relu", True, False),
        ("post_gn_bias", 1, True, True, True, None, True, False),
        ("post_bn_bias_train_relu6", 3, True, True, False, "relu6", True, True),
    ]
    man["conv_module"] = {"input": {"shape": [2, 64, 12, 16], "lo": -8.0, "hi": 8.0}, "cases": []}
    for ci, (tag, k, bias, norm, use_gn, actv, last, training) in enumerate(cm_cases):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = RefConvModule(64, 64, k, padding=k // 2, bias=bias, normalize=dict() if norm else None, use_gn=use_gn,
                              activation=actv, activate_last=last)
        sd = fill_state_dict(m.state_dict(), 2600 + 10 * ci)
        m.load_state_dict(sd)
        m.train(training)
        x = det_tensor((2, 64, 12, 16), 2601 + 10 * ci, -8.0, 8.0).requires_grad_(True)
        y = m(x * 1.0)   # a non-leaf input: the reference's in-place activations refuse a leaf that requires grad
        cot = det_tensor(tuple(y.shape), 2602 + 10 * ci, -1.0, 1.0)
        y.backward(cot)
        ps = {k_: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k_)
              for k_, v in sd.items()}
        x2 = x.detach().clone().requires_grad_(True)
        ctxm = O.bn_training() if training else contextlib.nullcontext()
        with ctxm:
            y2 = O.conv_module_forward(ps, x2, 1, k // 2, actv, last)
        y2.backward(cot)
        assert torch.equal(y, y2), "oracle != reference (ConvModule %s forward)" % tag
        assert torch.equal(x.grad, x2.grad), "oracle != reference (ConvModule %s dx)" % tag
        for k_, p_ in m.named_parameters():
            assert torch.equal(ps[k_].grad, p_.grad), "oracle != reference (ConvModule %s grad %s)" % (tag, k_)
        cm[tag + "/y"] = y.detach().numpy()
        cm[tag + "/dx"] = x.grad.numpy()
        for k_, p_ in m.named_parameters():
            cm[tag + "/grad/" + k_] = p_.grad.numpy()
        if training:
            new_sd = m.state_dict()
            for k_ in ("norm.running_mean", "norm.running_var"):
                assert torch.equal(ps[k_], new_sd[k_]), "oracle != reference (ConvModule %s %s)" % (tag, k_)
                cm[tag + "/stat/" + k_] = new_sd[k_].numpy()
        man["conv_module"]["cases"].append({"tag": tag, "kernel": k, "bias": bias, "normalize": norm,
                                            "use_gn": use_gn, "activation": actv, "activate_last": last,
                                            "training": training, "state_seed": 2600 + 10 * ci,
                                            "input_seed": 2601 + 10 * ci, "cot_seed": 2602 + 10 * ci})
    np.savez_compressed(os.path.join(GOLD, "conv_module.npz"), **cm)

    # ---- image batch staging: normalize -> flip -> pad to /32 -> CHW -> collate ------------
    from datasets.utils.image import img_flip, img_normalize, img_pad_size_divisor
    from datasets.utils import DataContainer
    from datasets.loader.collate import collate as ref_collate
    from oracle import stage_ref as SR
    rng = np.random.RandomState(1234)
    sizes_hw = [(37, 50), (40, 41), (64, 33)]
    imgs = [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in sizes_hw]
    flips = [False, True, True]
    c_means, c_stds = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
    cg = {}
    for tag, srcs in (("u8", imgs), ("f32", [im.astype(np.float32) * np.float32(0.5) for im in imgs])):
        samples = []
        for im, fl in zip(srcs, flips):
            x = img_normalize(im, np.array(c_means, np.float32), np.array(c_stds, np.float32))
            x, flag, _ = img_flip(x, 1 if fl else 0)        # flip_prob 1 / 0: deterministic
            assert flag == fl
            x = img_pad_size_divisor(x, size_divisor=32)
            x = x.transpose(2, 0, 1)
            samples.append(DataContainer(torch.from_numpy(np.ascontiguousarray(x)), stack=True, padding_value=0))
        out = ref_collate(samples, sample_per_gpu=len(samples)).data[0]
        mine, pads = SR.np_collate_images(srcs, c_means, c_stds, flips, 32)
        assert out.dtype == torch.float32 and np.array_equal(out.numpy(), mine), "oracle != reference (collate %s)" % tag
        cg[tag + "/batch"] = out.numpy()
        for i, im in enumerate(srcs):
            cg["%s/img%d" % (tag, i)] = im
    np.savez_compressed(os.path.join(GOLD, "collate.npz"), **cg)
    man["collate"] = {"sizes_hw": [list(s) for s in sizes_hw], "flips": flips, "means": list(c_means),
                      "stds": list(c_stds), "size_divisor": 32, "batch_shape": list(cg["u8/batch"].shape)}

    # ---- R50+FPN end-to-end fwd+bwd oracle == reference (small, not stored) ----------------------
    rb = RefResNet(50)
    rf = RefFPN([256, 512, 1024, 2048], 256, 5)
    sdb = fill_state_dict(rb.state_dict(), 50)
    sdf = fill_state_dict(rf.state_dict(), 51)
    rb.load_state_dict(sdb)
    rf.load_state_dict(sdf)
    rb.train()
    x = det_tensor((1, 3, 64, 64), 700, -2.0, 2.0)
    outs = rf(rb(x))
    cots = [det_tensor(tuple(o.shape), 710 + i, -1.0, 1.0) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, cots)
    outs2, grads = O.resnet_fpn_fwd_bwd(sdb, sdf, x, 50, cots)
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b)
    for k, p in rb.named_parameters():
        assert torch.equal(grads["backbone." + k], p.grad), k
    for k, p in rf.named_parameters():
        assert torch.equal(grads["neck." + k], p.grad), k
    man["r50_fpn_small"] = {"x_seed": 700, "cot_seed0": 710, "state_seeds": [50, 51],
                            "out_sum": [float(o.double().sum()) for o in outs],
                            "grad_abssum_conv1": float(rb.conv1.weight.grad.double().abs().sum())}

    man["torch_version"] = torch.__version__
    with open(os.path.join(GOLD, "manifest.json"), "w") as fh:
        json.dump(man, fh, indent=1, sort_keys=True)
    print("golden fixtures written to", GOLD)
    for fn in sorted(os.listdir(GOLD)):
        print("  %-24s %8.1f KB" % (fn, os.path.getsize(os.path.join(GOLD, fn)) / 1024))


if __name__ == "__main__":
    main()
