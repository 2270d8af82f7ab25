"""GPU parity of the drop-in modules (ResNet / blocks / FPN) against golden vectors captured from the
reference import (tests/golden/, oracle/gen_golden.py) and against the CPU oracle.

Tolerances.  The HIP path stores every activation and activation-gradient in bf16 (8 significant bits,
unit roundoff 2^-9 ~ 2e-3) while the golden vectors are fp32 end to end, so module-level agreement is
bounded by accumulated bf16 storage rounding, not by the kernels (which meet 1e-3 per layer on identical
operands: tests/test_gpu_kernels.py).  Gradients additionally pass through ReLU masks: a pre-activation
within the forward error of zero (a fraction p ~ 1e-3 of the elements) flips its mask and changes that
gradient element by O(1), i.e. a relative-L2 error ~ sqrt(p) ~ 3e-2 that no arithmetic can avoid once
activations are bf16.  A schedule bug (a missing residual / FPN / stage gradient, a wrong tap) shows up as
>= 3e-1.  Bounds used here (relative L2):
   vs golden fp32 (reference):   block / FPN forward <= 6e-3, block gradients <= 1e-1, FPN gradients (no
                                 ReLU) <= 1e-2, ResNet-18 config-1 forward (C2..C5) <= 1.5e-2, R-FPN fwd <= 2e-2
   vs oracle/sched_ref.py (same schedule, bf16 rounding at the same storage points, fp32 CPU arithmetic), in
   situ / teacher-forced (tests/parity_util.py): each fused forward launch <= 1e-3, every parameter gradient of
   the whole backward <= 3e-2
Measured values are appended to gpurun_out/parity_models.json when that directory exists.
"""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import det_tensor, fill_state_dict, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available()
    import torch_detection_amd as t
    return t


@pytest.fixture(scope="module")
def manifest(golden_dir):
    with open(os.path.join(golden_dir, "manifest.json")) as fh:
        return json.load(fh)


def _f32(t):
    return t.detach().float().cpu()


def _record(key, value):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        f = os.path.join(d, "parity_models.json")
        data = json.load(open(f)) if os.path.exists(f) else {}
        data[key] = value
        json.dump(data, open(f, "w"), indent=1, sort_keys=True)


def test_blocks_vs_golden(T, manifest, golden_dir):
    from torch_detection_amd.backbone.resnet import _make_res_layer
    gold = np.load(os.path.join(golden_dir, "blocks.npz"))
    for name, meta in sorted(manifest["blocks"].items()):
        cls = getattr(T, meta["cls"])
        blk = _make_res_layer(cls, meta["inplanes"], meta["planes"], 1, stride=meta["stride"])[0]
        keys = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in blk.state_dict().items()]
        assert keys == meta["state_keys"], name
        blk.load_state_dict(fill_state_dict(blk.state_dict(), meta["state_seed"]))
        blk.cuda().eval()
        x = det_tensor(tuple(meta["x_shape"]), meta["x_seed"], -1, 1).cuda().requires_grad_(True)
        y = blk(x)
        assert y.shape == gold[name + "/y"].shape and y.dtype == torch.bfloat16
        dy = det_tensor(tuple(y.shape), meta["dy_seed"], -1, 1).cuda()
        y.backward(dy.to(y.dtype))
        ey = rel_l2(_f32(y), torch.from_numpy(gold[name + "/y"]))
        edx = rel_l2(_f32(x.grad), torch.from_numpy(gold[name + "/dx"]))
        eg = {}
        for k, p in blk.named_parameters():
            g = torch.from_numpy(gold[name + "/grad/" + k])
            assert p.grad is not None and p.grad.shape == g.shape, (name, k)
            eg[k] = rel_l2(_f32(p.grad), g)
        _record("block/" + name, {"y": ey, "dx": edx, "grad_max": max(eg.values())})
        assert ey <= 6e-3, (name, ey)
        assert edx <= 1e-1, (name, edx)
        assert max(eg.values()) <= 1e-1, (name, eg)


def test_fpn_vs_golden(T, manifest, golden_dir):
    meta = manifest["fpn_small"]
    gold = np.load(os.path.join(golden_dir, "fpn.npz"))
    fpn = T.FPN(meta["in_channels"], meta["out_channels"], meta["num_outs"])
    fpn.load_state_dict(fill_state_dict(fpn.state_dict(), meta["state_seed"]))
    fpn.cuda()
    ins = [det_tensor((meta["N"], c, h, w), meta["in_seed0"] + i, -1, 1).cuda().requires_grad_(True)
           for i, (c, (h, w)) in enumerate(zip(meta["in_channels"], meta["sizes"]))]
    outs = fpn(ins)
    assert isinstance(outs, tuple) and len(outs) == meta["num_outs"]
    assert [list(o.shape) for o in outs] == meta["out_shapes"]
    cots = [det_tensor(tuple(o.shape), meta["cot_seed0"] + i, -1, 1).cuda().to(o.dtype) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, cots)
    eo = [rel_l2(_f32(o), torch.from_numpy(gold["out%d" % i])) for i, o in enumerate(outs)]
    ei = [rel_l2(_f32(t.grad), torch.from_numpy(gold["din%d" % i])) for i, t in enumerate(ins)]
    eg = {k: rel_l2(_f32(p.grad), torch.from_numpy(gold["grad/" + k])) for k, p in fpn.named_parameters()}
    _record("fpn_small", {"out": eo, "din": ei, "grad_max": max(eg.values())})
    assert max(eo) <= 6e-3, eo
    assert max(ei) <= 1e-2, ei
    assert max(eg.values()) <= 1e-2, eg


def test_fpn_shape_errors(T, manifest):
    assert manifest["fpn_odd_size_error"] == "RuntimeError" and manifest["fpn_wrong_len_error"] == "AssertionError"
    fpn = T.FPN([64, 128, 256, 512], 64, 5).cuda()
    bad = [det_tensor((1, c, h, w), 1).cuda() for c, (h, w) in
           zip([64, 128, 256, 512], [(16, 24), (8, 11), (4, 6), (2, 3)])]
    with pytest.raises(RuntimeError):
        fpn(bad)
    with pytest.raises(AssertionError):
        fpn(bad[:3])


def test_resnet18_config1_vs_golden(T, manifest, golden_dir):
    """BASELINE config 1 (ResNet-18, 1x3x224x224): HIP path vs the reference's own CPU output."""
    meta = manifest["resnet18_c1"]
    gold = np.load(os.path.join(golden_dir, "resnet18_c1.npz"))
    m = T.ResNet(18)
    m.load_state_dict(fill_state_dict(m.state_dict(), meta["state_seed"]))
    m.cuda().train()
    i = meta["input"]
    x = det_tensor(tuple(i["shape"]), i["seed"], i["lo"], i["hi"]).cuda()
    with torch.no_grad():
        outs = m(x)
    assert [list(o.shape) for o in outs] == meta["out_shapes"]
    errs = [rel_l2(_f32(o), torch.from_numpy(gold["c%d" % (k + 2)])) for k, o in enumerate(outs)]
    _record("resnet18_c1", errs)
    assert max(errs) <= 1.5e-2, errs


@pytest.mark.parametrize("depth,shape", [(50, (2, 3, 128, 192)), (18, (1, 3, 64, 128)), (101, (1, 3, 64, 64))])
def test_resnet_fpn_fwd_bwd_in_situ(T, depth, shape):
    """ResNet + FPN forward/backward on a small padded image, checked launch by launch against the CPU schedule
    oracle (tests/parity_util.py explains why end-to-end bf16 comparisons cannot be tight)."""
    import parity_util
    res = parity_util.run_teacher_forced(T, depth, shape)
    _record("r%d_fpn_%dx%d" % (depth, shape[2], shape[3]), res)
    parity_util.check(res, depth)


def test_resnet_api_semantics(T, manifest):
    sem = manifest["train_semantics"]
    m = T.ResNet(18, out_indices=(3,)).cuda()
    assert m.train() is m  # documented deviation: the reference returns None
    assert sem["all_bn_eval_after_train"] and all(not b.training for b in m.modules()
                                                  if isinstance(b, torch.nn.BatchNorm2d))
    assert all(p.requires_grad for p in m.parameters()) == sem["all_params_require_grad"]
    assert {str(d): T.ResNet(d).feat_dim for d in (18, 50, 101)} == sem["feat_dim"]
    y = m(det_tensor((1, 3, 64, 64), 3).cuda())
    assert torch.is_tensor(y) and tuple(y.shape) == (1, 512, 2, 2)
    with pytest.raises(KeyError):
        T.ResNet(20)
    with pytest.raises(TypeError):
        T.ResNet(18).init_weights(pretrained=3)
    mt = T.ResNet(18, bn_eval=False).cuda().train()     # BatchNorm with batch statistics runs on the HIP path too
    mt.bn1.momentum = None                              # ... except cumulative averaging
    with pytest.raises(NotImplementedError):
        mt(det_tensor((2, 3, 64, 64), 3).cuda())
    with pytest.raises(RuntimeError):
        T.ResNet(18)(det_tensor((1, 3, 64, 64), 3))  # CPU tensors: no fallback
    # frozen stages: evident intent of resnet.py:281-294
    fz = T.ResNet(18, frozen_stages=1).cuda().train()
    assert not fz.conv1.weight.requires_grad and not fz.layer1[0].conv1.weight.requires_grad
    assert fz.layer2[0].conv1.weight.requires_grad


def test_checkpoint_roundtrip(T, tmp_path):
    m = T.ResNet(18)
    m.load_state_dict(fill_state_dict(m.state_dict(), 7))
    f = str(tmp_path / "r18.pth")
    T.save_checkpoint(m, f)
    ck = torch.load(f, weights_only=True)
    assert all(v.is_contiguous() for v in ck["state_dict"].values())
    m2 = T.ResNet(18)
    m2.init_weights(pretrained=f)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    m2.cuda().train()
    m.cuda().train()
    x = det_tensor((1, 3, 64, 96), 9).cuda()
    with torch.no_grad():
        for a, b in zip(m(x), m2(x)):
            assert torch.equal(a, b)


def test_weight_update_is_picked_up(T):
    """Packed bf16 weights are cached by parameter version: an in-place update must invalidate them."""
    blk = T.ConvModule(64, 64, 3, padding=1).cuda()
    x = det_tensor((1, 64, 8, 8), 5).cuda()
    with torch.no_grad():
        y0 = blk(x).clone()
        blk.conv.weight.mul_(2.0)
        blk.conv.bias.zero_()
        y1 = blk(x)
    assert not torch.equal(y0, y1)


@pytest.mark.parametrize("tag,actv", [("none", None), ("relu", "relu"), ("relu6", "relu6")])
def test_pafpn_vs_golden(T, manifest, golden_dir, tag, actv):
    """PAFPN on the HIP path vs vectors captured from the reference import.  The relu6 case feeds
    inputs four times as large, so that a visible share of the PA activations sits at the clamp (fractions in the
    manifest): the backward mask 0 < y < 6 of nn.ReLU6 (layers.py:117-118) is exercised, not just y > 0."""
    meta = manifest["pafpn_small"]
    gold = np.load(os.path.join(golden_dir, "pafpn.npz"))
    mod = T.PAFPN(meta["in_channels"], meta["out_channels"], meta["num_outs"], activation=actv)
    mod.load_state_dict(fill_state_dict(mod.state_dict(), meta["state_seed"]))
    mod.cuda()
    amp = 4.0 if actv == "relu6" else 1.0
    ins = [det_tensor((meta["N"], c, h, w), meta["in_seed0"] + i, -amp, amp).cuda().requires_grad_(True)
           for i, (c, (h, w)) in enumerate(zip(meta["in_channels"], meta["sizes"]))]
    outs = mod(ins)
    assert len(outs) == meta["num_outs"]
    if actv == "relu6":
        assert max(float((o >= 6).float().mean()) for o in outs[1:4]) > 0.01
    cots = [det_tensor(tuple(o.shape), meta["cot_seed0"] + i, -1, 1).cuda().to(o.dtype) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, cots)
    eo = [rel_l2(_f32(o), torch.from_numpy(gold["%s/out%d" % (tag, i)])) for i, o in enumerate(outs)]
    ei = [rel_l2(_f32(t.grad), torch.from_numpy(gold["%s/din%d" % (tag, i)])) for i, t in enumerate(ins)]
    eg = {}
    for k, p in mod.named_parameters():
        key = "%s/grad/%s" % (tag, k)
        if key in gold.files:
            eg[k] = rel_l2(_f32(p.grad), torch.from_numpy(gold[key]))
    _record("pafpn_small/" + tag, {"out": eo, "din": ei, "grad_max": max(eg.values())})
    assert max(eo) <= 8e-3, eo
    # without an activation the path is linear: tight.  With ReLU the masks of tiny maps (2x3 .. 16x24) flip on a
    # visible fraction of elements (module docstring): measured 3e-2 .. 1.3e-1, bound 2.5e-1 (a routing bug is O(1)
    # and is already excluded by the linear case, which shares all the code but the masks)
    tol = 1.5e-2 if actv is None else 2.5e-1
    assert max(ei) <= tol, ei
    assert max(eg.values()) <= tol, eg


def test_groupnorm_variants_vs_golden(T, manifest, golden_dir):
    """use_gn=True residual blocks, GN FPN and GN ResNet-18 against the golden vectors captured
    from the reference.  Bounds as for the BN variants; GroupNorm re-normalises every layer, so the end-to-end bf16
    distance stays at the 1e-2 level (forward 1.5e-2, gradients 1e-1 like the BN blocks)."""
    from torch_detection_amd.backbone.resnet import _make_res_layer
    gold = np.load(os.path.join(golden_dir, "gn.npz"))
    for name, meta in sorted(manifest["gn_blocks"].items()):
        blk = _make_res_layer(getattr(T, meta["cls"]), meta["inplanes"], meta["planes"], 1, stride=meta["stride"],
                              use_gn=True)[0]
        blk.load_state_dict(fill_state_dict(blk.state_dict(), meta["state_seed"]))
        blk.cuda()
        x = det_tensor(tuple(meta["x_shape"]), meta["x_seed"], -1, 1).cuda().requires_grad_(True)
        y = blk(x)
        assert y.shape == gold["blk/%s/y" % name].shape and y.dtype == torch.bfloat16
        y.backward(det_tensor(tuple(y.shape), meta["dy_seed"], -1, 1).cuda().to(y.dtype))
        ey = rel_l2(_f32(y), torch.from_numpy(gold["blk/%s/y" % name]))
        edx = rel_l2(_f32(x.grad), torch.from_numpy(gold["blk/%s/dx" % name]))
        eg = {k: rel_l2(_f32(p.grad), torch.from_numpy(gold["blk/%s/grad/%s" % (name, k)]))
              for k, p in blk.named_parameters()}
        _record("gn_block/" + name, {"y": ey, "dx": edx, "grad_max": max(eg.values())})
        assert ey <= 1.5e-2, (name, ey)
        assert edx <= 1e-1, (name, edx)
        assert max(eg.values()) <= 1e-1, (name, eg)
    meta = manifest["gn_fpn_small"]
    fpn = T.FPN(meta["in_channels"], meta["out_channels"], meta["num_outs"], normalize=dict(type="GN"), use_gn=True)
    fpn.load_state_dict(fill_state_dict(fpn.state_dict(), meta["state_seed"]))
    fpn.cuda()
    ins = [det_tensor((meta["N"], c, h, w), meta["in_seed0"] + i, -1, 1).cuda().requires_grad_(True)
           for i, (c, (h, w)) in enumerate(zip(meta["in_channels"], meta["sizes"]))]
    outs = fpn(ins)
    torch.autograd.backward(outs, [det_tensor(tuple(o.shape), meta["cot_seed0"] + i, -1, 1).cuda().to(o.dtype)
                                   for i, o in enumerate(outs)])
    eo = [rel_l2(_f32(o), torch.from_numpy(gold["fpn/out%d" % i])) for i, o in enumerate(outs)]
    ei = [rel_l2(_f32(t.grad), torch.from_numpy(gold["fpn/din%d" % i])) for i, t in enumerate(ins)]
    eg = {k: rel_l2(_f32(p.grad), torch.from_numpy(gold["fpn/grad/" + k])) for k, p in fpn.named_parameters()}
    _record("gn_fpn_small", {"out": eo, "din": ei, "grad_max": max(eg.values())})
    assert max(eo) <= 1.5e-2, eo
    assert max(ei) <= 5e-2, ei
    assert max(eg.values()) <= 5e-2, eg
    meta = manifest["gn_resnet18"]
    m = T.ResNet(18, use_gn=True)
    m.load_state_dict(fill_state_dict(m.state_dict(), meta["state_seed"]))
    m.cuda().train()
    i = meta["input"]
    x = det_tensor(tuple(i["shape"]), i["seed"], i["lo"], i["hi"]).cuda()
    outs = m(x)
    assert [list(o.shape) for o in outs] == meta["out_shapes"]
    errs = [rel_l2(_f32(o), torch.from_numpy(gold["r18/c%d" % (k + 2)])) for k, o in enumerate(outs)]
    _record("gn_resnet18", errs)
    assert max(errs) <= 2e-2, errs
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])    # stem GN + max-pool adjoint path runs
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_resnext_vs_golden(T, manifest, golden_dir):
    """ResNeXt grouped-conv bottlenecks (fwd + all grads) and ResNeXt-50 32x4d forward + a backward pass, against
    golden vectors from the reference (tests/golden/resnext.npz).  Bounds as for the ResNet blocks."""
    from torch_detection_amd.backbone.resnext import ResNeXtBottleneck, _make_resX_layer
    gold = np.load(os.path.join(golden_dir, "resnext.npz"))
    for name, meta in sorted(manifest["resnext_blocks"].items()):
        blk = _make_resX_layer(ResNeXtBottleneck, meta["inplanes"], meta["planes"], 1, meta["base_width"],
                               meta["cardinality"], stride=meta["stride"])[0]
        blk.load_state_dict(fill_state_dict(blk.state_dict(), meta["state_seed"]))
        blk.cuda().eval()
        x = det_tensor(tuple(meta["x_shape"]), meta["x_seed"], -1, 1).cuda().requires_grad_(True)
        y = blk(x)
        assert y.shape == gold["blk/%s/y" % name].shape and y.dtype == torch.bfloat16
        y.backward(det_tensor(tuple(y.shape), meta["dy_seed"], -1, 1).cuda().to(y.dtype))
        ey = rel_l2(_f32(y), torch.from_numpy(gold["blk/%s/y" % name]))
        edx = rel_l2(_f32(x.grad), torch.from_numpy(gold["blk/%s/dx" % name]))
        eg = {}
        for k, p in blk.named_parameters():
            g = torch.from_numpy(gold["blk/%s/grad/%s" % (name, k)])
            assert p.grad is not None and p.grad.shape == g.shape, (name, k)
            eg[k] = rel_l2(_f32(p.grad), g)
        _record("resnext_block/" + name, {"y": ey, "dx": edx, "grad_max": max(eg.values())})
        assert ey <= 6e-3, (name, ey)
        assert edx <= 1e-1, (name, edx)
        assert max(eg.values()) <= 1e-1, (name, eg)
    meta = manifest["resnext50_32x4d"]
    m = T.BACKBONES.module_dict["ResNeXt"](50, 4, 32)
    m.load_state_dict(fill_state_dict(m.state_dict(), meta["state_seed"]))
    m.cuda().train()
    i = meta["input"]
    outs = m(det_tensor(tuple(i["shape"]), i["seed"], i["lo"], i["hi"]).cuda())
    assert [list(o.shape) for o in outs] == meta["out_shapes"]
    errs = [rel_l2(_f32(o), torch.from_numpy(gold["x50/c%d" % (k + 2)])) for k, o in enumerate(outs)]
    _record("resnext50_32x4d", errs)
    assert max(errs) <= 2e-2, errs
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    w = m.layer1[0].conv2.weight
    assert w.grad.shape == w.shape and w.grad.stride() == w.stride()     # layout contract: no copy in AccumulateGrad


def test_resnext_use_gn_keeps_training_mode_bn_in_downsample(T):
    """Reference quirk (resnext.py:147,303-310): a use_gn=True ResNeXt has BatchNorm in its downsample branches and
    leaves it in training mode — GroupNorm units and batch-statistics BN units in one net, forward and backward."""
    m = T.ResNeXt(50, 4, 32, use_gn=True).cuda().train()
    m.init_weights()
    ds = [mod for name, mod in m.named_modules() if name.endswith("downsample.1")]
    assert ds and all(isinstance(d, torch.nn.BatchNorm2d) and d.training for d in ds)
    before = ds[0].running_mean.clone()
    outs = m(det_tensor((2, 3, 64, 96), 13, -1, 1).cuda())
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    assert all(bool(torch.isfinite(o).all()) for o in outs)
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    assert not torch.equal(ds[0].running_mean, before) and int(ds[0].num_batches_tracked) == 1


def test_graphed_step_matches_eager(T):
    """GraphedStep: one captured forward+backward replayed == the eager step, bit for bit (static tensors)."""
    m = T.ResNet(18).cuda().train()
    m.init_weights()
    neck = T.FPN([64, 128, 256, 512], 256, 5).cuda()
    neck.init_weights()
    x = det_tensor((2, 3, 64, 96), 9, -1, 1).cuda()
    with torch.no_grad():   # shape probe only: an autograd graph built on the default stream must not outlive this
        outs = neck(m(x))   # line (its AccumulateGrad nodes would tie the captured backward to the default stream)
    cots = [det_tensor(tuple(o.shape), 20 + i, -1, 1).cuda().to(o.dtype) for i, o in enumerate(outs)]
    del outs
    params = list(m.parameters()) + list(neck.parameters())
    held = {}

    def step():
        for p in params:
            p.grad = None
        o = neck(m(x))
        torch.autograd.backward(o, cots)
        held["outs"] = o

    gs = T.GraphedStep(step)     # capture first (warm-up runs on a side stream), eager reference afterwards
    assert gs.captured, gs.error
    for _ in range(2):
        gs()
    torch.cuda.synchronize()
    got_o = [t.clone() for t in held["outs"]]
    got_g = [p.grad.clone() for p in params]
    step()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(held["outs"], got_o))
    assert all(torch.equal(p.grad, g) for p, g in zip(params, got_g))


def _small_net(T):
    m = T.ResNet(18).cuda().train()
    m.init_weights()
    neck = T.FPN([64, 128, 256, 512], 256, 5).cuda()
    neck.init_weights()
    x = det_tensor((2, 3, 64, 96), 9, -1, 1).cuda()
    return m, neck, x, list(m.parameters()) + list(neck.parameters())


def test_graphed_step_refuses_live_outside_graph(T):
    """An ordinary grad-mode forward whose outputs are still alive keeps the parameters' AccumulateGrad nodes bound
    to the default stream; capturing then used to crash inside capture_end.  GraphedStep(params=...) must detect
    it, not capture, and run eagerly with the right results; once the outputs are gone the capture works."""
    m, neck, x, params = _small_net(T)
    outs = neck(m(x))                     # grad mode, outputs held
    cots = [det_tensor(tuple(o.shape), 20 + i, -1, 1).cuda().to(o.dtype) for i, o in enumerate(outs)]

    def step():
        for p in params:
            p.grad = None
        torch.autograd.backward(neck(m(x)), cots)

    gs = T.GraphedStep(step, params=params, verbose=False)
    assert not gs.captured and "autograd graph" in str(gs.error)
    gs()                                  # eager fallback
    torch.cuda.synchronize()
    eager = [p.grad.clone() for p in params]
    del outs
    gs2 = T.GraphedStep(step, params=params)
    assert gs2.captured, gs2.error
    gs2()
    torch.cuda.synchronize()
    assert all(torch.equal(p.grad, g) for p, g in zip(params, eager))


@pytest.mark.parametrize("how", ["inplace", "data_copy"])
def test_graph_replay_sees_weight_update(T, how):
    """replay -> SGD update OUTSIDE the captured callable -> replay must equal the eager step on the updated weights:
    the pack / BN-fold kernels are part of the graph (repack=True) and read the live fp32 parameters."""
    m, neck, x, params = _small_net(T)
    with torch.no_grad():
        outs = neck(m(x))
    cots = [det_tensor(tuple(o.shape), 20 + i, -1, 1).cuda().to(o.dtype) for i, o in enumerate(outs)]
    del outs
    held = {}

    def step():
        for p in params:
            p.grad = None
        o = neck(m(x))
        torch.autograd.backward(o, cots)
        held["o"] = o

    gs = T.GraphedStep(step, params=params)
    assert gs.captured, gs.error
    gs()
    torch.cuda.synchronize()
    before = [t.clone() for t in held["o"]]
    with torch.no_grad():                 # the update an optimizer would make, outside the graph
        for i, p in enumerate(params):
            upd = p * (1.0 + 0.01 * ((i % 3) - 1)) + 0.01
            if how == "inplace":
                p.copy_(upd)
            else:
                p.data.copy_(upd)         # does not bump p._version: only the in-graph repack can see it
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.add_(0.05)
    gs()
    torch.cuda.synchronize()
    got_o = [t.clone() for t in held["o"]]
    got_g = [p.grad.clone() for p in params]
    assert not all(torch.equal(a, b) for a, b in zip(got_o, before))
    T.invalidate_packed(m, neck)          # eager reference: needed for the data_copy case (no version bump)
    step()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(held["o"], got_o))
    assert all(torch.equal(p.grad, g) for p, g in zip(params, got_g))


def test_prepared_step_matches_eager(T):
    """PreparedStep (the launch plan of libtdn: one C call enqueues the recorded forward + backward, side streams and
    their event dependencies included) == the eager step, bit for bit, repeatedly; it sees weight updates because the
    fold / pack launches are part of the plan; and a step that does GPU work outside the library is refused."""
    m, neck, x, params = _small_net(T)
    with torch.no_grad():
        outs = neck(m(x))
    cots = [det_tensor(tuple(o.shape), 20 + i, -1, 1).cuda().to(o.dtype) for i, o in enumerate(outs)]
    del outs
    held = {}

    def step():
        for p in params:
            p.grad = None
        o = neck(m(x))
        torch.autograd.backward(o, cots)
        held["o"] = o

    ps = T.PreparedStep(step, params=params, modules=(m, neck))
    assert ps.prepared, ps.error
    nl, ne, nw = ps.stats()
    assert nl > 60 and ne > 0 and nw > 0
    ps()
    ps()
    torch.cuda.synchronize()
    got_o = [t.detach().clone() for t in held["o"]]
    got_g = [p.grad.clone() for p in params]
    with torch.no_grad():
        for i, p in enumerate(params):
            p.mul_(1.0 + 0.01 * ((i % 3) - 1))
    ps()
    torch.cuda.synchronize()
    new_o = [t.detach().clone() for t in held["o"]]
    new_g = [p.grad.clone() for p in params]
    assert not all(torch.equal(a, b) for a, b in zip(new_o, got_o))
    ps.close()
    held.clear()
    step()                                 # eager, on the updated weights
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(held["o"], new_o))
    assert all(torch.equal(p.grad, g) for p, g in zip(params, new_g))

    def impure():
        step()
        params[0].grad.mul_(2.0)           # a PyTorch kernel the plan cannot contain

    bad = T.PreparedStep(impure, params=params, modules=(m, neck), verbose=False)
    assert not bad.prepared and "outside the library" in str(bad.error)


@pytest.mark.parametrize("depth", [18, 50])
def test_branch_streams_change_nothing(T, depth, monkeypatch):
    """functional.branch moves independent launches (downsample conv / dgrad, coarse FPN output convs, lateral dgrads)
    to streams of their own: outputs and every gradient must equal the single-stream run bit for bit, repeatedly
    (a missing dependency would show up as run-to-run differences)."""
    m = T.ResNet(depth).cuda().train()
    m.init_weights()
    chans = [64, 128, 256, 512] if depth == 18 else [256, 512, 1024, 2048]
    neck = T.FPN(chans, 256, 5).cuda()
    neck.init_weights()
    x = det_tensor((2, 3, 96, 160), 11, -1, 1).cuda()
    params = list(m.parameters()) + list(neck.parameters())

    def run():
        for p in params:
            p.grad = None
        o = neck(m(x))
        torch.autograd.backward(o, [det_tensor(tuple(t.shape), 30 + i, -1, 1).cuda().to(t.dtype)
                                    for i, t in enumerate(o)])
        torch.cuda.synchronize()
        return [t.clone() for t in o], [p.grad.clone() for p in params]

    monkeypatch.setenv("TDN_BRANCH", "0")
    ref_o, ref_g = run()
    monkeypatch.setenv("TDN_BRANCH", "2")
    for _ in range(3):
        o, g = run()
        assert all(torch.equal(a, b) for a, b in zip(o, ref_o))
        assert all(torch.equal(a, b) for a, b in zip(g, ref_g))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_conv_module_relu6_and_preactivation_vs_golden(T, manifest, golden_dir, dtype):
    """ConvModule on the HIP path with activation='relu6' and with activate_last=False (norm -> activation -> conv;
    BatchNorm2d in eval and in training mode, GroupNorm, no norm) against the reference golden: output, input
    gradient, every parameter gradient, running statistics."""
    import warnings
    meta = manifest["conv_module"]
    gold = np.load(os.path.join(golden_dir, "conv_module.npz"))
    i = meta["input"]
    for c in meta["cases"]:
        tag, k = c["tag"], c["kernel"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = T.ConvModule(64, 64, k, padding=k // 2, bias=c["bias"], normalize=dict() if c["normalize"] else None,
                             use_gn=c["use_gn"], activation=c["activation"], activate_last=c["activate_last"])
        m.load_state_dict(fill_state_dict(m.state_dict(), c["state_seed"]))
        m.cuda().train(c["training"])
        x = det_tensor(tuple(i["shape"]), c["input_seed"], i["lo"], i["hi"]).cuda().requires_grad_(True)
        m.compute_dtype = dtype
        y = m(x)
        assert y.dtype == dtype
        y.backward(det_tensor(tuple(y.shape), c["cot_seed"], -1, 1).cuda().to(y.dtype))
        errs = {"y": rel_l2(_f32(y), torch.from_numpy(gold[tag + "/y"])),
                "dx": rel_l2(_f32(x.grad), torch.from_numpy(gold[tag + "/dx"]))}
        for k_, p in m.named_parameters():
            errs[k_] = rel_l2(p.grad.float().cpu(), torch.from_numpy(gold[tag + "/grad/" + k_]))
        if c["training"] and c["bias"] and c["activate_last"] and c["normalize"] and not c["use_gn"]:
            # batch statistics cancel a bias in front of them: its gradient is zero up to round-off on both sides
            # (the reference's is ~1e-7 noise), so compare magnitudes against the norm's own bias gradient instead
            scale = float(m.norm.bias.grad.abs().max())
            assert float(m.conv.bias.grad.abs().max()) <= 1e-3 * scale
            assert float(np.abs(gold[tag + "/grad/conv.bias"]).max()) <= 1e-3 * scale
            errs.pop("conv.bias")
        _record("conv_module_%s_%s" % (tag, "f16" if dtype == torch.float16 else "bf16"), [errs[k_] for k_ in sorted(errs)])
        # bf16 activations (2^-9 per stored value: measured <= 2.4e-3).  The kernels keep values under 6 under 6 when
        # they store them (relu6_top), so the backward's mask 0 < y < 6 agrees with the fp32 reference element for
        # element; rounding (5.984, 6) up to 6.0 instead cost 4e-2 on these gradients.
        assert all(v <= 5e-3 for v in errs.values()), (tag, errs)
        if c["training"]:
            for k_ in ("norm.running_mean", "norm.running_var"):
                got = dict(m.named_buffers())[k_].float().cpu()
                assert rel_l2(got, torch.from_numpy(gold[tag + "/stat/" + k_])) <= 1e-3, (tag, k_)
            assert int(m.norm.num_batches_tracked) == 1


def test_dilated_resnet_vs_golden(T, manifest, golden_dir):
    """ResNet(strides=(1,2,1,1), dilations=(1,1,2,4)) forward against the reference golden (R18) and the oracle (R50),
    plus a backward pass through the dilated stages."""
    from oracle import torch_ref as O
    gold = np.load(os.path.join(golden_dir, "dilated.npz"))
    for d in (18, 50):
        meta = manifest["resnet%d_dilated" % d]
        m = T.ResNet(d, strides=tuple(meta["strides"]), dilations=tuple(meta["dilations"]))
        sd = fill_state_dict(m.state_dict(), meta["state_seed"])
        m.load_state_dict(sd)
        m.cuda().train()
        i = meta["input"]
        x = det_tensor(tuple(i["shape"]), i["seed"], i["lo"], i["hi"])
        outs = m(x.cuda())
        assert [list(o.shape) for o in outs] == meta["out_shapes"]
        if d == 18:
            refs = [torch.from_numpy(gold["r18/c%d" % (k + 2)]) for k in range(4)]
        else:
            with torch.no_grad():
                refs = O.resnet_forward(sd, x, d, strides=tuple(meta["strides"]), dilations=tuple(meta["dilations"]))
        errs = [rel_l2(_f32(o), r) for o, r in zip(outs, refs)]
        _record("resnet%d_dilated" % d, errs)
        assert max(errs) <= 2e-2, errs
        torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_bn_training_mode_vs_golden(T, manifest, golden_dir):
    """ResNet(18, bn_eval=False).train() — BatchNorm with batch statistics on the HIP path — against the reference
    golden: outputs, sampled parameter gradients, running statistics and num_batches_tracked after one step."""
    meta = manifest["bn_train_r18"]
    gold = np.load(os.path.join(golden_dir, "bn_train.npz"))
    m = T.ResNet(18, bn_eval=False)
    m.load_state_dict(fill_state_dict(m.state_dict(), meta["state_seed"]))
    m.cuda().train()
    i = meta["input"]
    outs = m(det_tensor(tuple(i["shape"]), i["seed"], i["lo"], i["hi"]).cuda())
    assert [list(o.shape) for o in outs] == meta["out_shapes"]
    eo = [rel_l2(_f32(o), torch.from_numpy(gold["c%d" % (k + 2)])) for k, o in enumerate(outs)]
    torch.autograd.backward(outs, [det_tensor(tuple(o.shape), meta["cot_seed0"] + k, -1, 1).cuda().to(o.dtype)
                                   for k, o in enumerate(outs)])
    ps = dict(m.named_parameters())
    eg = {k: rel_l2(_f32(ps[k].grad), torch.from_numpy(gold["grad/" + k])) for k in meta["grad_keys"]}
    sd = m.state_dict()
    es = {k: rel_l2(sd[k].float().cpu(), torch.from_numpy(gold["stat/" + k]))
          for k in ("bn1.running_mean", "bn1.running_var", "layer4.1.bn2.running_mean", "layer4.1.bn2.running_var")}
    _record("bn_train_r18", {"out": eo, "grad": eg, "stat": es})
    # batch statistics over very few elements (layer4 of this 64x96 input: 2 images x 2 x 3 pixels per channel) turn a
    # bf16 rounding of one conv output into a visible change of mean / rstd: the forward distance to the fp32 reference
    # grows with depth (measured 0.7 % at C2 ... 4 % at C5; eval-mode BN: 0.3 ... 0.5 %)
    assert eo[0] <= 1.5e-2 and max(eo) <= 8e-2, eo
    assert es["bn1.running_mean"] <= 1e-3 and es["bn1.running_var"] <= 1e-3 and max(es.values()) <= 3e-2, es
    assert int(sd["bn1.num_batches_tracked"]) == meta["num_batches_tracked_after"]
    # batch statistics couple every element of a channel: the bf16 decorrelation reaches the gradients in full, so
    # the end-to-end bound is the loose one of the other block tests; the kernels themselves are checked tightly in
    # tests/test_gpu_gn.py::test_bn_train_fwd_bwd
    assert max(eg.values()) <= 4e-1, eg
    # switching the same module to eval mode re-folds BN (packed operands are keyed on the mode)
    m.eval()
    with torch.no_grad():
        o_eval = m(det_tensor(tuple(i["shape"]), i["seed"], i["lo"], i["hi"]).cuda())
    assert all(bool(torch.isfinite(o).all()) for o in o_eval)


def test_pafpn_extra_convs_vs_golden(T, manifest, golden_dir):
    """PAFPN(add_extra_convs=True, num_outs=6): stride-2 conv levels on the last backbone input (pafpn.py:139-147),
    including the reference's in-place ReLU on P6 (the returned P6 is rectified), against the reference golden."""
    meta = manifest["pafpn_extra"]
    gold = np.load(os.path.join(golden_dir, "pafpn.npz"))
    pa = T.PAFPN(meta["in_channels"], meta["out_channels"], meta["num_outs"], add_extra_convs=True)
    keys = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in pa.state_dict().items()]
    assert keys == meta["state_keys"]
    pa.load_state_dict(fill_state_dict(pa.state_dict(), meta["state_seed"]))
    pa.cuda()
    ins = [det_tensor((meta["N"], c, h, w), meta["in_seed0"] + i, -1, 1).cuda().requires_grad_(True)
           for i, (c, (h, w)) in enumerate(zip(meta["in_channels"], meta["sizes"]))]
    outs = pa(ins)
    assert len(outs) == 6
    torch.autograd.backward(outs, [det_tensor(tuple(o.shape), meta["cot_seed0"] + i, -1, 1).cuda().to(o.dtype)
                                   for i, o in enumerate(outs)])
    eo = [rel_l2(_f32(o), torch.from_numpy(gold["extra/out%d" % i])) for i, o in enumerate(outs)]
    ei = [rel_l2(_f32(t.grad), torch.from_numpy(gold["extra/din%d" % i])) for i, t in enumerate(ins)]
    ps = dict(pa.named_parameters())
    eg = {k: rel_l2(_f32(ps[k].grad), torch.from_numpy(gold["extra/grad/" + k])) for k in meta["grad_keys"]}
    _record("pafpn_extra", {"out": eo, "din": ei, "grad": eg})
    assert max(eo) <= 1e-2, eo
    assert max(ei) <= 3e-2, ei
    assert max(eg.values()) <= 3e-2, eg
    assert float(outs[4].min()) >= 0.0     # P6 comes back rectified (in-place ReLU of the reference)


def test_fpn_extra_convs_vs_oracle(T):
    """FPN(add_extra_convs=True, num_outs=6) (RetinaNet levels, fpn.py:118-124) against oracle/torch_ref.py, whose
    extra-level code is pinned to the reference through the PAFPN golden (same lines, same in-place ReLU)."""
    from oracle import torch_ref as O
    chans, sizes = [64, 128, 256, 512], [(16, 24), (8, 12), (4, 6), (2, 3)]
    fpn = T.FPN(chans, 64, 6, add_extra_convs=True)
    sd = fill_state_dict(fpn.state_dict(), 860)
    fpn.load_state_dict(sd)
    fpn.cuda()
    ins = [det_tensor((2, c, h, w), 870 + i, -1, 1) for i, (c, (h, w)) in enumerate(zip(chans, sizes))]
    gins = [t.cuda().requires_grad_(True) for t in ins]
    outs = fpn(gins)
    cots = [det_tensor(tuple(o.shape), 880 + i, -1, 1) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, [c.cuda().to(o.dtype) for c, o in zip(cots, outs)])
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rins = [t.clone().requires_grad_(True) for t in ins]
    routs = O.fpn_forward(ps, rins, 6, add_extra_convs=True)
    torch.autograd.backward(routs, cots)
    eo = [rel_l2(_f32(o), r.detach()) for o, r in zip(outs, routs)]
    ei = [rel_l2(_f32(t.grad), r.grad) for t, r in zip(gins, rins)]
    eg = {k: rel_l2(_f32(p.grad), ps[k].grad) for k, p in fpn.named_parameters()}
    _record("fpn_extra", {"out": eo, "din": ei, "grad_max": max(eg.values())})
    assert max(eo) <= 1e-2 and max(ei) <= 3e-2 and max(eg.values()) <= 3e-2, (eo, ei, eg)
