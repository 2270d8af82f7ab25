"""GPU parity tests of the float16 operand path (BASELINE config C5: R101-FPN, fp16 operands, fp32 accumulate).

Same kernels as the bf16 path (templates on the element type), so this file covers what differs: the conversions,
the MFMA f16 opcode, the ones-fragment of the wgrad column sums, and the module-level dtype plumbing
(``x.half()`` in -> fp16 out; ``module.compute_dtype = torch.float16``).  Operands are made fp16-representable and the
fp32 CPU reference computes on exactly those values; tolerances:
  * fp32-output kernels: max|err| / max|ref| <= 1e-3 (same figure as bf16; measured ~1e-6)
  * fp16-output kernels: <= 1 fp16 ulp (2^-10 relative) of the fp32 reference
  * weight / affine gradients: rel-L2 <= 1e-3
  * ResNet+FPN in situ / teacher-forced: the bounds of tests/parity_util.py (fp16 has 3 more mantissa bits than bf16,
    so the measured numbers are ~8x smaller; recorded in gpurun_out/parity_models.json)
"""
import pytest
import torch
import torch.nn.functional as F

from golden_util import det_tensor, max_rel, rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-3
H16 = torch.float16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from torch_detection_amd import ops as _ops
    from torch_detection_amd import _lib
    _lib.load()
    return _ops


@pytest.fixture(scope="module")
def T():
    import torch_detection_amd as _T
    return _T


def q(t):  # make fp16-representable (on CPU, fp32 container)
    return t.half().float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().half().cuda()


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


CASES = [
    # N, H, W, Cin, Cout, k, stride
    (1, 20, 24, 64, 256, 1, 1),
    (2, 25, 42, 128, 128, 3, 2),
    (1, 13, 21, 128, 128, 3, 1),
    (2, 9, 11, 512, 512, 3, 1),     # few tiles, long K: in-workgroup split-K configuration
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_f16(ops, case):
    N, H, W, Cin, Cout, k, s = case
    x = q(det_tensor((N, Cin, H, W), 1, -1, 1, bf16=False))
    w = q(det_tensor((Cout, Cin, k, k), 2, -0.2, 0.2, bf16=False))
    scale = det_tensor((Cout,), 3, 0.5, 1.5, bf16=False)
    shift = det_tensor((Cout,), 4, -0.5, 0.5, bf16=False)
    ref = F.conv2d(x, w, None, s, k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = q(det_tensor(tuple(ref.shape), 5, -1, 1, bf16=False))
    ref_r = F.relu(ref + res)
    wf, wd = ops.pack_conv_weight(w.cuda(), scale.cuda(), True, H16)
    assert wf.dtype == H16 and torch.equal(wf.float().cpu(), w.permute(0, 2, 3, 1))
    y32 = ops.conv2d_fwd(nhwc(x), wf, k, s, k // 2, scale.cuda(), shift.cuda(), nhwc(res), ops.ADD_SAME, True,
                         out_f32=True)
    assert max_rel(nchw(y32), ref_r) <= TOL
    y16 = ops.conv2d_fwd(nhwc(x), wf, k, s, k // 2, scale.cuda(), shift.cuda(), nhwc(res), ops.ADD_SAME, True)
    assert y16.dtype == H16
    # one fp16 ulp of the reference + the fp32 accumulation noise (relative to the largest output, K up to 4608)
    assert bool(((nchw(y16) - ref_r).abs() <= ref_r.abs() * 2 ** -10 + 1e-5 * float(ref_r.abs().max())).all())
    # dgrad with the folded fp16(scale * fp16(w)) operand, + addend, ReLU mask
    Ho, Wo = ref.shape[2], ref.shape[3]
    g = q(det_tensor((N, Cout, Ho, Wo), 11, -1, 1, bf16=False))
    w_eff = wd.float().cpu().permute(3, 0, 1, 2).contiguous()
    # the fold fp16(scale * fp16(w)): the GPU may form it with one rounding (mixed-precision multiply) where the CPU
    # rounds twice (fp32 product, then fp16) — at most one fp16 ulp apart, and only on a ~1e-4 fraction of elements
    w_ref = q(w * scale.view(-1, 1, 1, 1))
    assert bool(((w_eff - w_ref).abs() <= w_ref.abs() * 2 ** -10 + 1e-7).all())
    assert float((w_eff != w_ref).float().mean()) <= 1e-3
    xz = torch.zeros(N, Cin, H, W, requires_grad=True)
    F.conv2d(xz, w_eff, None, s, k // 2).backward(g)
    add = q(det_tensor((N, Cin, H, W), 14, -1, 1, bf16=False))
    msk = q(det_tensor((N, Cin, H, W), 15, -1, 1, bf16=False))
    ref2 = (xz.grad + add) * (msk > 0).float()
    dx = ops.conv2d_dgrad(nhwc(g), wd, (H, W), k, s, k // 2, nhwc(add), ops.ADD_SAME, nhwc(msk), out_f32=True)
    assert max_rel(nchw(dx), ref2) <= TOL
    # mixing element types is refused on the host
    with pytest.raises(ValueError):
        ops.conv2d_fwd(nhwc(x).bfloat16(), wf, k, s, k // 2)


@pytest.mark.parametrize("case", CASES[:3])
def test_conv_wgrad_f16(ops, case):
    N, H, W, Cin, Cout, k, s = case
    x = q(det_tensor((N, Cin, H, W), 31, -1, 1, bf16=False))
    w = q(det_tensor((Cout, Cin, k, k), 33, -0.2, 0.2, bf16=False)).requires_grad_(True)
    gamma = det_tensor((Cout,), 34, 0.5, 1.5, bf16=False).requires_grad_(True)
    beta = det_tensor((Cout,), 35, -0.5, 0.5, bf16=False).requires_grad_(True)
    mean = det_tensor((Cout,), 36, -0.2, 0.2, bf16=False)
    var = det_tensor((Cout,), 37, 0.5, 1.5, bf16=False)
    y = F.batch_norm(F.conv2d(x, w, None, s, k // 2), mean, var, gamma, beta, False, 0.1, 1e-5)
    g = q(det_tensor(tuple(y.shape), 32, -1, 1, bf16=False))
    y.backward(g)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.detach() * invstd
    wf, _ = ops.pack_conv_weight(w.detach().cuda(), None, False, H16)
    dw, dg, db = ops.conv2d_wgrad(nhwc(x), nhwc(g), wf, k, s, k // 2, scale.cuda(), mean.cuda(), invstd.cuda())
    assert rel_l2(dw.cpu().permute(0, 3, 1, 2), w.grad) <= TOL
    assert rel_l2(dg.cpu(), gamma.grad) <= TOL
    assert rel_l2(db.cpu(), beta.grad) <= TOL      # the ones-fragment column sum


def test_stem_pool_elementwise_f16(ops):
    N, H, W = 2, 32, 48
    img = q(det_tensor((N, 3, H, W), 41, -2, 2, bf16=False))
    w = q(det_tensor((64, 3, 7, 7), 42, -0.2, 0.2, bf16=False)).requires_grad_(True)
    gamma = det_tensor((64,), 43, 0.5, 1.5, bf16=False).requires_grad_(True)
    beta = det_tensor((64,), 44, -0.5, 0.5, bf16=False).requires_grad_(True)
    mean = det_tensor((64,), 45, -0.2, 0.2, bf16=False)
    var = det_tensor((64,), 46, 0.5, 1.5, bf16=False)
    pre = F.batch_norm(F.conv2d(img, w, None, 2, 3), mean, var, gamma, beta, False, 0.1, 1e-5)
    xp = ops.stage_image(img.cuda(), H16)
    assert xp.dtype == H16 and torch.equal(xp.float().cpu()[:, 3:3 + H, 3:3 + W, :3], img.permute(0, 2, 3, 1))
    ws = ops.pack_stem_weight(w.detach().cuda(), H16)
    scale, shift, invstd = ops.bn_fold(gamma.detach().cuda(), beta.detach().cuda(), mean.cuda(), var.cuda(), 1e-5)
    y = ops.stem_conv_fwd(xp, ws, (H, W), scale, shift, True, out_f32=True)
    assert max_rel(nchw(y), F.relu(pre)) <= TOL
    g = q(det_tensor(tuple(pre.shape), 47, -1, 1, bf16=False))
    pre.backward(g)
    dw, dg, db = ops.stem_conv_wgrad(xp, nhwc(g), ws, (H, W), scale, mean.cuda(), invstd)
    assert rel_l2(dw.cpu(), w.grad) <= TOL and rel_l2(dg.cpu(), gamma.grad) <= TOL and rel_l2(db.cpu(), beta.grad) <= TOL
    # max pool (ties, first-maximum rule) + adjoint + fused mask
    x = F.relu((q(det_tensor((2, 64, 15, 21), 51, -2, 2, bf16=False)) * 2).round() / 2).requires_grad_(True)
    yp = F.max_pool2d(x, 3, 2, 1)
    dy = q(det_tensor(tuple(yp.shape), 52, -1, 1, bf16=False))
    yp.backward(dy)
    yg, idx = ops.maxpool3x3s2_fwd(nhwc(x.detach()))
    assert yg.dtype == H16 and torch.equal(nchw(yg), yp.detach())
    dxm = ops.maxpool3x3s2_bwd(nhwc(dy), idx, (15, 21), nhwc(x.detach()))
    assert max_rel(nchw(dxm), x.grad * (x.detach() > 0).float()) <= 2 ** -10
    # subsample adjoint, add + mask, layout converters
    base = q(det_tensor((2, 64, 25, 42), 63, -1, 1, bf16=False))
    dy2 = q(det_tensor((2, 64, 13, 21), 62, -1, 1, bf16=False))
    ref = base.clone()
    ref[:, :, ::2, ::2] += dy2
    assert max_rel(nchw(ops.subsample2_bwd(nhwc(dy2), (25, 42), nhwc(base))), ref) <= 2 ** -10
    assert torch.equal(nchw(ops.subsample2_fwd(nhwc(base))), F.max_pool2d(base, 1, stride=2))
    a, b, m = (q(det_tensor((2, 64, 8, 8), s, -1, 1, bf16=False)) for s in (64, 65, 66))
    assert max_rel(nchw(ops.add_relu_mask(nhwc(a), nhwc(b), nhwc(m))), (a + b) * (m > 0).float()) <= 2 ** -10
    xc = q(det_tensor((2, 70, 9, 13), 71, -1, 1, bf16=False))
    yl = ops.to_nhwc_bf16(xc.cuda(), H16)
    assert yl.dtype == H16 and torch.equal(yl.float().cpu(), xc.permute(0, 2, 3, 1))
    assert torch.equal(ops.nhwc_to_nchw_f32(yl).cpu(), xc)


@pytest.mark.parametrize("depth,shape", [(50, (2, 3, 64, 128)), (101, (1, 3, 64, 64))])
def test_resnet_fpn_fwd_bwd_in_situ_f16(T, depth, shape):
    """C5's precision on the whole schedule: every fused launch in situ, whole backward teacher-forced."""
    import json
    import os
    import parity_util
    # the synthetic (un-normalised) weights of the parity nets let activations and gradients grow ~1.4x per residual
    # block: through R101's 33 blocks both leave fp16's range (65504).  That run therefore damps every residual
    # branch (last BN gamma x 1/4, exactly representable) and scales its cotangents by 2^-6 — a static loss scale,
    # SURVEY §8(d) C5, here used downwards.  R50 runs on the same weights as the bf16 test.
    deep = depth > 50
    res = parity_util.run_teacher_forced(T, depth, shape, dtype=H16, cot_scale=2.0 ** -6 if deep else 1.0,
                                         res_gain=0.25 if deep else 1.0)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_models_f16_r%d.json" % depth, "w") as f:
        json.dump(res, f, indent=1, default=float)
    parity_util.check(res, depth)
    # fp16 keeps 3 more mantissa bits than bf16: the same checks hold with tighter bounds
    assert max(res["forward_in_situ"].values()) <= 2e-4, res["forward_in_situ"]
    assert res["backward_teacher_forced"]["grad_median"] <= 1e-2, res["backward_teacher_forced"]


def test_half_input_selects_f16(T):
    """``model(x.half())`` computes and returns float16 (the reference's ``model.half()`` usage); a float32 image
    uses ``compute_dtype`` (default bfloat16).  Both operand sets are cached side by side."""
    m = T.ResNet(18).cuda().train()
    m.init_weights()
    x = det_tensor((1, 3, 64, 64), 5, -1, 1).cuda()
    o_bf = m(x)
    o_h = m(x.half())
    assert all(o.dtype == torch.bfloat16 for o in o_bf) and all(o.dtype == H16 for o in o_h)
    for a, b in zip(o_bf, o_h):
        assert rel_l2(a.float().cpu(), b.float().cpu()) <= 2e-2
    m.compute_dtype = H16
    assert all(o.dtype == H16 for o in m(x))
    neck = T.FPN([64, 128, 256, 512], 256, 5).cuda()
    neck.init_weights()
    outs = neck(o_h)
    assert all(o.dtype == H16 for o in outs)
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    assert all(p.grad is not None and p.grad.dtype == torch.float32 for p in list(m.parameters())[:3])
