"""GPU parity tests of the individual HIP kernels (through the C ABI) against plain PyTorch fp32 on CPU.

Operands are bf16-representable, accumulation is fp32.  Tolerances (written here, used below):
  * forward / dgrad with fp32 output (pre-rounding value): max|err| / max|ref| <= 1e-3   (north-star figure)
  * the same kernels with bf16 output: <= 1 bf16 ulp of the bf16-rounded reference (2^-7 relative, loose bound)
  * wgrad (fp32 output, K up to N*H*W): rel-L2 <= 1e-3
"""
import os

import pytest
import torch
import torch.nn.functional as F

from golden_util import det_tensor, max_rel, rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-3


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from torch_detection_amd import ops as _ops
    from torch_detection_amd import _lib
    _lib.load()
    return _ops


def nhwc(t):  # NCHW float cpu -> NHWC bf16 cuda
    return t.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()


def nchw(t):  # NHWC cuda -> NCHW float cpu
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def pack_w(w):  # OIHW float (bf16-representable) -> [O][kh][kw][I] bf16 cuda
    return w.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()


def pack_wd(w, scale=None):  # -> [I][kh][kw][O] bf16 cuda with scale folded
    ws = w if scale is None else w * scale.view(-1, 1, 1, 1)
    return ws.permute(1, 2, 3, 0).contiguous().bfloat16().cuda()


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride
    (1, 20, 24, 64, 256, 1, 1),
    (2, 10, 12, 256, 64, 1, 1),
    (1, 20, 24, 64, 64, 3, 1),
    (2, 25, 42, 128, 128, 3, 2),
    (1, 12, 16, 256, 512, 1, 2),
    (1, 13, 21, 128, 128, 3, 1),
    (1, 25, 43, 64, 128, 1, 2),
]


# GEMM tile configurations forced through TDN_GEMM_CFG (ids of kCfgs in csrc/conv_igemm.hip): the production set —
# 0: 64x64, 1: 64x128 (pipelined fragments), 3: 192x256 (8 waves), 25: 64x64 with two split-K wave groups,
# 46: 128x128 (8 waves), 50: 64x64 with 128-deep K-steps (falls back to the library's choice where Cin % 128 != 0) —
# on top of whatever the library picks by itself (None)
GEMM_CFGS = {None: 64, 0: 64, 1: 128, 3: 256, 25: 64, 46: 128, 50: 64}


@pytest.mark.parametrize("tile", list(GEMM_CFGS))
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(ops, case, tile):
    N, H, W, Cin, Cout, k, s = case
    if tile is not None and Cout % GEMM_CFGS[tile] != 0:
        pytest.skip("tile does not divide Cout")
    os.environ.pop("TDN_GEMM_CFG", None)
    if tile is not None:
        os.environ["TDN_GEMM_CFG"] = str(tile)
    try:
        x = det_tensor((N, Cin, H, W), 1, -1, 1)
        w = det_tensor((Cout, Cin, k, k), 2, -0.2, 0.2)
        scale = det_tensor((Cout,), 3, 0.5, 1.5, bf16=False)
        shift = det_tensor((Cout,), 4, -0.5, 0.5, bf16=False)
        ref = F.conv2d(x, w, None, s, k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        res = det_tensor(tuple(ref.shape), 5, -1, 1)
        ref_r = F.relu(ref + res)
        xg, wg = nhwc(x), pack_w(w)
        y32 = ops.conv2d_fwd(xg, wg, k, s, k // 2, scale.cuda(), shift.cuda(), nhwc(res), ops.ADD_SAME, True,
                             out_f32=True)
        assert max_rel(nchw(y32), ref_r) <= TOL
        y16 = ops.conv2d_fwd(xg, wg, k, s, k // 2, scale.cuda(), shift.cuda(), nhwc(res), ops.ADD_SAME, True)
        err = (nchw(y16) - ref_r).abs()
        assert bool((err <= ref_r.abs() * 2 ** -7 + 1e-6).all())
        # no epilogue at all
        y0 = ops.conv2d_fwd(xg, wg, k, s, k // 2, out_f32=True)
        assert max_rel(nchw(y0), F.conv2d(x, w, None, s, k // 2)) <= TOL
    finally:
        os.environ.pop("TDN_GEMM_CFG", None)


@pytest.mark.parametrize("tile", [None, 3, 25, 46, 50])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad(ops, case, tile, monkeypatch):
    N, H, W, Cin, Cout, k, s = case
    if tile is not None:
        if Cin % GEMM_CFGS[tile] != 0:
            pytest.skip("tile does not divide the dgrad GEMM's N = Cin")
        monkeypatch.setenv("TDN_GEMM_CFG", str(tile))
    Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
    g = det_tensor((N, Cout, Ho, Wo), 11, -1, 1)
    w = det_tensor((Cout, Cin, k, k), 12, -0.2, 0.2)
    scale = det_tensor((Cout,), 13, 0.5, 1.5)
    wd = pack_wd(w, scale)  # bf16(scale*w)
    w_eff = wd.float().cpu().permute(3, 0, 1, 2).contiguous()  # OIHW of the folded, rounded weights
    xz = torch.zeros(N, Cin, H, W, requires_grad=True)
    F.conv2d(xz, w_eff, None, s, k // 2).backward(g)
    ref = xz.grad
    dx = ops.conv2d_dgrad(nhwc(g), wd, (H, W), k, s, k // 2, out_f32=True)
    assert max_rel(nchw(dx), ref) <= TOL
    # fused epilogue: + addend, ReLU mask
    add = det_tensor((N, Cin, H, W), 14, -1, 1)
    msk = det_tensor((N, Cin, H, W), 15, -1, 1)
    ref2 = (ref + add) * (msk > 0).float()
    dx2 = ops.conv2d_dgrad(nhwc(g), wd, (H, W), k, s, k // 2, nhwc(add), ops.ADD_SAME, nhwc(msk), out_f32=True)
    assert max_rel(nchw(dx2), ref2) <= TOL
    dx3 = ops.conv2d_dgrad(nhwc(g), wd, (H, W), k, s, k // 2, nhwc(add), ops.ADD_SAME, nhwc(msk))
    err = (nchw(dx3) - ref2).abs()
    assert bool((err <= ref2.abs() * 2 ** -7 + 1e-6).all())


def test_conv_epilogue_up2x_and_sumpool(ops):
    N, H, W, C = 2, 8, 12, 64
    x = det_tensor((N, 128, H, W), 21, -1, 1)
    w = det_tensor((C, 128, 1, 1), 22, -0.2, 0.2)
    bias = det_tensor((C,), 23, -0.5, 0.5, bf16=False)
    coarse = det_tensor((N, C, H // 2, W // 2), 24, -1, 1)
    ref = F.conv2d(x, w, bias) + F.interpolate(coarse, scale_factor=2, mode="nearest")
    y = ops.conv2d_fwd(nhwc(x), pack_w(w), 1, 1, 0, None, bias.cuda(), nhwc(coarse), ops.ADD_UP2X, False, out_f32=True)
    assert max_rel(nchw(y), ref) <= TOL
    # adjoint: dgrad of a 3x3 conv + 2x2 sum-pool of the finer gradient
    g = det_tensor((N, C, H, W), 25, -1, 1)
    w3 = det_tensor((C, C, 3, 3), 26, -0.2, 0.2)
    fine = det_tensor((N, C, 2 * H, 2 * W), 27, -1, 1)
    xz = torch.zeros(N, C, H, W, requires_grad=True)
    F.conv2d(xz, w3, None, 1, 1).backward(g)
    ref2 = xz.grad + F.avg_pool2d(fine, 2) * 4
    dx = ops.conv2d_dgrad(nhwc(g), pack_wd(w3), (H, W), 3, 1, 1, nhwc(fine), ops.ADD_SUMPOOL2, None, out_f32=True)
    assert max_rel(nchw(dx), ref2) <= TOL
    with pytest.raises(RuntimeError):
        ops.conv2d_fwd(nhwc(x), pack_w(w), 1, 1, 0, None, None, nhwc(det_tensor((N, C, 3, 6), 1)), ops.ADD_UP2X)


# the nine-tap wgrad kernel (3x3 / stride 1 / pad 1, Cout % 128 == 0): odd widths, several images, pixel counts that
# are not multiples of the 64-pixel K-step, single- and multi-split plans
T9_CASES = [(3, 13, 21, 128, 128, 3, 1), (2, 40, 48, 64, 256, 3, 1), (1, 7, 7, 192, 384, 3, 1),
            (2, 25, 42, 512, 512, 3, 1), (1, 3, 70, 64, 128, 3, 1), (5, 2, 8, 64, 128, 3, 1), (3, 5, 9, 64, 128, 3, 1),
            (1, 9, 64, 64, 128, 3, 1), (1, 6, 65, 64, 128, 3, 1), (1, 7, 63, 64, 128, 3, 1)]


@pytest.mark.parametrize("case", CONV_CASES + [(2, 40, 48, 64, 64, 3, 1), (2, 16, 16, 512, 512, 3, 1)] + T9_CASES)
@pytest.mark.parametrize("bn", [True, False])
def test_conv_wgrad(ops, case, bn, monkeypatch):
    if case in T9_CASES:
        monkeypatch.setenv("TDN_WGRAD9", "1")   # the plan would pick the tap-per-tile kernel at these small sizes
    N, H, W, Cin, Cout, k, s = case
    Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
    x = det_tensor((N, Cin, H, W), 31, -1, 1)
    g = det_tensor((N, Cout, Ho, Wo), 32, -1, 1)
    w = det_tensor((Cout, Cin, k, k), 33, -0.2, 0.2).requires_grad_(True)
    gamma = det_tensor((Cout,), 34, 0.5, 1.5, bf16=False).requires_grad_(True)
    beta = det_tensor((Cout,), 35, -0.5, 0.5, bf16=False).requires_grad_(True)
    mean = det_tensor((Cout,), 36, -0.2, 0.2, bf16=False)
    var = det_tensor((Cout,), 37, 0.5, 1.5, bf16=False)
    y = F.conv2d(x, w, None, s, k // 2)
    if bn:
        y = F.batch_norm(y, mean, var, gamma, beta, False, 0.1, 1e-5)
    else:
        y = y + beta.view(1, -1, 1, 1)
    y.backward(g)
    wf = pack_w(w.detach())
    if bn:
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        scale = (gamma.detach() * invstd)
        dw, dg, db = ops.conv2d_wgrad(nhwc(x), nhwc(g), wf, k, s, k // 2, scale.cuda(), mean.cuda(), invstd.cuda())
        assert rel_l2(dg.cpu(), gamma.grad) <= TOL
    else:
        dw, dg, db = ops.conv2d_wgrad(nhwc(x), nhwc(g), wf, k, s, k // 2)
    assert rel_l2(dw.cpu().permute(0, 3, 1, 2), w.grad) <= TOL
    assert rel_l2(db.cpu(), beta.grad) <= TOL
    # accumulate mode (beta=1): second call doubles
    dw2, _, db2 = ops.conv2d_wgrad(nhwc(x), nhwc(g), wf, k, s, k // 2,
                                   *( (scale.cuda(), mean.cuda(), invstd.cuda()) if bn else (None, None, None)),
                                   dw=dw.clone(), dgamma=dg.clone() if bn else None, dbeta=db.clone(), beta=1.0)
    assert rel_l2(dw2.cpu().permute(0, 3, 1, 2), 2 * w.grad) <= TOL
    assert rel_l2(db2.cpu(), 2 * beta.grad) <= TOL


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1, 30, 44), (3, 64, 96), (2, 128, 160), (1, 226, 318)])
def test_stem_pool_fused_equals_two_launches(ops, shape, dtype):
    """tdn_stem_pool_fwd (conv7x7/s2 + BN + ReLU + max pool in one launch) against tdn_stem_conv_fwd followed by
    tdn_maxpool3x3s2_fwd: pooled values and window indices bit for bit (odd conv grids, patches that overhang the
    image, ties at zero after the ReLU)."""
    N, H, W = shape
    img = (det_tensor((N, 3, H, W), 241, -2, 2) * 4).round() / 4          # coarse values: many exact ties
    w = det_tensor((64, 3, 7, 7), 242, -0.2, 0.2)
    scale = det_tensor((64,), 243, 0.5, 1.5, bf16=False).cuda()
    shift = det_tensor((64,), 244, -1.5, 0.5, bf16=False).cuda()         # mostly negative: ReLU zeros are common
    xp = ops.stage_image(img.cuda(), dtype)
    ws = ops.pack_stem_weight(w.cuda(), dtype)
    s = ops.stem_conv_fwd(xp, ws, (H, W), scale, shift, True)
    y0, i0 = ops.maxpool3x3s2_fwd(s)
    y1, i1 = ops.stem_pool_fwd(xp, ws, (H, W), scale, shift)
    assert torch.equal(y0, y1)
    assert torch.equal(i0, i1)
    assert float((y0 == 0).float().mean()) > 0.05     # the tie case is really exercised


@pytest.mark.parametrize("shape", [(1, 32, 48), (2, 64, 96)])
def test_stem(ops, shape):
    N, H, W = shape
    img = det_tensor((N, 3, H, W), 41, -2, 2)
    w = det_tensor((64, 3, 7, 7), 42, -0.2, 0.2).requires_grad_(True)
    gamma = det_tensor((64,), 43, 0.5, 1.5, bf16=False).requires_grad_(True)
    beta = det_tensor((64,), 44, -0.5, 0.5, bf16=False).requires_grad_(True)
    mean = det_tensor((64,), 45, -0.2, 0.2, bf16=False)
    var = det_tensor((64,), 46, 0.5, 1.5, bf16=False)
    pre = F.batch_norm(F.conv2d(img, w, None, 2, 3), mean, var, gamma, beta, False, 0.1, 1e-5)
    ref = F.relu(pre)
    xp = ops.stage_image(img.cuda())
    # staging: interior equals the image, halo is zero
    xpc = xp.float().cpu()
    assert torch.equal(xpc[:, 3:3 + H, 3:3 + W, :3], img.permute(0, 2, 3, 1))
    assert float(xpc[:, :3].abs().sum()) == 0 and float(xpc[..., 3].abs().sum()) == 0
    ws = ops.pack_stem_weight(w.detach().cuda())
    scale, shift, invstd = ops.bn_fold(gamma.detach().cuda(), beta.detach().cuda(), mean.cuda(), var.cuda(), 1e-5)
    y = ops.stem_conv_fwd(xp, ws, (H, W), scale, shift, True, out_f32=True)
    assert max_rel(nchw(y), ref) <= TOL
    # wgrad
    g = det_tensor(tuple(ref.shape), 47, -1, 1)
    pre.backward(g)
    dw, dg, db = ops.stem_conv_wgrad(xp, nhwc(g), ws, (H, W), scale, mean.cuda(), invstd)
    assert rel_l2(dw.cpu(), w.grad) <= TOL
    assert rel_l2(dg.cpu(), gamma.grad) <= TOL
    assert rel_l2(db.cpu(), beta.grad) <= TOL


@pytest.mark.parametrize("shape", [(2, 16, 24, 64), (1, 15, 21, 64), (1, 8, 8, 128)])
def test_maxpool(ops, shape):
    N, H, W, C = shape
    # quantise so that ties are common (post-ReLU activations have many equal zeros)
    x = (det_tensor((N, C, H, W), 51, -2, 2) * 2).round() / 2
    x = F.relu(x).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = det_tensor(tuple(y.shape), 52, -1, 1)
    y.backward(dy)
    yg, idx = ops.maxpool3x3s2_fwd(nhwc(x.detach()))
    assert torch.equal(nchw(yg), y.detach())
    dx = ops.maxpool3x3s2_bwd(nhwc(dy), idx, (H, W))
    assert max_rel(nchw(dx), x.grad) <= 2 ** -7
    # fused ReLU mask of the stem output
    dxm = ops.maxpool3x3s2_bwd(nhwc(dy), idx, (H, W), nhwc(x.detach()))
    assert max_rel(nchw(dxm), x.grad * (x.detach() > 0).float()) <= 2 ** -7
    # the same mask read from the pool's output (what the backbone's backward pass uses): identical bit for bit
    dxp = ops.maxpool3x3s2_bwd(nhwc(dy), idx, (H, W), pooled=yg)
    assert torch.equal(dxp, dxm)
    with pytest.raises(RuntimeError):
        ops.maxpool3x3s2_bwd(nhwc(dy), idx, (H, W), nhwc(x.detach()), pooled=yg)


def test_subsample_and_mask(ops):
    x = det_tensor((2, 64, 25, 42), 61, -1, 1)
    y = ops.subsample2_fwd(nhwc(x))
    assert torch.equal(nchw(y), F.max_pool2d(x, 1, stride=2))
    dy = det_tensor((2, 64, 13, 21), 62, -1, 1)
    base = det_tensor((2, 64, 25, 42), 63, -1, 1)
    ref = base.clone()
    ref[:, :, ::2, ::2] += dy
    dx = ops.subsample2_bwd(nhwc(dy), (25, 42), nhwc(base))
    assert max_rel(nchw(dx), ref) <= 2 ** -7
    dx0 = ops.subsample2_bwd(nhwc(dy), (25, 42))
    ref0 = torch.zeros_like(base)
    ref0[:, :, ::2, ::2] = dy
    assert torch.equal(nchw(dx0), ref0)
    a, b, m = (det_tensor((2, 64, 8, 8), s, -1, 1) for s in (64, 65, 66))
    out = ops.add_relu_mask(nhwc(a), nhwc(b), nhwc(m))
    assert max_rel(nchw(out), (a + b) * (m > 0).float()) <= 2 ** -7


SPLITK_CASES = [
    # N, H, W, Cin, Cout, k, stride   — few 128x128 tiles, long K: the launches that cut K over several workgroups
    (1, 25, 42, 512, 512, 3, 1),      # layer4 conv2 of one image: 36 tiles, 72 K-steps
    (2, 50, 84, 256, 256, 3, 1),      # layer3 conv2: 132 tiles, 36 K-steps
    (1, 50, 84, 1024, 256, 1, 1),     # layer3 conv1: 16 K-steps
    (1, 50, 84, 512, 512, 3, 2),      # stride 2: the dgrad runs as four output-parity classes with 1..4 taps each
    (1, 25, 42, 2048, 512, 1, 1),     # layer4 conv1 of one image
]


@pytest.mark.parametrize("mode", ["1", "2"], ids=["xcd_local", "agent_scope"])
@pytest.mark.parametrize("case", SPLITK_CASES)
def test_cross_workgroup_split_k(ops, case, mode, monkeypatch):
    """Cross-workgroup split-K (partials exchanged through the scratch buffer, summed in split order by the last
    workgroup to arrive): fp32 results equal the unsplit launch to summation-order accuracy and the fp32 reference
    at 1e-3, repeated launches are bit-identical (the ticket area returns to zero), forward and dgrad with the full
    epilogue (affine, residual, ReLU / mask)."""
    import ctypes
    from torch_detection_amd import _lib
    if mode == "1" and _lib.load().tdn_probe_xcd_mapping() != 1:
        pytest.skip("workgroup -> XCD mapping is not static on this device: the XCD-local exchange stays off")
    N, H, W, Cin, Cout, k, s = case
    x = det_tensor((N, Cin, H, W), 21, -1, 1)
    w = det_tensor((Cout, Cin, k, k), 22, -0.05, 0.05)
    scale = det_tensor((Cout,), 23, 0.5, 1.5, bf16=False)
    shift = det_tensor((Cout,), 24, -0.5, 0.5, bf16=False)
    Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
    res = det_tensor((N, Cout, Ho, Wo), 25, -1, 1)
    plan = (ctypes.c_int32 * 16)()
    xg, wg, rg = nhwc(x), pack_w(w), nhwc(res)
    sc, sh = scale.cuda(), shift.cuda()

    def fwd(out_f32):
        return ops.conv2d_fwd(xg, wg, k, s, k // 2, sc, sh, rg, ops.ADD_SAME, True, out_f32=out_f32)

    monkeypatch.setenv("TDN_SPLITK", mode)                       # off by default (slower on MI355X, see splitk_for)
    y1 = fwd(True)
    y2 = fwd(True)
    yb = fwd(False)
    assert torch.equal(y1, y2)                                   # deterministic, ticket area re-armed
    monkeypatch.setenv("TDN_SPLITK", "0")
    y0 = fwd(True)
    monkeypatch.setenv("TDN_SPLITK", mode)
    ref = torch.relu(F.conv2d(x, w, None, s, k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    assert not torch.equal(y1, y0)                               # the split launch really took the other path
    assert max_rel(nchw(y1), nchw(y0)) <= 1e-5
    assert max_rel(nchw(y1), ref) <= 1e-3
    assert max_rel(nchw(yb), ref) <= 2 ** -7
    # dgrad (K = taps * Cout), mask + addend epilogue
    g = det_tensor((N, Cout, Ho, Wo), 26, -1, 1)
    wd = pack_wd(w, scale)
    add = det_tensor((N, Cin, H, W), 27, -1, 1)
    msk = det_tensor((N, Cin, H, W), 28, -1, 1)
    gg, ag, mg = nhwc(g), nhwc(add), nhwc(msk)

    def dgrad():
        return ops.conv2d_dgrad(gg, wd, (H, W), k, s, k // 2, ag, ops.ADD_SAME, mg, out_f32=True)

    d1 = dgrad()
    d2 = dgrad()
    assert torch.equal(d1, d2)
    monkeypatch.setenv("TDN_SPLITK", "0")
    d0 = dgrad()
    assert max_rel(nchw(d1), nchw(d0)) <= 1e-5
    w_eff = wd.float().cpu().permute(3, 0, 1, 2).contiguous()
    oph = H - ((Ho - 1) * s - 2 * (k // 2) + k)
    opw = W - ((Wo - 1) * s - 2 * (k // 2) + k)
    dref = (F.conv_transpose2d(g, w_eff, None, s, k // 2, (oph, opw)) + add) * (msk > 0).float()
    assert max_rel(nchw(d1), dref) <= 1e-3
    # the ticket area is zero at rest
    ws = ops.splitk_workspace(xg.device)
    assert int(ws[:_lib.SPLITK_TICKET_BYTES].view(torch.int64).abs().sum().item()) == 0


def test_layout_converters(ops):
    x = det_tensor((2, 70, 9, 13), 71, -1, 1)
    xg = x.cuda()
    y = ops.to_nhwc_bf16(xg)
    assert torch.equal(y.float().cpu(), x.permute(0, 2, 3, 1))
    # strided (channels_last) source
    y2 = ops.to_nhwc_bf16(xg.contiguous(memory_format=torch.channels_last))
    assert torch.equal(y2, y)
    back = ops.nhwc_to_nchw_f32(y)
    assert torch.equal(back.cpu(), x)
    # zero-copy for permuted bf16 NHWC
    z = y.permute(0, 3, 1, 2)
    assert ops.to_nhwc_bf16(z).data_ptr() == y.data_ptr()
    # 16-bit NCHW-contiguous (and sliced) sources go through the library's own transpose (tdn_nchw16_to_nhwc)
    for dt in (torch.bfloat16, torch.float16):
        src = xg.to(dt)                                  # plain NCHW strides
        assert torch.equal(ops.to_nhwc_bf16(src, dt), src.permute(0, 2, 3, 1).contiguous())
        sl = src[:, 3:67, 1:8, 2:11]                     # non-contiguous view
        assert torch.equal(ops.to_nhwc_bf16(sl, dt), sl.permute(0, 2, 3, 1).contiguous())


def test_pack_and_fold(ops):
    w = det_tensor((128, 64, 3, 3), 81, -1, 1, bf16=False)
    scale = det_tensor((128,), 82, 0.5, 1.5, bf16=False)
    wf, wd = ops.pack_conv_weight(w.cuda(), scale.cuda())
    assert torch.equal(wf.float().cpu(), w.bfloat16().float().permute(0, 2, 3, 1))
    ref_d = (w.bfloat16().float() * scale.view(-1, 1, 1, 1)).bfloat16().float().permute(1, 2, 3, 0)
    assert torch.equal(wd.float().cpu(), ref_d)
    # float16 operands: torch's own two-step rounding (fp32 product, then .half()), not a fused single rounding
    wf16, wd16 = ops.pack_conv_weight(w.cuda(), scale.cuda(), True, torch.float16)
    assert torch.equal(wf16.float().cpu(), w.half().float().permute(0, 2, 3, 1))
    assert torch.equal(wd16.cpu(), (w.half().float() * scale.view(-1, 1, 1, 1)).half().permute(1, 2, 3, 0))
    # channels_last-strided parameter gives the same packing
    wcl = w.cuda().contiguous(memory_format=torch.channels_last)
    wf2, _ = ops.pack_conv_weight(wcl, scale.cuda())
    assert torch.equal(wf2, wf)
    g, b, m, v = (det_tensor((128,), s, lo, hi, bf16=False) for s, lo, hi in
                  ((83, 0.5, 1.5), (84, -1, 1), (85, -1, 1), (86, 0.5, 1.5)))
    sc, sh, inv = ops.bn_fold(g.cuda(), b.cuda(), m.cuda(), v.cuda(), 1e-5)
    inv_ref = 1.0 / torch.sqrt(v + 1e-5)
    assert torch.allclose(inv.cpu(), inv_ref, rtol=1e-6, atol=0)
    assert torch.allclose(sc.cpu(), g * inv_ref, rtol=1e-6, atol=0)
    assert torch.allclose(sh.cpu(), b - m * g * inv_ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_prepare_group_equals_per_layer_calls(ops, dtype):
    """tdn_prepare_group (BN fold + pack of many convs in one launch, > 30 members: two launches) is bit-identical to
    tdn_bn_fold + tdn_pack_conv_weight per layer — contiguous and channels_last 3x3 parameters, 1x1, with and without
    a BatchNorm behind the conv."""
    shapes = [(128, 64, 3, 3, True, False), (64, 128, 3, 3, True, True), (256, 64, 1, 1, True, False),
              (64, 256, 1, 1, False, False), (192, 320, 3, 3, False, True)] * 7      # 35 members
    entries, refs = [], []
    for i, (O, I, kh, kw, with_bn, cl) in enumerate(shapes):
        w = det_tensor((O, I, kh, kw), 300 + i, -1, 1, bf16=False).cuda()
        if cl:
            w = w.contiguous(memory_format=torch.channels_last)
        bn = fold = None
        scale = None
        if with_bn:
            g, b, m, v = (det_tensor((O,), 400 + 4 * i + j, lo, hi, bf16=False).cuda()
                          for j, (lo, hi) in enumerate(((0.5, 1.5), (-1, 1), (-1, 1), (0.5, 1.5))))
            bn = (g, b, m, v, 1e-5)
            fold = torch.empty(3, O, dtype=torch.float32, device="cuda")
            scale, shift, inv = ops.bn_fold(g, b, m, v, 1e-5)
            refs.append((scale, shift, inv))
        else:
            refs.append(None)
        wf, wd = ops.pack_conv_weight(w, scale, True, dtype)
        refs[-1] = (refs[-1], wf, wd)
        entries.append((w, bn, torch.empty_like(wf), torch.empty_like(wd), fold))
    ops.prepare_group(entries, dtype)
    torch.cuda.synchronize()
    bad = []
    for idx, ((w, bn, wf, wd, fold), (fr, rf, rd)) in enumerate(zip(entries, refs)):
        if bn is not None:
            for j, nm in enumerate(("scale", "shift", "invstd")):
                if not torch.equal(fold[j], fr[j]):
                    bad.append((idx, tuple(w.shape), nm, int((fold[j] != fr[j]).sum())))
        if not torch.equal(wf, rf):
            bad.append((idx, tuple(w.shape), "w_fwd", int((wf != rf).sum())))
        if not torch.equal(wd, rd):
            bad.append((idx, tuple(w.shape), "w_dgrad", int((wd != rd).sum())))
    assert not bad, bad[:8]


def test_bad_shapes_fail_loudly(ops):
    x = nhwc(det_tensor((1, 48, 8, 8), 1))
    w = pack_w(det_tensor((64, 48, 1, 1), 2))
    with pytest.raises(RuntimeError):
        ops.conv2d_fwd(x, w, 1, 1, 0)
    with pytest.raises(ValueError):
        ops.conv2d_fwd(x.float(), w, 1, 1, 0)


@pytest.mark.parametrize("case", [
    # N, H, W, C, groups, k, stride
    (2, 12, 16, 128, 32, 3, 1),    # 4 channels per group (ResNeXt-50 32x4d layer1)
    (1, 13, 21, 256, 32, 3, 2),    # 8 per group, stride 2
    (2, 9, 10, 512, 32, 3, 1),     # 16 per group
    (1, 8, 8, 1024, 32, 3, 2),     # 32 per group
    (1, 10, 12, 128, 2, 3, 1),     # 64 per group: one group per block
])
def test_grouped_conv(ops, case):
    """Block-diagonal grouped conv (ResNeXt conv2) forward / dgrad / wgrad + BN affine grads vs F.conv2d(groups=)."""
    N, H, W, C, G, k, s = case
    cpg = C // G
    x = det_tensor((N, C, H, W), 91, -1, 1).requires_grad_(True)
    w = det_tensor((C, cpg, k, k), 92, -0.3, 0.3).requires_grad_(True)
    gamma = det_tensor((C,), 93, 0.5, 1.5, bf16=False).requires_grad_(True)
    beta = det_tensor((C,), 94, -0.5, 0.5, bf16=False).requires_grad_(True)
    mean = det_tensor((C,), 95, -0.2, 0.2, bf16=False)
    var = det_tensor((C,), 96, 0.5, 1.5, bf16=False)
    pre = F.batch_norm(F.conv2d(x, w, None, s, k // 2, 1, G), mean, var, gamma, beta, False, 0.1, 1e-5)
    ref = F.relu(pre)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.detach() * invstd
    shift = beta.detach() - mean * scale
    wf, wd = ops.pack_gconv_weight(w.detach().cuda(), G, scale.cuda())
    y = ops.gconv2d_fwd(nhwc(x.detach()), wf, G, k, s, k // 2, scale.cuda(), shift.cuda(), relu=True, out_f32=True)
    assert max_rel(nchw(y), ref.detach()) <= TOL
    g = det_tensor(tuple(pre.shape), 97, -1, 1)
    pre.backward(g)
    # dgrad: the packed operand folds scale (rounded to bf16): compare against conv_transpose with those weights
    w_eff = (w.detach() * scale.view(-1, 1, 1, 1)).bfloat16().float()
    xz = torch.zeros(N, C, H, W, requires_grad=True)
    F.conv2d(xz, w_eff, None, s, k // 2, 1, G).backward(g)
    dx = ops.gconv2d_dgrad(nhwc(g), wd, G, (H, W), k, s, k // 2, out_f32=True)
    assert max_rel(nchw(dx), xz.grad) <= TOL
    dw, dg, db = ops.gconv2d_wgrad(nhwc(x.detach()), nhwc(g), wf, G, k, s, k // 2, scale.cuda(), mean.cuda(),
                                   invstd.cuda())
    assert tuple(dw.shape) == (C, k, k, cpg)
    assert rel_l2(dw.cpu().permute(0, 3, 1, 2), w.grad) <= TOL
    assert rel_l2(dg.cpu(), gamma.grad) <= TOL
    assert rel_l2(db.cpu(), beta.grad) <= TOL
    with pytest.raises(RuntimeError):
        ops.pack_gconv_weight(torch.zeros(96, 3, 3, 3, device="cuda"), 32)   # 96 channels: not a multiple of 64


FULL_SIZE = [
    # name, Cin, Cout, k, stride, H, W   — BASELINE shapes at the bench's 2 x 800 x 1344 batch
    ("fpn_convs.0 3x3 256->256", 256, 256, 3, 1, 200, 336),     # the dominant launch: 192x256 tiles, M = 134400
    ("layer1 conv3 1x1 64->256", 64, 256, 1, 1, 200, 336),
    ("layer2.0 conv2 3x3 s2 128->128", 128, 128, 3, 2, 200, 336),
    ("layer3 conv1 1x1 1024->256", 1024, 256, 1, 1, 50, 84),
]


@pytest.mark.parametrize("shape", FULL_SIZE, ids=[s[0] for s in FULL_SIZE])
def test_full_size_layers(ops, shape, monkeypatch):
    """BASELINE-size layers (the configurations only large M selects, e.g. the 192x256 tile at M = 134400), checked
    through size-independent properties: (1) every tile configuration accumulates K in the same order, so the library's
    own choice must equal the 64x64 tile BIT FOR BIT, forward and dgrad, with the fused epilogue; (2) a sample of
    output pixels against an fp32 CPU convolution of the corresponding input patches; (3) linearity in the input by
    a power of two (exact in bf16)."""
    name, Cin, Cout, k, s, H, W = shape
    N = 2
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, W, Cin, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) * 0.05).bfloat16().cuda()
    scale = (torch.rand(Cout, generator=g) + 0.5).cuda()
    shift = (torch.randn(Cout, generator=g) * 0.1).cuda()
    Ho, Wo = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
    res = torch.randn(N, Ho, Wo, Cout, generator=g).bfloat16().cuda()

    def fwd():
        return ops.conv2d_fwd(x, w, k, s, k // 2, scale, shift, res, ops.ADD_SAME, True)

    y_auto = fwd()
    monkeypatch.setenv("TDN_GEMM_CFG", "0")
    y_ref = fwd()
    monkeypatch.delenv("TDN_GEMM_CFG")
    assert torch.equal(y_auto, y_ref)
    # (2) sampled pixels vs CPU fp32
    xs, ws = x.float().cpu(), w.float().cpu().permute(0, 3, 1, 2).contiguous()
    rng = torch.Generator().manual_seed(5)
    pad = k // 2
    for _ in range(24):
        n = int(torch.randint(0, N, (1,), generator=rng))
        oh = int(torch.randint(0, Ho, (1,), generator=rng))
        ow = int(torch.randint(0, Wo, (1,), generator=rng))
        patch = torch.zeros(1, Cin, k, k)
        for kh in range(k):
            for kw in range(k):
                h, wq = oh * s - pad + kh, ow * s - pad + kw
                if 0 <= h < H and 0 <= wq < W:
                    patch[0, :, kh, kw] = xs[n, h, wq]
        ref = F.conv2d(patch, ws).view(-1) * scale.cpu() + shift.cpu() + res[n, oh, ow].float().cpu()
        ref = F.relu(ref)
        got = y_auto[n, oh, ow].float().cpu()
        assert bool(((got - ref).abs() <= ref.abs() * 2 ** -7 + 1e-5 * float(ref.abs().max() + 1)).all()), (n, oh, ow)
    # (3) linearity: conv(4x) == 4 conv(x) exactly (no epilogue)
    y1 = ops.conv2d_fwd(x, w, k, s, k // 2, out_f32=True)
    y4 = ops.conv2d_fwd((x.float() * 4).bfloat16(), w, k, s, k // 2, out_f32=True)
    assert torch.equal(y4, y1 * 4)
    # dgrad: auto == 64x64 tile, with the fused (+ addend, ReLU mask) epilogue
    gg = torch.randn(N, Ho, Wo, Cout, generator=g).bfloat16().cuda()
    wd = (torch.randn(Cin, k, k, Cout, generator=g) * 0.05).bfloat16().cuda()
    add = torch.randn(N, H, W, Cin, generator=g).bfloat16().cuda()
    dx_auto = ops.conv2d_dgrad(gg, wd, (H, W), k, s, k // 2, add, ops.ADD_SAME, x)
    monkeypatch.setenv("TDN_GEMM_CFG", "0")
    dx_ref = ops.conv2d_dgrad(gg, wd, (H, W), k, s, k // 2, add, ops.ADD_SAME, x)
    monkeypatch.delenv("TDN_GEMM_CFG")
    assert torch.equal(dx_auto, dx_ref)
    # wgrad at full size: the column sums (dbeta) are an exact-able checksum — sum of g over all pixels
    dw, _, db = ops.conv2d_wgrad(x, gg, w, k, s, k // 2)
    assert rel_l2(db.cpu(), gg.float().sum((0, 1, 2)).cpu()) <= 1e-4
    # the nine-tap kernel (the plan's choice for the big 3x3 layers) and the tap-per-tile kernel are two fp32
    # accumulations of the same bf16 products: they may differ only by summation order
    monkeypatch.setenv("TDN_WGRAD9", "0")
    dw_t1, _, db_t1 = ops.conv2d_wgrad(x, gg, w, k, s, k // 2)
    monkeypatch.delenv("TDN_WGRAD9")
    assert rel_l2(dw, dw_t1) <= 1e-5 and rel_l2(db, db_t1) <= 1e-5
    # and the weight gradient against a CPU einsum on a channel sample (full K = all pixels)
    co, ci = [0, Cout // 2, Cout - 1], [0, Cin - 1]
    gs = gg.float().cpu()[..., co]                                    # N,Ho,Wo,3
    for kh in range(k):
        for kw in range(k):
            xsub = torch.zeros(N, Ho, Wo, len(ci))
            for a_, oh in enumerate(range(Ho)):
                h = oh * s - pad + kh
                if not 0 <= h < H:
                    continue
                ows = [ow for ow in range(Wo) if 0 <= ow * s - pad + kw < W]
                wqs = [ow * s - pad + kw for ow in ows]
                xsub[:, oh, ows] = xs[:, h, wqs][..., ci]
            ref = torch.einsum("nhwo,nhwi->oi", gs, xsub)
            got = dw[co][:, kh, kw][:, ci].cpu()
            assert rel_l2(got, ref) <= 2e-3, (kh, kw)


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, dilation, stride
    (2, 14, 18, 64, 128, 2, 1),
    (1, 13, 21, 128, 64, 4, 1),
    (1, 16, 20, 64, 64, 2, 2),      # dilated and strided: 4 output-parity classes in the dgrad
    (1, 9, 9, 256, 256, 8, 1),      # dilation larger than half the map: most taps fall outside
])
def test_dilated_conv(ops, case):
    """3x3 convs with padding = dilation (conv3x3_group, layers.py:20-32; ResNet(dilations=...)): forward, dgrad,
    wgrad against F.conv2d(dilation=)."""
    N, H, W, Cin, Cout, d, s = case
    x = det_tensor((N, Cin, H, W), 101, -1, 1).requires_grad_(True)
    w = det_tensor((Cout, Cin, 3, 3), 102, -0.2, 0.2).requires_grad_(True)
    ref = F.conv2d(x, w, None, s, d, d)
    g = det_tensor(tuple(ref.shape), 103, -1, 1)
    ref.backward(g)
    y = ops.conv2d_fwd(nhwc(x.detach()), pack_w(w.detach()), 3, s, d, out_f32=True)
    assert tuple(y.shape) == (N, ref.shape[2], ref.shape[3], Cout)
    assert max_rel(nchw(y), ref.detach()) <= TOL
    dx = ops.conv2d_dgrad(nhwc(g), pack_wd(w.detach()), (H, W), 3, s, d, out_f32=True)
    assert max_rel(nchw(dx), x.grad) <= TOL
    dw, _, db = ops.conv2d_wgrad(nhwc(x.detach()), nhwc(g), pack_w(w.detach()), 3, s, d)
    assert rel_l2(dw.cpu().permute(0, 3, 1, 2), w.grad) <= TOL
    assert rel_l2(db.cpu(), g.sum((0, 2, 3))) <= TOL
