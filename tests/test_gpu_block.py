"""GPU parity tests of the one-launch Bottleneck kernel, csrc/conv_block.hip, through the C ABI
(tdn_bottleneck_fwd / tdn_bottleneck_dgrad; reference path: Bottleneck.forward, models/backbone/resnet.py:97-119).

Two checks per case:
  * against plain PyTorch fp32 on the CPU (F.conv2d chain with the intermediates rounded to the 16-bit type where the
    kernel stores them; autograd-free explicit backward), max|err| / max|ref| <= 1e-3 on every output;
  * BIT FOR BIT against the three tdn_conv2d_fwd / tdn_conv2d_dgrad launches it replaces (generic 64 x 64 tile forced:
    same K order and epilogue arithmetic) — a stricter check of the patch addressing, the tap mirroring of the
    backward pass and the zero padding of the recomputed halo than any tolerance.
Shapes cover whole tiles, ragged right / bottom tiles (H, W not multiples of 8 / 16), images smaller than one tile,
several images, and the BASELINE geometry of layer1 (one image of 200 x 336).
"""
import os

import pytest
import torch
import torch.nn.functional as F

from golden_util import det_tensor, max_rel

pytestmark = pytest.mark.gpu

TOL = 1e-3


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from torch_detection_amd import ops as _ops
    from torch_detection_amd import _lib
    _lib.load()
    return _ops


@pytest.fixture()
def generic_tiles():
    """Reference launches on the generic 64 x 64 tile (no halo kernel): the K order the block kernel reproduces."""
    saved = {k: os.environ.get(k) for k in ("TDN_GEMM_CFG", "TDN_HALO")}
    os.environ["TDN_GEMM_CFG"] = "0"
    os.environ["TDN_HALO"] = "0"
    yield
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def _case(N, H, W, C, dtype, seed):
    C4 = 4 * C
    x = det_tensor((N, H, W, C4), seed + 1).to(dtype)
    x = torch.relu(x)                                   # a block input is a ReLU output (mask source of the backward)
    w1 = (det_tensor((C, 1, 1, C4), seed + 2) * (2.0 / C4) ** 0.5).to(dtype)
    w2 = (det_tensor((C, 3, 3, C), seed + 3) * (2.0 / (9 * C)) ** 0.5).to(dtype)
    w3 = (det_tensor((C4, 1, 1, C), seed + 4) * (2.0 / C) ** 0.5).to(dtype)
    aff = []
    for i, n in enumerate((C, C, C, C, C4, C4)):
        t = det_tensor((n,), seed + 10 + i).float()
        aff.append(t * 0.25 + 1.0 if i % 2 == 0 else t * 0.1)     # scales around 1, shifts around 0
    return x, w1, w2, w3, aff


def _oihw(w):
    return w.float().permute(0, 3, 1, 2).contiguous()


def _fwd_ref(x, w1, w2, w3, aff, dtype):
    """fp32 CPU chain; intermediates rounded to the 16-bit type like the stored h1 / h2."""
    xc = x.float().permute(0, 3, 1, 2)

    def bn(z, s, b):
        return z * s.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    h1 = torch.relu(bn(F.conv2d(xc, _oihw(w1)), aff[0], aff[1])).to(dtype).float()
    h2 = torch.relu(bn(F.conv2d(h1, _oihw(w2), padding=1), aff[2], aff[3])).to(dtype).float()
    out = torch.relu(bn(F.conv2d(h2, _oihw(w3)), aff[4], aff[5]) + xc)
    return [t.permute(0, 2, 3, 1).contiguous() for t in (h1, h2, out)]


SHAPES = [(1, 8, 16), (1, 16, 32), (2, 24, 48), (1, 13, 21), (3, 5, 7), (1, 17, 40), (2, 9, 33)]


@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W", SHAPES)
def test_block_forward(ops, generic_tiles, N, H, W, dtype, C):
    if not ops.bottleneck_supported(H, W, C):
        pytest.skip("no one-launch kernel for C=%d in this build" % C)
    x, w1, w2, w3, aff = _case(N, H, W, C, dtype, 100 * H + W)
    dev = torch.device("cuda")
    xg, w1g, w2g, w3g = (t.contiguous().to(dev) for t in (x, w1, w2, w3))
    affg = [a.to(dev) for a in aff]
    h1, h2, out = ops.bottleneck_fwd(xg, w1g, w2g, w3g, affg)
    # (a) three separate launches, bit for bit
    r1 = ops.conv2d_fwd(xg, w1g, 1, 1, 0, affg[0], affg[1], relu=True)
    r2 = ops.conv2d_fwd(r1, w2g, 3, 1, 1, affg[2], affg[3], relu=True)
    r3 = ops.conv2d_fwd(r2, w3g, 1, 1, 0, affg[4], affg[5], xg, ops.ADD_SAME, True)
    torch.cuda.synchronize()
    for name, a, b in (("h1", h1, r1), ("h2", h2, r2), ("out", out, r3)):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), \
            "%s differs from the per-conv launches: %d of %d elements" % (
                name, int((a.view(torch.int16) != b.view(torch.int16)).sum()), a.numel())
    # (b) fp32 CPU reference
    ref = _fwd_ref(x, w1, w2, w3, aff, dtype)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for name, a, b in zip(("h1", "h2", "out"), (h1, h2, out), ref):
        # 16-bit outputs: within one rounding step of the fp32 reference (plus the fp32 accumulation tolerance)
        err = (a.float().cpu() - b).abs().max().item()
        assert err <= (TOL + ulp) * b.abs().max().item() + 1e-6, (name, err, b.abs().max().item())


@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W", SHAPES)
@pytest.mark.parametrize("with_mask3", [True, False])
def test_block_dgrad(ops, generic_tiles, N, H, W, dtype, with_mask3, C):
    if not ops.bottleneck_supported(H, W, C):
        pytest.skip("no one-launch kernel for C=%d in this build" % C)
    C4 = 4 * C
    seed = 100 * H + W + 7
    x, w1, w2, w3, aff = _case(N, H, W, C, dtype, seed)
    dev = torch.device("cuda")
    # saved activations of a forward pass (any 16-bit tensors serve as mask sources; use the real ones)
    xg, w1g, w2g, w3g = (t.contiguous().to(dev) for t in (x, w1, w2, w3))
    affg = [a.to(dev) for a in aff]
    h1, h2, out = ops.bottleneck_fwd(xg, w1g, w2g, w3g, affg)
    # dgrad packs [Cin][k][k][Cout] = scale[co] * w, rounded like tdn_pack_conv_weight does (two roundings)
    def dpack(w, scale):
        wf = w.float() * scale.view(-1, 1, 1, 1)          # w is [Cout][k][k][Cin] 16-bit values
        return wf.to(dtype).permute(3, 1, 2, 0).contiguous().to(dev)
    w1d, w2d, w3d = dpack(w1, aff[0]), dpack(w2, aff[2]), dpack(w3, aff[4])
    g = (det_tensor((N, H, W, C4), seed + 50) * 0.1).to(dtype).to(dev)
    g = torch.where(out > 0, g, torch.zeros_like(g)).contiguous()          # already masked by the block's own ReLU
    m3 = xg if with_mask3 else None
    g2, g1, dx = ops.bottleneck_dgrad(g, w3d, w2d, w1d, (h2, h1, m3))
    r2 = ops.conv2d_dgrad(g, w3d, (H, W), 1, 1, 0, mask_src=h2)
    r1 = ops.conv2d_dgrad(r2, w2d, (H, W), 3, 1, 1, mask_src=h1)
    rx = ops.conv2d_dgrad(r1, w1d, (H, W), 1, 1, 0, g, ops.ADD_SAME, m3)
    torch.cuda.synchronize()
    for name, a, b in (("g2", g2, r2), ("g1", g1, r1), ("dx", dx, rx)):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), \
            "%s differs from the per-conv launches: %d of %d elements" % (
                name, int((a.view(torch.int16) != b.view(torch.int16)).sum()), a.numel())
    # fp32 CPU reference of the chain (conv_transpose with the packed, scale-folded weights)
    def dconv(gin, wd, k):
        wt = wd.float().cpu().permute(3, 0, 1, 2).contiguous()      # [Cout][Cin][k][k] as conv_transpose2d wants
        return F.conv_transpose2d(gin.float().cpu().permute(0, 3, 1, 2), wt, padding=k // 2).permute(0, 2, 3, 1)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    c2 = torch.where(h2.cpu() > 0, dconv(g, w3d, 1), torch.zeros(()))
    assert max_rel(g2.float().cpu(), c2) <= TOL + ulp
    c1 = torch.where(h1.cpu() > 0, dconv(g2, w2d, 3), torch.zeros(()))
    assert max_rel(g1.float().cpu(), c1) <= TOL + ulp
    cx = dconv(g1, w1d, 1) + g.float().cpu()
    if with_mask3:
        cx = torch.where(x > 0, cx, torch.zeros(()))
    assert max_rel(dx.float().cpu(), cx) <= TOL + ulp


@pytest.mark.parametrize("rep", range(3))
@pytest.mark.parametrize("H,W,C", [(200, 336, 64), (100, 168, 128)])
def test_block_baseline_geometry(ops, generic_tiles, rep, H, W, C):
    """BASELINE geometry of layer1 / layer2 (one image, 200 x 336 x 64 / 100 x 168 x 128 mid channels): bit for bit
    against the per-conv path, forward and backward; repeated, because a hazard between workgroup phases only shows
    with every CU busy (scripts/block_race.py)."""
    N, dtype = 1, torch.bfloat16
    x, w1, w2, w3, aff = _case(N, H, W, C, dtype, 4242)
    dev = torch.device("cuda")
    xg, w1g, w2g, w3g = (t.contiguous().to(dev) for t in (x, w1, w2, w3))
    affg = [a.to(dev) for a in aff]
    h1, h2, out = ops.bottleneck_fwd(xg, w1g, w2g, w3g, affg)
    r1 = ops.conv2d_fwd(xg, w1g, 1, 1, 0, affg[0], affg[1], relu=True)
    r2 = ops.conv2d_fwd(r1, w2g, 3, 1, 1, affg[2], affg[3], relu=True)
    r3 = ops.conv2d_fwd(r2, w3g, 1, 1, 0, affg[4], affg[5], xg, ops.ADD_SAME, True)
    torch.cuda.synchronize()
    w1d, w2d, w3d = (w.permute(3, 1, 2, 0).contiguous() for w in (w1g, w2g, w3g))
    g = torch.where(out > 0, (det_tensor((N, H, W, 4 * C), 777) * 0.1).to(dtype).to(dev), torch.zeros((), device=dev, dtype=dtype))
    g = g.contiguous()
    bits = ops.bottleneck_bit_planes(N, H, W, C, dev)
    ops.bottleneck_fwd(xg, w1g, w2g, w3g, affg, bits=bits)
    g2, g1, dx = ops.bottleneck_dgrad(g, w3d, w2d, w1d, None, bits=bits)        # the production form: bit planes
    q2 = ops.conv2d_dgrad(g, w3d, (H, W), 1, 1, 0, mask_src=h2)
    q1 = ops.conv2d_dgrad(q2, w2d, (H, W), 3, 1, 1, mask_src=h1)
    qx = ops.conv2d_dgrad(q1, w1d, (H, W), 1, 1, 0, g, ops.ADD_SAME, xg)
    torch.cuda.synchronize()
    for name, a, b in (("h1", h1, r1), ("h2", h2, r2), ("out", out, r3), ("g2", g2, q2), ("g1", g1, q1), ("dx", dx, qx)):
        ne = (a.view(torch.int16) != b.view(torch.int16))
        if bool(ne.any()):
            idx = ne.nonzero()[:8].tolist()
            raise AssertionError("%s: %d of %d elements differ, first at (n, y, x, c) %s" % (name, int(ne.sum()), a.numel(), idx))


def _unpack_bits(b, ch):
    """[N][H][W][ch / 32] int32 words -> bool [N][H][W][ch] (bit c % 32 of word c / 32)."""
    w = b.cpu().to(torch.int64) & 0xFFFFFFFF
    sh = torch.arange(32, dtype=torch.int64)
    bits = ((w.unsqueeze(-1) >> sh) & 1).bool()
    return bits.reshape(*b.shape[:3], ch)


@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W", [(1, 16, 32), (2, 13, 21), (1, 17, 40)])
def test_block_relu_bit_planes(ops, generic_tiles, N, H, W, dtype, C):
    """The forward kernel's ReLU bit planes are exactly (h1 > 0), (h2 > 0), (x > 0) of the stored 16-bit tensors, and
    the backward kernel driven by them is bit-identical to the one driven by the 16-bit mask sources."""
    x, w1, w2, w3, aff = _case(N, H, W, C, dtype, 300 * H + W + C)
    dev = torch.device("cuda")
    xg, w1g, w2g, w3g = (t.contiguous().to(dev) for t in (x, w1, w2, w3))
    affg = [a.to(dev) for a in aff]
    bits = ops.bottleneck_bit_planes(N, H, W, C, dev)
    for b in bits:
        b.fill_(0x5A5A5A5A)            # poison: every word must be written
    h1, h2, out = ops.bottleneck_fwd(xg, w1g, w2g, w3g, affg, bits=bits)
    r1, r2, r3 = ops.bottleneck_fwd(xg, w1g, w2g, w3g, affg)
    torch.cuda.synchronize()
    for a, b in ((h1, r1), (h2, r2), (out, r3)):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))       # the planes do not perturb the outputs
    for name, b, ref, ch in (("h1", bits[0], h1, C), ("h2", bits[1], h2, C), ("x", bits[2], xg, 4 * C)):
        assert torch.equal(_unpack_bits(b, ch), (ref.float().cpu() > 0)), name
    w1d, w2d, w3d = (w.permute(3, 1, 2, 0).contiguous() for w in (w1g, w2g, w3g))
    g = torch.where(out > 0, (det_tensor((N, H, W, 4 * C), 991) * 0.1).to(dtype).to(dev), torch.zeros((), device=dev, dtype=dtype))
    g = g.contiguous()
    a2, a1, ax = ops.bottleneck_dgrad(g, w3d, w2d, w1d, (h2, h1, xg))
    b2, b1, bx = ops.bottleneck_dgrad(g, w3d, w2d, w1d, None, bits=bits)
    torch.cuda.synchronize()
    for name, a, b in (("g2", a2, b2), ("g1", a1, b1), ("dx", ax, bx)):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), name


# ---------------------------------------------------------------------------------------------------
# head block: the stage's first Bottleneck with a 1x1 downsample branch (layer1.0; resnet.py:130-136, :113-114)
# ---------------------------------------------------------------------------------------------------
def _head_case(N, H, W, dtype, seed):
    C, C4 = 64, 256
    x = torch.relu(det_tensor((N, H, W, C), seed + 1)).to(dtype)          # the max pool's output: >= 0
    w1 = (det_tensor((C, 1, 1, C), seed + 2) * (2.0 / C) ** 0.5).to(dtype)
    w2 = (det_tensor((C, 3, 3, C), seed + 3) * (2.0 / (9 * C)) ** 0.5).to(dtype)
    w3 = (det_tensor((C4, 1, 1, C), seed + 4) * (2.0 / C) ** 0.5).to(dtype)
    wd = (det_tensor((C4, 1, 1, C), seed + 5) * (2.0 / C) ** 0.5).to(dtype)
    aff = []
    for i, n in enumerate((C, C, C, C, C4, C4, C4, C4)):
        t = det_tensor((n,), seed + 10 + i).float()
        aff.append(t * 0.25 + 1.0 if i % 2 == 0 else t * 0.1)
    return x, w1, w2, w3, wd, aff


HEAD_SHAPES = [(1, 8, 16), (2, 24, 48), (1, 13, 21), (3, 5, 7), (1, 17, 40), (1, 200, 336)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W", HEAD_SHAPES)
@pytest.mark.parametrize("with_bits", [False, True])
def test_head_block_forward_and_dgrad(ops, generic_tiles, N, H, W, dtype, with_bits):
    """Bit for bit against the per-conv launches of layer1.0: downsample conv, conv1, conv2, conv3 + residual, and their
    input gradients (dx = conv1^T(g1) + downsample^T(g), unmasked)."""
    C, C4 = 64, 256
    if not ops.bottleneck_head_supported(H, W, C, C):
        pytest.skip("no head kernel in this build")
    x, w1, w2, w3, wd, aff = _head_case(N, H, W, dtype, 100 * H + W + 3)
    dev = torch.device("cuda")
    xg, w1g, w2g, w3g, wdg = (t.contiguous().to(dev) for t in (x, w1, w2, w3, wd))
    affg = [a.to(dev) for a in aff]
    bits = None
    if with_bits:
        bits = ops.bottleneck_bit_planes(N, H, W, C, dev)[:2]
        for b in bits:
            b.fill_(0x5A5A5A5A)
    res = ops.conv2d_fwd(xg, wdg, 1, 1, 0, affg[6], affg[7], relu=False)
    h1, h2, out = ops.bottleneck_head_fwd(xg, w1g, w2g, w3g, affg[:6], res, bits=bits)
    r1 = ops.conv2d_fwd(xg, w1g, 1, 1, 0, affg[0], affg[1], relu=True)
    r2 = ops.conv2d_fwd(r1, w2g, 3, 1, 1, affg[2], affg[3], relu=True)
    r3 = ops.conv2d_fwd(r2, w3g, 1, 1, 0, affg[4], affg[5], res, ops.ADD_SAME, True)
    torch.cuda.synchronize()
    for name, a, b in (("h1", h1, r1), ("h2", h2, r2), ("out", out, r3)):
        ne = a.view(torch.int16) != b.view(torch.int16)
        assert not bool(ne.any()), "%s: %d of %d elements differ, first %s" % (name, int(ne.sum()), a.numel(), ne.nonzero()[:4].tolist())
    # the same block with the downsample branch computed inside the launch
    k1, k2, kout = ops.bottleneck_head_fwd(xg, w1g, w2g, w3g, affg[:6], None, bits=bits, down=(wdg, affg[6], affg[7]))
    torch.cuda.synchronize()
    for name, a, b in (("h1", k1, r1), ("h2", k2, r2), ("out", kout, r3)):
        ne = a.view(torch.int16) != b.view(torch.int16)
        assert not bool(ne.any()), "in-launch downsample, %s: %d of %d elements differ, first %s" % (
            name, int(ne.sum()), a.numel(), ne.nonzero()[:4].tolist())
    if with_bits:
        for name, b, ref in (("h1", bits[0], h1), ("h2", bits[1], h2)):
            assert torch.equal(_unpack_bits(b, C), (ref.float().cpu() > 0)), name
    # fp32 reference of the forward pass
    xc = x.float().permute(0, 3, 1, 2)

    def bn(z, s, b):
        return z * s.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    f1 = torch.relu(bn(F.conv2d(xc, _oihw(w1)), aff[0], aff[1])).to(dtype).float()
    f2 = torch.relu(bn(F.conv2d(f1, _oihw(w2), padding=1), aff[2], aff[3])).to(dtype).float()
    fr = bn(F.conv2d(xc, _oihw(wd)), aff[6], aff[7]).to(dtype).float()
    fo = torch.relu(bn(F.conv2d(f2, _oihw(w3)), aff[4], aff[5]) + fr).permute(0, 2, 3, 1)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert max_rel(out.float().cpu(), fo) <= TOL + 2 * ulp
    # backward
    w1d, w2d, w3d, wdd = (w.permute(3, 1, 2, 0).contiguous() for w in (w1g, w2g, w3g, wdg))
    g = torch.where(out > 0, (det_tensor((N, H, W, C4), 555) * 0.1).to(dtype).to(dev), torch.zeros((), device=dev, dtype=dtype))
    g = g.contiguous()
    t = ops.conv2d_dgrad(g, wdd, (H, W), 1, 1, 0)
    if with_bits:
        g2, g1, dx = ops.bottleneck_head_dgrad(g, w3d, w2d, w1d, None, t, bits=bits)
    else:
        g2, g1, dx = ops.bottleneck_head_dgrad(g, w3d, w2d, w1d, (h2, h1), t)
    q2 = ops.conv2d_dgrad(g, w3d, (H, W), 1, 1, 0, mask_src=h2)
    q1 = ops.conv2d_dgrad(q2, w2d, (H, W), 3, 1, 1, mask_src=h1)
    qx = ops.conv2d_dgrad(q1, w1d, (H, W), 1, 1, 0, t, ops.ADD_SAME, None)
    torch.cuda.synchronize()
    for name, a, b in (("g2", g2, q2), ("g1", g1, q1), ("dx", dx, qx)):
        ne = a.view(torch.int16) != b.view(torch.int16)
        assert not bool(ne.any()), "%s: %d of %d elements differ, first %s" % (name, int(ne.sum()), a.numel(), ne.nonzero()[:4].tolist())
    assert dx.shape == (N, H, W, C)
    # the same with the downsample conv's input gradient computed inside the launch
    if with_bits:
        k2, k1, kx = ops.bottleneck_head_dgrad(g, w3d, w2d, w1d, None, None, bits=bits, down=wdd)
    else:
        k2, k1, kx = ops.bottleneck_head_dgrad(g, w3d, w2d, w1d, (h2, h1), None, down=wdd)
    torch.cuda.synchronize()
    for name, a, b in (("g2", k2, q2), ("g1", k1, q1), ("dx", kx, qx)):
        ne = a.view(torch.int16) != b.view(torch.int16)
        assert not bool(ne.any()), "in-launch downsample, %s: %d of %d elements differ, first %s" % (
            name, int(ne.sum()), a.numel(), ne.nonzero()[:4].tolist())


# ---------------------------------------------------------------------------------------------------
# C = 128 with 10 x 16 tiles (bottleneck128t_kernel; chosen by the library where it saves a round of workgroups)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W", SHAPES + [(1, 100, 168), (2, 100, 168), (1, 31, 50)])
@pytest.mark.parametrize("mode", ["masks", "nomask3", "bits"])
def test_block128_tall_tiles(ops, generic_tiles, monkeypatch, N, H, W, dtype, mode):
    """Forward and input-gradient chain with the 10-row tile forced, bit for bit against the per-conv launches; the bit
    planes it writes are exactly the signs of the stored tensors."""
    C, C4 = 128, 512
    monkeypatch.setenv("TDN_BLOCK128_TH", "10")
    x, w1, w2, w3, aff = _case(N, H, W, C, dtype, 77 * H + W)
    dev = torch.device("cuda")
    xg, w1g, w2g, w3g = (t.contiguous().to(dev) for t in (x, w1, w2, w3))
    affg = [a.to(dev) for a in aff]
    bits = None
    if mode == "bits":
        bits = ops.bottleneck_bit_planes(N, H, W, C, dev)
        for b in bits:
            b.fill_(0x5A5A5A5A)
    h1, h2, out = ops.bottleneck_fwd(xg, w1g, w2g, w3g, affg, bits=bits)
    r1 = ops.conv2d_fwd(xg, w1g, 1, 1, 0, affg[0], affg[1], relu=True)
    r2 = ops.conv2d_fwd(r1, w2g, 3, 1, 1, affg[2], affg[3], relu=True)
    r3 = ops.conv2d_fwd(r2, w3g, 1, 1, 0, affg[4], affg[5], xg, ops.ADD_SAME, True)
    torch.cuda.synchronize()
    for name, a, b in (("h1", h1, r1), ("h2", h2, r2), ("out", out, r3)):
        ne = a.view(torch.int16) != b.view(torch.int16)
        assert not bool(ne.any()), "%s: %d of %d elements differ, first %s" % (name, int(ne.sum()), a.numel(), ne.nonzero()[:4].tolist())
    if bits is not None:
        for name, b, ref, ch in (("h1", bits[0], h1, C), ("h2", bits[1], h2, C), ("x", bits[2], xg, C4)):
            assert torch.equal(_unpack_bits(b, ch), (ref.float().cpu() > 0)), name
    w1d, w2d, w3d = (w.permute(3, 1, 2, 0).contiguous() for w in (w1g, w2g, w3g))
    g = torch.where(out > 0, (det_tensor((N, H, W, C4), 313) * 0.1).to(dtype).to(dev), torch.zeros((), device=dev, dtype=dtype))
    g = g.contiguous()
    m3 = None if mode == "nomask3" else xg
    if bits is not None:
        g2, g1, dx = ops.bottleneck_dgrad(g, w3d, w2d, w1d, None, bits=bits)
    else:
        g2, g1, dx = ops.bottleneck_dgrad(g, w3d, w2d, w1d, (h2, h1, m3))
    q2 = ops.conv2d_dgrad(g, w3d, (H, W), 1, 1, 0, mask_src=h2)
    q1 = ops.conv2d_dgrad(q2, w2d, (H, W), 3, 1, 1, mask_src=h1)
    qx = ops.conv2d_dgrad(q1, w1d, (H, W), 1, 1, 0, g, ops.ADD_SAME, m3)
    torch.cuda.synchronize()
    for name, a, b in (("g2", g2, q2), ("g1", g1, q1), ("dx", dx, qx)):
        ne = a.view(torch.int16) != b.view(torch.int16)
        assert not bool(ne.any()), "%s: %d of %d elements differ, first %s" % (name, int(ne.sum()), a.numel(), ne.nonzero()[:4].tolist())
