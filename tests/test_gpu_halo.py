"""GPU parity tests of the LDS-resident patch ("halo tile") conv kernel, csrc/conv_halo.hip, through the C ABI
(tdn_conv2d_fwd / tdn_conv2d_dgrad route 3x3 stride-1 layers to it; TDN_HALO_CFG3 / TDN_HALO_CFG1 force a
configuration on every shape it can take, TDN_HALO=3 lets it take 1x1 layers as well).

Reference: plain PyTorch fp32 on CPU (F.conv2d and its autograd), identical bf16-/fp16-representable operands.
Tolerances as in test_gpu_kernels.py: fp32 (pre-rounding) output max|err| / max|ref| <= 1e-3; 16-bit output within
1 ulp of the rounded reference.  On top of that the halo kernel must agree BIT FOR BIT with the generic kernel's
64 x 64 tile (same K order: channel chunk outer, taps inner, two 32-deep MFMA sub-steps) — a stricter check of the
patch addressing than any tolerance.
"""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from golden_util import det_tensor, max_rel

pytestmark = pytest.mark.gpu

TOL = 1e-3
HALO_ENV = ("TDN_HALO", "TDN_HALO_CFG3", "TDN_HALO_CFG1", "TDN_HALO_TH", "TDN_HALO_TW", "TDN_HALO_NT", "TDN_HALO_XBUF",
            "TDN_GEMM_CFG")
# configuration ids of kHalo3 / kHalo1 (csrc/conv_halo.hip) and the output-channel multiple each needs
CFG3 = {0: 128, 1: 128, 2: 64, 3: 64, 4: 128, 5: 128, 8: 128, 9: 128, 10: 64, 11: 128}
CFG1 = {0: 128, 1: 128, 2: 64, 3: 64, 4: 128, 6: 128, 7: 256, 8: 128, 9: 128, 10: 64}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from torch_detection_amd import ops as _ops
    from torch_detection_amd import _lib
    _lib.load()
    return _ops


@pytest.fixture(autouse=True)
def clean_env():
    saved = {k: os.environ.pop(k, None) for k in HALO_ENV}
    yield
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def nhwc(t, dtype=torch.bfloat16):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def pack_w(w, dtype=torch.bfloat16):
    return w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def pack_wd(w, dtype=torch.bfloat16):
    return w.permute(1, 2, 3, 0).contiguous().to(dtype).cuda()


def uses_halo(ops, kind, N, H, W, Cin, Cout, k, s, pad):
    o = (ctypes.c_int32 * 16)()
    assert ops._lib.load().tdn_conv2d_plan(kind, N, H, W, Cin, Cout, k, s, pad, o) == 0
    return o[8] >= 100


# N, H, W, Cin, Cout: odd sizes, patches that overhang the image, several images, one and many channel chunks
CASES3 = [(1, 20, 24, 64, 128), (2, 13, 21, 128, 128), (1, 25, 42, 256, 256), (3, 9, 10, 64, 64), (1, 50, 84, 128, 64),
          (2, 7, 7, 192, 384), (1, 33, 18, 320, 128)]


@pytest.mark.parametrize("cfg", sorted(CFG3))
@pytest.mark.parametrize("case", CASES3)
def test_halo_conv3x3_fwd_dgrad(ops, case, cfg):
    N, H, W, Cin, Cout = case
    x = det_tensor((N, Cin, H, W), 1, -1, 1)
    w = det_tensor((Cout, Cin, 3, 3), 2, -0.2, 0.2)
    scale = det_tensor((Cout,), 3, 0.5, 1.5, bf16=False)
    shift = det_tensor((Cout,), 4, -0.5, 0.5, bf16=False)
    res = det_tensor((N, Cout, H, W), 5, -1, 1)
    ref = F.relu(F.conv2d(x, w, None, 1, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    args = (nhwc(x), pack_w(w), 3, 1, 1, scale.cuda(), shift.cuda(), nhwc(res), ops.ADD_SAME, True)
    os.environ["TDN_GEMM_CFG"] = "0"
    y_gen = ops.conv2d_fwd(*args, out_f32=True)
    os.environ.pop("TDN_GEMM_CFG")
    os.environ["TDN_HALO_CFG3"] = str(cfg)
    ran = False
    if Cout % CFG3[cfg] == 0 and uses_halo(ops, 0, N, H, W, Cin, Cout, 3, 1, 1):   # (else: patch + ring exceed LDS)
        ran = True
        y = ops.conv2d_fwd(*args, out_f32=True)
        assert max_rel(nchw(y), ref) <= TOL
        assert torch.equal(y, y_gen), "halo and generic kernels share the K order: bit-identical"
        y16 = ops.conv2d_fwd(*args)
        err = (nchw(y16) - ref).abs()
        assert bool((err <= ref.abs() * 2 ** -7 + 1e-6).all())
    # input gradient (GEMM N = Cin): + addend, ReLU mask of the producer
    if Cin % CFG3[cfg] != 0 or not uses_halo(ops, 1, N, H, W, Cin, Cout, 3, 1, 1):
        if not ran:
            pytest.skip("configuration does not apply to this shape")
        return
    g = det_tensor((N, Cout, H, W), 11, -1, 1)
    wd = pack_wd(w)
    xz = torch.zeros(N, Cin, H, W, requires_grad=True)
    F.conv2d(xz, w, None, 1, 1).backward(g)
    add = det_tensor((N, Cin, H, W), 14, -1, 1)
    msk = det_tensor((N, Cin, H, W), 15, -1, 1)
    ref2 = (xz.grad + add) * (msk > 0).float()
    dx = ops.conv2d_dgrad(nhwc(g), wd, (H, W), 3, 1, 1, nhwc(add), ops.ADD_SAME, nhwc(msk), out_f32=True)
    assert max_rel(nchw(dx), ref2) <= TOL
    os.environ.pop("TDN_HALO_CFG3")
    os.environ["TDN_GEMM_CFG"] = "0"
    dx_gen = ops.conv2d_dgrad(nhwc(g), wd, (H, W), 3, 1, 1, nhwc(add), ops.ADD_SAME, nhwc(msk), out_f32=True)
    assert torch.equal(dx, dx_gen)


def test_halo_default_shapes_take_the_kernel(ops):
    """What the library routes to the halo kernel by itself (the BASELINE step's 3x3 stride-1 layers with >= 128
    output channels, per image and per batch) — and what it leaves to the generic kernel."""
    taken = [(1, 100, 168, 128, 128), (2, 50, 84, 256, 256), (1, 50, 84, 256, 256), (1, 25, 42, 512, 512)]
    for N, H, W, Cin, Cout in taken:
        assert uses_halo(ops, 0, N, H, W, Cin, Cout, 3, 1, 1), (N, H, W, Cin, Cout)
        assert uses_halo(ops, 1, N, H, W, Cin, Cout, 3, 1, 1), (N, H, W, Cin, Cout)
    assert not uses_halo(ops, 0, 2, 200, 336, 64, 64, 3, 1, 1)       # layer1: 64 output channels
    # the largest layer (fpn_convs.0): faster alone with the halo tile, slower in the step -> generic unless asked for
    assert not uses_halo(ops, 0, 2, 200, 336, 256, 256, 3, 1, 1)
    os.environ["TDN_HALO_BIG"] = "1"
    try:
        assert uses_halo(ops, 0, 2, 200, 336, 256, 256, 3, 1, 1) and uses_halo(ops, 1, 2, 200, 336, 256, 256, 3, 1, 1)
    finally:
        os.environ.pop("TDN_HALO_BIG")
    assert not uses_halo(ops, 0, 2, 100, 168, 128, 128, 3, 2, 1)     # stride 2
    assert not uses_halo(ops, 0, 2, 50, 84, 256, 1024, 1, 1, 0)      # 1x1 layers stay with the generic kernel
    assert not uses_halo(ops, 0, 2, 25, 42, 256, 256, 3, 1, 1)       # fpn_convs.3: too few pixels at 256 channels
    os.environ["TDN_HALO"] = "0"
    assert not uses_halo(ops, 0, 2, 50, 84, 256, 256, 3, 1, 1)


@pytest.mark.parametrize("patch", [(8, 16), (3, 42), (10, 12), (16, 8), (4, 32), (2, 50)])
def test_halo_patch_shapes(ops, patch):
    """Patch geometry: widths that are / are not multiples of 16 (fragments straddling patch rows), patches that do
    not divide the image."""
    th, tw = patch
    N, H, W, Cin, Cout = 2, 25, 42, 128, 128
    x = det_tensor((N, Cin, H, W), 41, -1, 1)
    w = det_tensor((Cout, Cin, 3, 3), 42, -0.2, 0.2)
    ref = F.conv2d(x, w, None, 1, 1)
    os.environ.update({"TDN_HALO_CFG3": "5", "TDN_HALO_TH": str(th), "TDN_HALO_TW": str(tw)})
    assert uses_halo(ops, 0, N, H, W, Cin, Cout, 3, 1, 1)
    y = ops.conv2d_fwd(nhwc(x), pack_w(w), 3, 1, 1, out_f32=True)
    assert max_rel(nchw(y), ref) <= TOL


def test_halo_dilated_and_epilogue_modes(ops):
    """Dilation 2 (conv3x3_group: padding = dilation), the FPN epilogues (nearest-2x add, 2x2 sum-pool add), ReLU6,
    float16."""
    N, H, W, C = 2, 12, 16, 128
    x = det_tensor((N, C, H, W), 51, -1, 1)
    w = det_tensor((C, C, 3, 3), 52, -0.2, 0.2)
    os.environ["TDN_HALO_CFG3"] = "5"
    assert uses_halo(ops, 0, N, H, W, C, C, 3, 1, 2)
    y = ops.conv2d_fwd(nhwc(x), pack_w(w), 3, 1, 2, out_f32=True)
    assert max_rel(nchw(y), F.conv2d(x, w, None, 1, 2, 2)) <= TOL
    coarse = det_tensor((N, C, H // 2, W // 2), 53, -1, 1)
    bias = det_tensor((C,), 54, -0.5, 0.5, bf16=False)
    ref = F.conv2d(x, w, bias, 1, 1) + F.interpolate(coarse, scale_factor=2, mode="nearest")
    y = ops.conv2d_fwd(nhwc(x), pack_w(w), 3, 1, 1, None, bias.cuda(), nhwc(coarse), ops.ADD_UP2X, True, out_f32=True)
    assert max_rel(nchw(y), ref.clamp(min=0)) <= TOL
    # ReLU6: values stored under 6 stay under 6 (relu6_top, common.h), so compare below the knee and at the clamp
    y6 = nchw(ops.conv2d_fwd(nhwc(x), pack_w(w), 3, 1, 1, None, bias.cuda(), nhwc(coarse), ops.ADD_UP2X, 2,
                             out_f32=True))
    r6 = ref.clamp(0, 6)
    low = r6 < 5.9
    assert float((y6 - r6)[low].abs().max()) <= TOL * 6 and bool((y6[r6 >= 6] == 6).all()) and float(y6.max()) <= 6
    g = det_tensor((N, C, H, W), 55, -1, 1)
    fine = det_tensor((N, C, 2 * H, 2 * W), 56, -1, 1)
    xz = torch.zeros(N, C, H, W, requires_grad=True)
    F.conv2d(xz, w, None, 1, 1).backward(g)
    ref2 = xz.grad + F.avg_pool2d(fine, 2) * 4
    dx = ops.conv2d_dgrad(nhwc(g), pack_wd(w), (H, W), 3, 1, 1, nhwc(fine), ops.ADD_SUMPOOL2, None, out_f32=True)
    assert max_rel(nchw(dx), ref2) <= TOL
    # float16 operands
    xh, wh = x.half().float(), w.half().float()
    yh = ops.conv2d_fwd(nhwc(xh, torch.float16), pack_w(wh, torch.float16), 3, 1, 1, out_f32=True)
    assert max_rel(nchw(yh), F.conv2d(xh, wh, None, 1, 1)) <= TOL


CASES1 = [(1, 20, 24, 64, 256, 1), (2, 10, 12, 256, 128, 1), (1, 12, 16, 256, 512, 2), (1, 25, 43, 64, 128, 2),
          (2, 25, 42, 512, 2048, 1), (3, 7, 9, 1024, 256, 1)]


@pytest.mark.parametrize("cfg", sorted(CFG1))
@pytest.mark.parametrize("case", CASES1)
def test_halo_conv1x1(ops, case, cfg):
    """1x1 layers (opt-in, TDN_HALO=3): linear pixel tiles, the chunk ring, stride 2 (downsample convs), several
    output-channel passes over the resident pixels."""
    N, H, W, Cin, Cout, s = case
    if Cout % CFG1[cfg] != 0:
        pytest.skip("configuration does not divide Cout")
    x = det_tensor((N, Cin, H, W), 61, -1, 1)
    w = det_tensor((Cout, Cin, 1, 1), 62, -0.2, 0.2)
    shift = det_tensor((Cout,), 63, -0.5, 0.5, bf16=False)
    ref = F.relu(F.conv2d(x, w, shift, s, 0))
    os.environ.update({"TDN_HALO": "3", "TDN_HALO_CFG1": str(cfg)})
    if not uses_halo(ops, 0, N, H, W, Cin, Cout, 1, s, 0):
        pytest.skip("plan does not fit LDS for this shape")
    for nt in ("1", "2"):
        os.environ["TDN_HALO_NT"] = nt
        y = ops.conv2d_fwd(nhwc(x), pack_w(w), 1, s, 0, None, shift.cuda(), None, ops.ADD_NONE, True, out_f32=True)
        assert max_rel(nchw(y), ref) <= TOL
