"""GPU parity of the GroupNorm kernels (tdn_gn_fwd / tdn_gn_bwd) against torch.nn.functional.group_norm (fp32, CPU)
on identical 16-bit-representable inputs.  Tolerances: outputs are 16-bit, so <= 1 ulp of the output type relative to
the fp32 reference (2^-7 bf16 / 2^-10 fp16) + the reduction noise; dgamma / dbeta (fp32) rel-L2 <= 1e-3."""
import pytest
import torch
import torch.nn.functional as F

from golden_util import det_tensor, max_rel, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from torch_detection_amd import ops as _ops
    return _ops


CASES = [  # N, C, H, W
    (2, 64, 13, 21),     # 2 channels per group: a lane's 8 channels span 4 groups
    (1, 128, 25, 42),
    (2, 256, 16, 24),
    (2, 512, 7, 9),
    (1, 2048, 4, 5),     # one pixel per block pass
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_gn_fwd_bwd(ops, case, dt):
    N, C, H, W = case
    G = 32
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    rq = lambda t: t.to(dt).float()   # noqa: E731
    z = rq(det_tensor((N, C, H, W), 1, -2, 2, bf16=False) + 0.3).requires_grad_(True)
    gamma = det_tensor((C,), 2, 0.5, 1.5, bf16=False).requires_grad_(True)
    beta = det_tensor((C,), 3, -0.5, 0.5, bf16=False).requires_grad_(True)
    res = rq(det_tensor((N, C, H, W), 4, -1, 1, bf16=False))
    nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dt).cuda()   # noqa: E731
    nc = lambda t: t.float().cpu().permute(0, 3, 1, 2).contiguous()            # noqa: E731
    pre = F.group_norm(z, G, gamma, beta, 1e-5)
    ref = F.relu(pre + res)
    y, stats = ops.gn_fwd(nh(z), gamma.detach().cuda(), beta.detach().cuda(), G, 1e-5, nh(res), True)
    assert y.dtype == dt and tuple(stats.shape) == (N, C, 2)
    err = (nc(y) - ref.detach()).abs()
    assert bool((err <= ref.detach().abs() * ulp + 1e-5 * float(ref.abs().max())).all())
    # statistics themselves
    zg = z.detach().view(N, G, -1)
    assert torch.allclose(stats[:, ::C // G, 0].cpu(), zg.mean(-1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(stats[:, ::C // G, 1].cpu(), 1.0 / torch.sqrt(zg.var(-1, unbiased=False) + 1e-5), rtol=1e-5)
    # no addend, no relu
    y0, _ = ops.gn_fwd(nh(z), gamma.detach().cuda(), beta.detach().cuda(), G)
    assert bool(((nc(y0) - pre.detach()).abs() <= pre.detach().abs() * ulp + 1e-5 * float(pre.abs().max())).all())
    # FPN top-down form: + nearest-2x upsampled coarser level
    if H % 2 == 0 and W % 2 == 0:
        coarse = rq(det_tensor((N, C, H // 2, W // 2), 6, -1, 1, bf16=False))
        ref_u = pre.detach() + F.interpolate(coarse, scale_factor=2, mode="nearest")
        yu, _ = ops.gn_fwd(nh(z), gamma.detach().cuda(), beta.detach().cuda(), G, 1e-5, nh(coarse), False, ops.ADD_UP2X)
        assert bool(((nc(yu) - ref_u).abs() <= ref_u.abs() * ulp + 1e-5 * float(ref_u.abs().max())).all())
    # backward: g = cotangent masked by the ReLU (what the consumer's dgrad epilogue hands over)
    cot = rq(det_tensor((N, C, H, W), 5, -1, 1, bf16=False))
    g = rq(cot * (ref.detach() > 0).float())
    pre.backward(g)
    dz, dg, db = ops.gn_bwd(nh(g), nh(z), stats, gamma.detach().cuda(), G)
    assert dz.dtype == dt
    assert rel_l2(dg.cpu(), gamma.grad) <= 1e-3 and rel_l2(db.cpu(), beta.grad) <= 1e-3
    assert max_rel(nc(dz), z.grad) <= 2 * ulp
    # accumulate into existing affine grads
    _, dg2, db2 = ops.gn_bwd(nh(g), nh(z), stats, gamma.detach().cuda(), G, dg.clone(), db.clone(), True)
    assert rel_l2(dg2.cpu(), 2 * gamma.grad) <= 1e-3 and rel_l2(db2.cpu(), 2 * beta.grad) <= 1e-3


def test_gn_bad_shapes(ops):
    z = torch.zeros(1, 4, 4, 96, dtype=torch.bfloat16, device="cuda")
    w = torch.ones(96, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gn_fwd(z, w, w, 32)       # 96 channels: not a power of two
    z = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device="cuda")
    w = torch.ones(64, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gn_fwd(z, w, w, 48)       # groups do not divide channels


@pytest.mark.parametrize("case", [(2, 64, 13, 21), (3, 256, 8, 12), (1, 1024, 5, 4)])
def test_bn_train_fwd_bwd(ops, case):
    """Training-mode BatchNorm2d (batch statistics, running-stat update) against F.batch_norm(training=True)."""
    N, C, H, W = case
    dt, ulp = torch.bfloat16, 2.0 ** -7
    rq = lambda t: t.to(dt).float()   # noqa: E731
    z = rq(det_tensor((N, C, H, W), 1, -2, 2, bf16=False) + 0.3).requires_grad_(True)
    gamma = det_tensor((C,), 2, 0.5, 1.5, bf16=False).requires_grad_(True)
    beta = det_tensor((C,), 3, -0.5, 0.5, bf16=False).requires_grad_(True)
    rm0 = det_tensor((C,), 4, -0.2, 0.2, bf16=False)
    rv0 = det_tensor((C,), 5, 0.5, 1.5, bf16=False)
    res = rq(det_tensor((N, C, H, W), 6, -1, 1, bf16=False))
    nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dt).cuda()   # noqa: E731
    nc = lambda t: t.float().cpu().permute(0, 3, 1, 2).contiguous()            # noqa: E731
    rm, rv = rm0.clone(), rv0.clone()
    pre = F.batch_norm(z, rm, rv, gamma, beta, True, 0.1, 1e-5)
    ref = F.relu(pre + res)
    rmg, rvg = rm0.clone().cuda(), rv0.clone().cuda()
    y, stats = ops.bn_train_fwd(nh(z), gamma.detach().cuda(), beta.detach().cuda(), rmg, rvg, 0.1, 1e-5, nh(res), True)
    err = (nc(y) - ref.detach()).abs()
    assert bool((err <= ref.detach().abs() * ulp + 1e-5 * float(ref.detach().abs().max())).all())
    assert torch.allclose(rmg.cpu(), rm, rtol=1e-5, atol=1e-6) and torch.allclose(rvg.cpu(), rv, rtol=1e-5, atol=1e-6)
    cot = rq(det_tensor((N, C, H, W), 7, -1, 1, bf16=False))
    g = rq(cot * (ref.detach() > 0).float())
    pre.backward(g)
    dz, dg, db = ops.bn_train_bwd(nh(g), nh(z), stats, gamma.detach().cuda())
    assert rel_l2(dg.cpu(), gamma.grad) <= 1e-3 and rel_l2(db.cpu(), beta.grad) <= 1e-3
    assert max_rel(nc(dz), z.grad) <= 2 * ulp
