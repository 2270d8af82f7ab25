"""GPU parity tests of the box ops (anchor grid / pairwise IoU / NMS): bit-exact against the C oracle.

Parity is UNPINNED by the reference (core/ is empty): the oracle is oracle/box_ref.c,
itself checked against the Appendix-B known answers in tests/test_oracle_box.py.
"""
import numpy as np
import pytest
import torch

from oracle import box_ref as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from torch_detection_amd import ops as _ops
    return _ops


def rand_boxes(n, seed, integer):
    g = torch.Generator().manual_seed(seed)
    w = torch.rand(n, generator=g) * (256 - 8) + 8
    h = torch.rand(n, generator=g) * (256 - 8) + 8
    x1 = torch.rand(n, generator=g) * (1344 - w)
    y1 = torch.rand(n, generator=g) * (800 - h)
    b = torch.stack([x1, y1, x1 + w, y1 + h], -1).float()
    if integer:
        b = b.floor()
    s = torch.rand(n, generator=g).float()
    return b.contiguous(), s.contiguous()


LEVELS = [((200, 336), 4), ((100, 168), 8), ((50, 84), 16), ((25, 42), 32), ((13, 21), 64)]


def test_anchor_grid_pyramid(ops):
    total = 0
    for (fh, fw), stride in LEVELS:
        base = B.base_anchors(stride, [8], [0.5, 1.0, 2.0])
        ref, vref = B.anchor_grid(base, (fh, fw), stride, (fh - 1, fw - 2))
        a, v = ops.anchor_grid(torch.from_numpy(base).cuda(), (fh, fw), stride, (fh - 1, fw - 2))
        assert np.array_equal(a.cpu().numpy().view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(a.cpu().numpy().astype(np.int32), ref.astype(np.int32))
        assert np.array_equal(v.cpu().numpy(), vref)
        total += a.shape[0]
    assert total == 268569


def test_anchor_pyramid_one_launch(ops):
    """tdn_anchor_pyramid: all five levels from one launch == the per-level kernel == the C oracle, bit for bit,
    valid flags included; an empty level in the middle is fine."""
    import torch_detection_amd as T
    gens = [T.AnchorGenerator(st, [8], [0.5, 1.0, 2.0]) for _, st in LEVELS]
    sizes = [fs for fs, _ in LEVELS]
    strides = [st for _, st in LEVELS]
    vsizes = [(fh - 1, fw - 2) for fh, fw in sizes]
    anchors, flags = T.anchor_pyramid(gens, sizes, strides, "cuda", vsizes)
    assert sum(a.shape[0] for a in anchors) == 268569
    for g, fs, st, vs, a, v in zip(gens, sizes, strides, vsizes, anchors, flags):
        ref, vref = B.anchor_grid(B.base_anchors(st, [8], [0.5, 1.0, 2.0]), fs, st, vs)
        assert np.array_equal(a.cpu().numpy().view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(v.cpu().numpy(), vref)
        assert torch.equal(a, g.grid_anchors(fs, st, "cuda"))
    a2, v2 = T.anchor_pyramid(gens[:3], [sizes[0], (0, 7), sizes[2]], strides[:3], "cuda")
    assert a2[1].shape == (0, 4) and torch.equal(a2[0], anchors[0]) and torch.equal(a2[2], anchors[2])
    assert bool(v2[0].all()) and bool(v2[2].all())


def test_anchor_grid_edge(ops):
    base = B.base_anchors(16, [8, 16, 32], [0.5, 1.0, 2.0])
    a, v = ops.anchor_grid(torch.from_numpy(base).cuda(), (0, 5), 16)
    assert a.shape == (0, 4) and v.shape == (0,)
    a, v = ops.anchor_grid(torch.from_numpy(base).cuda(), (1, 1), 16)
    assert np.array_equal(a.cpu().numpy(), base)


@pytest.mark.parametrize("integer", [True, False])
@pytest.mark.parametrize("n,m", [(1000, 1000), (10000, 100), (37, 53), (1, 1), (0, 5), (5, 0)])
def test_iou_bit_exact(ops, n, m, integer):
    a, _ = rand_boxes(n, 0, integer)
    b, _ = rand_boxes(m, 1, integer)
    ref = B.iou_pairwise(a.numpy(), b.numpy())
    out = ops.bbox_iou_pairwise(a.cuda(), b.cuda())
    assert out.shape == (n, m)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("integer", [True, False])
def test_iou_full_size(ops, integer):
    """N = M = 10k (BASELINE config 3): ALL 10^8 outputs against oracle/box_ref.c as uint32 bit patterns, in row
    chunks (the C oracle does the whole matrix in ~0.7 s; 400 MB output), plus symmetry, unit diagonal and range."""
    a, _ = rand_boxes(10000, 0, integer)
    out = ops.bbox_iou_pairwise(a.cuda(), a.cuda())
    assert bool((out.diagonal() == 1).all())
    assert torch.equal(out, out.t())
    assert float(out.min()) >= 0 and float(out.max()) <= 1
    an = a.numpy()
    for r0 in range(0, 10000, 1000):
        ref = B.iou_pairwise(an[r0:r0 + 1000], an)
        got = out[r0:r0 + 1000].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "rows %d..%d" % (r0, r0 + 999)


@pytest.mark.parametrize("integer", [True, False])
@pytest.mark.parametrize("n", [10000, 1000, 65, 64, 63, 2, 1])
def test_nms_bit_exact(ops, n, integer):
    b, s = rand_boxes(n, 0, integer)
    keep_ref, kept_ref, cnt = B.nms(b.numpy(), s.numpy(), 0.5)
    keep, kept, num = ops.nms(b.cuda(), s.cuda(), 0.5)
    assert int(num.item()) == cnt
    assert np.array_equal(keep.cpu().numpy(), keep_ref)
    assert np.array_equal(kept.cpu().numpy(), kept_ref)


@pytest.mark.parametrize("n", [1023, 1024, 1025, 2048 + 17, 5000])
def test_nms_block_scan_equals_single_wave_scan(ops, n, monkeypatch):
    """The 1024-thread super-chunk scan and the single-wave scan (TDN_NMS_ONEWAVE=1) give identical keep masks, kept
    indices and counts — sizes on both sides of the 1024-box super-chunk boundary, a low threshold (few boxes kept:
    long removal chains) and a high one (almost everything kept: longest serial resolve)."""
    b, s = rand_boxes(n, 11, False)
    for thr in (0.05, 0.5, 0.95):
        monkeypatch.setenv("TDN_NMS_ONEWAVE", "1")
        k1, i1, n1 = ops.nms(b.cuda(), s.cuda(), thr)
        monkeypatch.setenv("TDN_NMS_ONEWAVE", "0")
        k2, i2, n2 = ops.nms(b.cuda(), s.cuda(), thr)
        assert int(n1.item()) == int(n2.item())
        assert torch.equal(k1, k2) and torch.equal(i1, i2)
        keep_ref, kept_ref, cnt = B.nms(b.numpy(), s.numpy(), thr)
        assert cnt == int(n2.item()) and np.array_equal(k2.cpu().numpy(), keep_ref)
        assert np.array_equal(i2.cpu().numpy(), kept_ref)


def test_nms_edge_cases(ops):
    # N = 0
    keep, kept, num = ops.nms(torch.zeros(0, 4).cuda(), torch.zeros(0).cuda(), 0.5)
    assert keep.numel() == 0 and int(num.item()) == 0
    # all-equal scores: index order decides
    b, _ = rand_boxes(500, 3, True)
    s = torch.full((500,), 0.5)
    keep_ref, kept_ref, cnt = B.nms(b.numpy(), s.numpy(), 0.3)
    keep, kept, num = ops.nms(b.cuda(), s.cuda(), 0.3)
    assert np.array_equal(keep.cpu().numpy(), keep_ref) and np.array_equal(kept.cpu().numpy(), kept_ref)
    # IoU exactly at the threshold is NOT suppressed (strict >): boxes [0,0,9,9] and [0,5,9,14]: inter 50, union 150
    bb = torch.tensor([[0, 0, 9, 9], [0, 5, 9, 14], [100, 100, 110, 110]], dtype=torch.float32)
    ss = torch.tensor([0.9, 0.8, 0.7])
    thr = float(np.float32(50.0) / np.float32(150.0))
    keep, kept, num = ops.nms(bb.cuda(), ss.cuda(), thr)
    assert keep.cpu().tolist() == [1, 1, 1] and int(num.item()) == 3
    keep, kept, num = ops.nms(bb.cuda(), ss.cuda(), np.nextafter(np.float32(thr), np.float32(0)).item())
    assert keep.cpu().tolist() == [1, 0, 1] and kept.cpu().tolist() == [0, 2, -1]
    # idempotence: NMS of the kept set keeps everything
    b, s = rand_boxes(3000, 5, False)
    keep, kept, num = ops.nms(b.cuda(), s.cuda(), 0.5)
    k = int(num.item())
    sel = kept[:k]
    keep2, _, num2 = ops.nms(b.cuda()[sel].contiguous(), s.cuda()[sel].contiguous(), 0.5)
    assert int(num2.item()) == k and bool(keep2.all())


def test_bbox_normalize_denormalize_vs_reference_golden():
    """datasets/utils/bbox.py:118-166 — bit-exact against vectors captured from the reference's own functions,
    including the in-place contract of bbox_normalize and the 4C (class-specific) layout of bbox_denormalize."""
    import json
    import os
    import torch_detection_amd as T
    from golden_util import det_tensor
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    man = json.load(open(os.path.join(gd, "manifest.json")))["bbox_norm"]
    gold = np.load(os.path.join(gd, "bbox_norm.npz"))
    b4 = det_tensor((257, 4), man["seed4"], man["lo"], man["hi"], bf16=False)
    b12 = det_tensor((65, 12), man["seed12"], man["lo"], man["hi"], bf16=False)
    for tag in ("a", "b"):
        m, s = man[tag]
        t = b4.clone().cuda()
        r = T.bbox_normalize(t, m, s)
        assert r.data_ptr() == t.data_ptr()                      # in place, like the reference
        assert np.array_equal(t.cpu().numpy().view(np.uint32), gold[tag + "/norm"].view(np.uint32))
        d4 = T.bbox_denormalize(b4.clone().cuda(), m, s)
        d12 = T.bbox_denormalize(b12.clone().cuda(), m, s)
        assert np.array_equal(d4.cpu().numpy().view(np.uint32), gold[tag + "/denorm4"].view(np.uint32))
        assert np.array_equal(d12.cpu().numpy().view(np.uint32), gold[tag + "/denorm12"].view(np.uint32))
    # round trip at full anchor count (268,569 x 4): denormalize(normalize(x)) ~ x
    x = torch.rand(268569, 4).cuda()
    y = T.bbox_denormalize(T.bbox_normalize(x.clone(), [0, 0, 0, 0], [0.1, 0.1, 0.2, 0.2]), [0, 0, 0, 0],
                           [0.1, 0.1, 0.2, 0.2])
    assert torch.allclose(x, y, rtol=1e-6, atol=1e-7)
    assert T.bbox_denormalize(torch.zeros(0, 4).cuda()).shape == (0, 4)
    with pytest.raises(AssertionError):
        T.bbox_denormalize(torch.zeros(3, 6).cuda())
