"""CPU: the C box-op oracle (oracle/box_ref.c) against the known answers of SURVEY Appendix B and an independent
numpy restatement.  Parity of the box ops is UNPINNED by the reference (core/__init__.py is empty); these known
answers are the pin."""
import numpy as np

from oracle import box_ref as B


def test_base_anchors_known_answers():
    a = B.base_anchors(16, [8, 16, 32], [0.5, 1, 2]).astype(int).tolist()
    assert a == [[-83, -37, 98, 52], [-173, -83, 188, 98], [-354, -173, 369, 188], [-56, -56, 71, 71],
                 [-120, -120, 135, 135], [-248, -248, 263, 263], [-37, -83, 52, 98], [-83, -173, 98, 188],
                 [-173, -354, 188, 369]]
    b = B.base_anchors(4, [8], [0.5, 1, 2]).astype(int).tolist()
    assert b == [[-21, -9, 24, 12], [-14, -14, 17, 17], [-9, -21, 12, 24]]
    assert np.array_equal(B.np_base_anchors(16, [8, 16, 32], [0.5, 1, 2]), B.base_anchors(16, [8, 16, 32], [0.5, 1, 2]))


def test_anchor_grid_order_and_flags():
    base = B.base_anchors(4, [8], [0.5, 1, 2])
    a, v = B.anchor_grid(base, (2, 3), 4)
    assert a[3].tolist() == [-17, -9, 28, 12] and a[6].tolist() == [-13, -9, 32, 12]  # x fastest, anchor innermost
    a2, v2 = B.np_anchor_grid(base, (2, 3), 4, (1, 2))
    a1, v1 = B.anchor_grid(base, (2, 3), 4, (1, 2))
    assert np.array_equal(a1, a2) and np.array_equal(v1, v2)
    assert v1.reshape(2, 3, 3)[:, :, 0].tolist() == [[1, 1, 0], [0, 0, 0]]
    # the 5-level pyramid of the 800x1344 input has 268,569 anchors
    n = sum(B.anchor_grid(B.base_anchors(s, [8], [0.5, 1, 2]), hw, s)[0].shape[0] for hw, s in
            [((200, 336), 4), ((100, 168), 8), ((50, 84), 16), ((25, 42), 32), ((13, 21), 64)])
    assert n == 268569


def test_iou_known_answers():
    f = np.float32
    a = np.array([[0, 0, 9, 9]], dtype=f)
    assert B.iou_pairwise(a, a)[0, 0] == 1.0                                           # identical
    assert B.iou_pairwise(a, np.array([[20, 20, 29, 29]], dtype=f))[0, 0] == 0.0       # disjoint
    assert B.iou_pairwise(a, np.array([[10, 0, 19, 9]], dtype=f))[0, 0] == 0.0         # touching: x2_a + 1 == x1_b
    one_col = B.iou_pairwise(a, np.array([[9, 0, 18, 9]], dtype=f))[0, 0]              # one shared pixel column
    assert one_col == f(10.0) / f(190.0)
    rng = np.random.default_rng(0)
    x1, y1 = rng.uniform(0, 1000, 400).astype(f), rng.uniform(0, 700, 400).astype(f)
    b = np.stack([x1, y1, x1 + rng.uniform(8, 256, 400).astype(f), y1 + rng.uniform(8, 256, 400).astype(f)], -1)
    assert np.array_equal(B.iou_pairwise(b, b[:77]).view(np.uint32), B.np_iou_pairwise(b, b[:77]).view(np.uint32))
    assert B.iou_pairwise(b[:0], b).shape == (0, 400)


def test_nms_known_answers():
    f = np.float32
    bb = np.array([[0, 0, 9, 9], [0, 5, 9, 14], [100, 100, 110, 110]], dtype=f)   # IoU(0,1) = 50/150
    ss = np.array([0.9, 0.8, 0.7], dtype=f)
    thr = float(f(50.0) / f(150.0))
    keep, kept, cnt = B.nms(bb, ss, thr)
    assert keep.tolist() == [1, 1, 1] and cnt == 3            # exactly at the threshold: not suppressed (strict >)
    keep, kept, cnt = B.nms(bb, ss, float(np.nextafter(f(thr), f(0))))
    assert keep.tolist() == [1, 0, 1] and kept.tolist() == [0, 2, -1]
    # all-equal scores: index order
    keep, kept, cnt = B.nms(np.repeat(bb[:1], 4, 0), np.full(4, 0.5, dtype=f), 0.5)
    assert keep.tolist() == [1, 0, 0, 0] and kept[0] == 0
    # N = 0, N = 1
    assert B.nms(bb[:0], ss[:0], 0.5)[2] == 0
    assert B.nms(bb[:1], ss[:1], 0.5)[0].tolist() == [1]
    # random vs the O(N^2) Python loop
    rng = np.random.default_rng(1)
    x1, y1 = rng.uniform(0, 600, 300).astype(f), rng.uniform(0, 400, 300).astype(f)
    b = np.stack([x1, y1, x1 + rng.uniform(8, 256, 300).astype(f), y1 + rng.uniform(8, 256, 300).astype(f)], -1)
    s = (rng.integers(0, 50, 300) / 50.0).astype(f)   # many ties
    k1, kk1, c = B.nms(b, s, 0.5)
    k2, kk2 = B.py_nms(b, s, 0.5)
    assert np.array_equal(k1, k2) and np.array_equal(kk1[:c], kk2)


def test_bbox_norm_oracle_vs_reference_golden(golden_dir=None):
    """Box delta (de)normalisation: the numpy restatement reproduces vectors captured from the reference's own
    datasets/utils/bbox.py:118-166 bit for bit (this row IS pinned by the reference)."""
    import json
    import os
    from golden_util import det_tensor
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    man = json.load(open(os.path.join(gd, "manifest.json")))["bbox_norm"]
    gold = np.load(os.path.join(gd, "bbox_norm.npz"))
    b4 = det_tensor((257, 4), man["seed4"], man["lo"], man["hi"], bf16=False).numpy()
    b12 = det_tensor((65, 12), man["seed12"], man["lo"], man["hi"], bf16=False).numpy()
    for tag in ("a", "b"):
        m, s = man[tag]
        assert np.array_equal(B.np_bbox_normalize(b4, m, s), gold[tag + "/norm"])
        assert np.array_equal(B.np_bbox_denormalize(b4, m, s), gold[tag + "/denorm4"])
        assert np.array_equal(B.np_bbox_denormalize(b12, m, s), gold[tag + "/denorm12"])
