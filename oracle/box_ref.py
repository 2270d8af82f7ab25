"""oracle/box_ref.py — loader for the C box-op oracle + an independent numpy restatement.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  PARITY UNPINNED by
the reference: /root/reference/core/__init__.py is empty; semantics are SURVEY.md Appendix B (see
box_ref.c header for the reference conventions they are consistent with:
datasets/utils/bbox.py:39,375-377, datasets/dataset_transforms.py:120-131).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libboxref.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.ref_nms.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def base_anchors(base_size, scales, ratios):
    scales = np.ascontiguousarray(scales, dtype=np.float32)
    ratios = np.ascontiguousarray(ratios, dtype=np.float32)
    out = np.empty((len(ratios) * len(scales), 4), dtype=np.float32)
    lib().ref_base_anchors(ctypes.c_float(base_size), _p(scales), len(scales), _p(ratios), len(ratios), _p(out))
    return out


def anchor_grid(base, featmap_size, stride, valid_size=None):
    base = np.ascontiguousarray(base, dtype=np.float32)
    fh, fw = featmap_size
    vh, vw = valid_size if valid_size is not None else (fh, fw)
    A = base.shape[0]
    anchors = np.empty((fh * fw * A, 4), dtype=np.float32)
    valid = np.empty((fh * fw * A,), dtype=np.uint8)
    lib().ref_anchor_grid(_p(base), A, fh, fw, int(stride), int(vh), int(vw), _p(anchors), _p(valid))
    return anchors, valid


def iou_pairwise(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    if a.shape[0] and b.shape[0]:
        lib().ref_iou_pairwise(_p(a), a.shape[0], _p(b), b.shape[0], _p(out))
    return out


def nms(boxes, scores, thr):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    keep = np.zeros((n,), dtype=np.uint8)
    kept = np.full((n,), -1, dtype=np.int64)
    cnt = lib().ref_nms(_p(boxes), _p(scores), n, ctypes.c_float(thr), _p(keep), _p(kept)) if n else 0
    return keep, kept, cnt


# ---- independent numpy restatement (Appendix B formulas written with array ops) -----------------------
def np_base_anchors(base_size, scales, ratios):
    f = np.float32
    w = h = f(base_size)
    cx = f(0.5) * (w - f(1))
    cy = f(0.5) * (h - f(1))
    h_r = np.sqrt(np.asarray(ratios, dtype=f))
    w_r = f(1) / h_r
    s = np.asarray(scales, dtype=f)
    ws = ((w * w_r[:, None]) * s[None, :]).reshape(-1)
    hs = ((h * h_r[:, None]) * s[None, :]).reshape(-1)
    out = np.stack([cx - f(0.5) * (ws - f(1)), cy - f(0.5) * (hs - f(1)),
                    cx + f(0.5) * (ws - f(1)), cy + f(0.5) * (hs - f(1))], axis=-1)
    return np.rint(out).astype(f)


def np_anchor_grid(base, featmap_size, stride, valid_size=None):
    f = np.float32
    fh, fw = featmap_size
    sx = (np.arange(fw) * stride).astype(f)
    sy = (np.arange(fh) * stride).astype(f)
    xx, yy = np.meshgrid(sx, sy)  # x fastest (dataset_transforms.py:122-126)
    shifts = np.stack([xx.ravel(), yy.ravel(), xx.ravel(), yy.ravel()], axis=-1)
    allb = (np.asarray(base, dtype=f)[None, :, :] + shifts[:, None, :]).reshape(-1, 4)
    vh, vw = valid_size if valid_size is not None else (fh, fw)
    vx = np.arange(fw) < vw
    vy = np.arange(fh) < vh
    vxx, vyy = np.meshgrid(vx, vy)
    valid = (vxx & vyy).ravel()
    valid = np.repeat(valid[:, None], base.shape[0], axis=1).reshape(-1).astype(np.uint8)
    return allb, valid


def np_iou_pairwise(a, b):
    f = np.float32
    a = np.asarray(a, dtype=f)
    b = np.asarray(b, dtype=f)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip((rb - lt) + f(1), f(0), None)
    inter = wh[..., 0] * wh[..., 1]
    area_a = ((a[:, 2] - a[:, 0]) + f(1)) * ((a[:, 3] - a[:, 1]) + f(1))
    area_b = ((b[:, 2] - b[:, 0]) + f(1)) * ((b[:, 3] - b[:, 1]) + f(1))
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / ((area_a[:, None] + area_b[None, :]) - inter)


def py_nms(boxes, scores, thr):
    """O(N^2) pure-Python greedy loop (small N only)."""
    n = len(scores)
    order = sorted(range(n), key=lambda i: (-float(scores[i]), i))
    iou = np_iou_pairwise(boxes, boxes) if n else None
    dead = [False] * n
    keep = np.zeros((n,), dtype=np.uint8)
    kept = []
    for p, i in enumerate(order):
        if dead[i]:
            continue
        keep[i] = 1
        kept.append(i)
        for j in order[p + 1:]:
            if not dead[j] and iou[i, j] > np.float32(thr):
                dead[j] = True
    return keep, np.asarray(kept, dtype=np.int64)


# ---- box delta (de)normalisation: restatement of /root/reference/datasets/utils/bbox.py:118-166 ----------------
# PINNED by golden vectors captured from the reference functions themselves (tests/golden/bbox_norm.npz,
# oracle/gen_golden.py).
def np_bbox_normalize(bbox, means=(0, 0, 0, 0), stds=(1., 1., 1., 1.)):
    """bbox.sub_(means).div_(stds) (bbox.py:136-140); returns a new array (the reference works in place)."""
    b = np.asarray(bbox, dtype=np.float32)
    assert b.shape[1] == 4
    return (b - np.asarray(means, dtype=np.float32)[None, :]) / np.asarray(stds, dtype=np.float32)[None, :]


def np_bbox_denormalize(bbox, means=(0, 0, 0, 0), stds=(1., 1., 1., 1.)):
    """bbox * stds + means with means/stds tiled over 4C columns (bbox.py:157-166)."""
    b = np.asarray(bbox, dtype=np.float32)
    assert b.shape[1] % 4 == 0
    rep = b.shape[1] // 4
    m = np.tile(np.asarray(means, dtype=np.float32), rep)[None, :]
    s = np.tile(np.asarray(stds, dtype=np.float32), rep)[None, :]
    return b * s + m
