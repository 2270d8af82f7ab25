"""Box ops of the detection hot path: anchor grid, pairwise IoU, greedy NMS (HIP kernels, bit-exact vs the
C oracle).

The reference has NO box ops (``core/__init__.py`` is an empty file); names and semantics follow the
mmdetection-v0.x lineage its README cites, as fixed in SURVEY.md Appendix B and consistent with the
conventions the reference does pin: inclusive '+1' pixel boxes ``[x1, y1, x2, y2]`` float32
(datasets/utils/bbox.py:39,375-377) and x-fastest grid enumeration (datasets/dataset_transforms.py:120-131).
"""
import numpy as np
import torch

from . import ops


class AnchorGenerator(object):
    """Base anchors (ratio-major, scale-minor) + their expansion over a feature-map grid.

    ``base_anchors`` are a handful of boxes computed once on the host in float32 (Appendix B formula:
    ``round(centre -/+ 0.5*(size-1))`` with round-half-even); the grid expansion — the part that scales with
    the feature map — runs on the GPU (``tdn_anchor_grid``), output order (y, x, anchor).
    """

    def __init__(self, base_size, scales, ratios):
        self.base_size = base_size
        self.scales = np.asarray(scales, dtype=np.float32)
        self.ratios = np.asarray(ratios, dtype=np.float32)
        self.base_anchors = self.gen_base_anchors()
        self._dev = {}

    @property
    def num_base_anchors(self):
        return self.base_anchors.shape[0]

    def gen_base_anchors(self):
        f = np.float32
        w = h = f(self.base_size)
        cx, cy = f(0.5) * (w - f(1)), f(0.5) * (h - f(1))
        h_ratios = np.sqrt(self.ratios)
        w_ratios = f(1) / h_ratios
        ws = ((w * w_ratios[:, None]) * self.scales[None, :]).reshape(-1)
        hs = ((h * h_ratios[:, None]) * self.scales[None, :]).reshape(-1)
        half_w, half_h = f(0.5) * (ws - f(1)), f(0.5) * (hs - f(1))
        boxes = np.stack([cx - half_w, cy - half_h, cx + half_w, cy + half_h], axis=-1)
        return torch.from_numpy(np.rint(boxes).astype(f))

    def _base_on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = self.base_anchors.to(device).contiguous()
        return self._dev[key]

    def grid_anchors(self, featmap_size, stride=16, device='cuda'):
        """(featH*featW*A, 4) float32 anchors on ``device``."""
        return ops.anchor_grid(self._base_on(torch.device(device)), featmap_size, stride)[0]

    def valid_flags(self, featmap_size, valid_size, device='cuda'):
        """(featH*featW*A,) uint8: anchor cell inside the un-padded ``valid_size`` = (valid_h, valid_w) cells."""
        base = self._base_on(torch.device(device))
        return ops.anchor_grid(base, featmap_size, 1, valid_size)[1]

    def grid_anchors_and_flags(self, featmap_size, stride, valid_size, device='cuda'):
        return ops.anchor_grid(self._base_on(torch.device(device)), featmap_size, stride, valid_size)


def anchor_pyramid(generators, featmap_sizes, strides, device='cuda', valid_sizes=None):
    """Anchors (and valid flags) of every level of a feature pyramid from ONE kernel launch: ``generators[l]`` is the
    level's :class:`AnchorGenerator`.  Returns (list of per-level (H_l*W_l*A_l, 4) float32 anchors, list of per-level
    uint8 flags) — views of one allocation each, level order, bit-identical to ``grid_anchors`` / ``valid_flags``
    called level by level."""
    dev = torch.device(device)
    bases = [g._base_on(dev) for g in generators]
    anchors, valid, counts = ops.anchor_pyramid(bases, featmap_sizes, strides, valid_sizes)
    return list(torch.split(anchors, counts)), list(torch.split(valid, counts))


def bbox_overlaps(bboxes1, bboxes2, mode='iou'):
    """Pairwise IoU (N, M) float32 of inclusive-pixel boxes; strict IEEE fp32, Appendix-B operation order."""
    if mode != 'iou':
        raise NotImplementedError("only mode='iou' is implemented")
    return ops.bbox_iou_pairwise(bboxes1.contiguous(), bboxes2.contiguous())


def nms(dets, iou_thr):
    """Greedy NMS on ``dets`` (N, 5) = [x1, y1, x2, y2, score] (or a (boxes, scores) pair).

    Stable descending score order (ties: lower index first), suppress ``iou > iou_thr`` (strict).
    Returns ``(dets[inds], inds)`` like the mmdetection wrapper; use :func:`nms_mask` for the raw keep mask.
    """
    if isinstance(dets, (tuple, list)):
        boxes, scores = dets
        full = None
    else:
        boxes, scores, full = dets[:, :4].contiguous(), dets[:, 4].contiguous(), dets
    keep, kept_idx, num = ops.nms(boxes.contiguous(), scores.contiguous(), iou_thr)
    inds = kept_idx[:int(num.item())]
    if full is None:
        return (boxes[inds], scores[inds]), inds
    return full[inds], inds


def bbox_normalize(bbox, means=[0, 0, 0, 0], stds=[1., 1., 1., 1.]):
    """Normalise box deltas by means / stds — same name, arguments and IN-PLACE behaviour as the reference's
    ``datasets/utils/bbox.py:118-140`` (``bbox.sub_(means).div_(stds)``; the input tensor is modified and returned).
    Bit-identical to the reference on the same input (golden vectors: tests/golden/bbox_norm.npz)."""
    assert bbox.shape[1] == len(means) == len(stds) == 4
    return ops.bbox_normalize_(bbox, means, stds)


def bbox_denormalize(bbox, means=[0, 0, 0, 0], stds=[1., 1., 1., 1.]):
    """De-normalise (A, 4) or class-specific (A, 4C) deltas: ``bbox * stds + means`` —
    ``datasets/utils/bbox.py:143-166``.  Returns a new tensor (the reference does too: it needs the graph)."""
    assert bbox.shape[1] % 4 == 0
    assert len(means) == len(stds) == 4
    return ops.bbox_denormalize(bbox.contiguous(), means, stds)


def nms_mask(boxes, scores, iou_thr):
    """(keep uint8 (N,) in input order, kept indices int64 (N,) in score order padded with -1, count int32 (1,))
    without any host synchronisation."""
    return ops.nms(boxes.contiguous(), scores.contiguous(), iou_thr)
