"""Device-side input staging: the step right before the hot path.

The reference prepares every image on the host — ``ImageTransforms`` (datasets/dataset_transforms.py:12-46):
read, ``img_normalize`` (image.py:87-105), resize (cv2), ``img_flip`` (image.py:220-249), ``img_pad_size_divisor``
(image.py:300-347), HWC->CHW — and ``collate`` (datasets/loader/collate.py:42-63) then pads the samples of a GPU to
their common maximum and stacks them.  Here everything after the resize is ONE kernel launch on the GPU
(``tdn_collate_images``): the loader ships the small uint8 pixels, the float32 batch never exists on the host.

Difference to know: the reference normalises before it resizes; this path takes already-resized pixels (decode and
resize stay with the host image library — cv2 is not part of this repository's scope) and normalises after.  Both
orders are the same affine map up to interpolation rounding; on identical resized pixels the result is bit-identical
to the reference's normalize -> flip -> pad -> transpose -> collate chain (tests/golden/collate.npz).
"""
import numpy as np
import torch

from . import ops
from .functional import StagedImages  # noqa: F401


class ImageTransforms(object):
    """Same constructor as the reference's ``ImageTransforms(img_means, img_stds, size_divisor)``
    (dataset_transforms.py:21-27); ``__call__`` takes the resized images of one GPU's samples instead of a path."""

    def __init__(self, img_means=(0., 0., 0.), img_stds=(1., 1., 1.), size_divisor=None):
        self.img_means = np.array(img_means, np.float32)
        self.img_stds = np.array(img_stds, np.float32)
        self.size_divisor = size_divisor

    def __call__(self, images, flips=None, staged=False, dtype=torch.bfloat16):
        """images: list of CUDA (H_i, W_i, 3) uint8 / float32 tensors.  Returns ``(batch, img_shapes, pad_shapes)``:
        the collated float32 (N, 3, Hb, Wb) batch (or, with ``staged=True``, a ``StagedImages`` holding the stem
        kernel's 16-bit input, which ``ResNet.forward`` accepts as is), and per image the
        (h, w, 3) shape before and after padding to ``size_divisor`` — the reference's ``img_shape`` / ``pad_shape``."""
        d = self.size_divisor
        img_shapes = [tuple(im.shape) for im in images]
        if d is not None:
            pad_shapes = [(-(-h // d) * d, -(-w // d) * d, c) for h, w, c in img_shapes]
        else:
            pad_shapes = list(img_shapes)
        hb, wb = max(p[0] for p in pad_shapes), max(p[1] for p in pad_shapes)
        outs = []
        for i in range(0, len(images), ops.COLLATE_MAX):
            j = i + ops.COLLATE_MAX
            outs.append(ops.collate_images(images[i:j], self.img_means, self.img_stds,
                                           None if flips is None else flips[i:j], (hb, wb), d, staged, dtype))
        batch = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        if staged:
            batch = StagedImages(batch, (hb, wb))   # ResNet.forward takes this directly (no float32 batch at all)
        return batch, img_shapes, pad_shapes
