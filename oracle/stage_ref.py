"""CPU restatement (numpy) of the reference's image batch staging — TEST INFRASTRUCTURE ONLY (see oracle/README
rules: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package).

Path restated, one image at a time and then the batch:
    img_normalize          /root/reference/datasets/utils/image.py:87-105    (img - mean) / std  -> float32
    img_flip (horizontal)  /root/reference/datasets/utils/image.py:220-249   np.flip(img, 1)
    img_pad_size_divisor   /root/reference/datasets/utils/image.py:300-347   zero pad bottom/right to a multiple
    HWC -> CHW             /root/reference/datasets/dataset_transforms.py:44 img.transpose(2, 0, 1)
    collate (stack=True)   /root/reference/datasets/loader/collate.py:42-63  pad to the batch max, padding_value 0

PINNED by golden vectors produced by those reference functions themselves (oracle/gen_golden.py ->
tests/golden/collate.npz); tests/test_oracle_golden.py re-checks this file against them bit for bit.
The reference resizes (cv2) between normalize and flip; cv2 is not in the image, and the device path takes
already-resized pixels, so resize is not part of this restatement.
"""
import numpy as np


def np_normalize(img, means, stds):
    """image.py:104-105: ``(img - img_mean) / img_std`` with float32 mean/std arrays (dataset_transforms.py:26-27)
    -> float32.  uint8 and float32 inputs both promote to float32 before the subtraction."""
    m = np.asarray(means, dtype=np.float32)
    s = np.asarray(stds, dtype=np.float32)
    return ((img - m) / s).astype(np.float32)


def np_pad_size_divisor(img, size_divisor, pad_val=0):
    """image.py:340-347 (+ img_pad :300-322)."""
    h, w = img.shape[:2]
    ph = int(np.ceil(h / size_divisor) * size_divisor)
    pw = int(np.ceil(w / size_divisor) * size_divisor)
    out = np.empty((ph, pw) + img.shape[2:], dtype=img.dtype)
    out[...] = pad_val
    out[:h, :w, ...] = img
    return out


def np_collate_images(images, means, stds, flips=None, size_divisor=32, padding_value=0.0):
    """-> (batch float32 (N, 3, Hb, Wb), pad_shapes [(h, w)]): the tensor the reference's loader hands to the
    backbone for one GPU's samples."""
    chw = []
    for i, img in enumerate(images):
        x = np_normalize(np.asarray(img), means, stds)
        if flips is not None and flips[i]:
            x = np.flip(x, 1)
        if size_divisor is not None:
            x = np_pad_size_divisor(x, size_divisor)
        chw.append(np.ascontiguousarray(x.transpose(2, 0, 1)))
    hb = max(c.shape[1] for c in chw)
    wb = max(c.shape[2] for c in chw)
    batch = np.full((len(chw), 3, hb, wb), padding_value, dtype=np.float32)
    for i, c in enumerate(chw):
        batch[i, :, :c.shape[1], :c.shape[2]] = c
    return batch, [(c.shape[1], c.shape[2]) for c in chw]


def np_stage(batch, dtype_name="bfloat16"):
    """The stem kernel's input image: zero halo of 3 px (top/left) and 3/5 px (bottom/right), 4th channel 0,
    values rounded to the 16-bit compute type — (N, H+6, W+8, 4).  Returned as float32 holding the rounded values."""
    import torch
    t = torch.from_numpy(batch).to(getattr(torch, dtype_name)).float().numpy()
    n, _, h, w = t.shape
    xp = np.zeros((n, h + 6, w + 8, 4), dtype=np.float32)
    xp[:, 3:3 + h, 3:3 + w, :3] = t.transpose(0, 2, 3, 1)
    return xp
