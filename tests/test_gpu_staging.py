"""GPU parity of the image batch staging kernel (tdn_collate_images) — bit-exact against the golden batch the
reference's own normalize / flip / pad / collate functions produced (tests/golden/collate.npz) and against the
numpy oracle on ragged and full-size inputs."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available()
    import torch_detection_amd as _T
    return _T


def test_collate_vs_reference_golden(T):
    man = json.load(open(os.path.join(GD, "manifest.json")))["collate"]
    gold = np.load(os.path.join(GD, "collate.npz"))
    tr = T.ImageTransforms(man["means"], man["stds"], man["size_divisor"])
    for tag in ("u8", "f32"):
        imgs = [torch.from_numpy(gold["%s/img%d" % (tag, i)]).cuda() for i in range(3)]
        batch, img_shapes, pad_shapes = tr(imgs, man["flips"])
        assert batch.dtype == torch.float32 and list(batch.shape) == man["batch_shape"]
        assert np.array_equal(batch.cpu().numpy(), gold[tag + "/batch"])          # bit-exact
        assert img_shapes == [(h, w, 3) for h, w in man["sizes_hw"]] and pad_shapes == [(64, 64, 3)] * 3
        # staged output == stage_image(batch) exactly, both 16-bit types
        from torch_detection_amd import ops
        for dt in (torch.bfloat16, torch.float16):
            st, _, _ = tr(imgs, man["flips"], staged=True, dtype=dt)
            assert isinstance(st, T.StagedImages) and st.hw == (64, 64) and st.shape == (3, 3, 64, 64)
            assert torch.equal(st.xp, ops.stage_image(batch, dt))
    # the backbone takes the staged batch directly: same outputs as from the float32 batch, bit for bit
    m = T.ResNet(18).cuda().train()
    m.init_weights()
    st, _, _ = tr(imgs, man["flips"], staged=True)
    with torch.no_grad():
        a, b = m(batch), m(st)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_collate_ragged_vs_oracle(T):
    from oracle import stage_ref as SR
    rng = np.random.RandomState(7)
    sizes = [(1, 1), (33, 95), (64, 64), (17, 130), (96, 31)]
    imgs = [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in sizes]
    flips = [True, False, True, True, False]
    means, stds = (102.9801, 115.9465, 122.7717), (1.0, 1.0, 1.0)
    for div in (32, None, 1):
        tr = T.ImageTransforms(means, stds, div)
        got, _, pads = tr([torch.from_numpy(im).cuda() for im in imgs], flips)
        ref, rpads = SR.np_collate_images(imgs, means, stds, flips, div)
        assert np.array_equal(got.cpu().numpy(), ref)
        assert [p[:2] for p in pads] == rpads
    # more images than one launch takes (TDN_COLLATE_MAX = 16): chunked, same result
    many = [rng.randint(0, 256, size=(8 + i, 40 - i, 3)).astype(np.uint8) for i in range(19)]
    tr = T.ImageTransforms(means, stds, 32)
    got, _, _ = tr([torch.from_numpy(im).cuda() for im in many])
    ref, _ = SR.np_collate_images(many, means, stds, None, 32)
    assert np.array_equal(got.cpu().numpy(), ref)


def test_collate_full_size_and_errors(T):
    """BASELINE's image size (800 x 1333 -> 800 x 1344): checked through properties that need no CPU pass over the
    full batch — linearity in the pixel values and the flip involution — plus one sampled comparison."""
    from torch_detection_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    img = torch.randint(0, 256, (800, 1333, 3), generator=g, dtype=torch.uint8).cuda()
    tr = T.ImageTransforms((123.675, 116.28, 103.53), (58.395, 57.12, 57.375), 32)
    b, _, pads = tr([img, img], [False, True])
    assert tuple(b.shape) == (2, 3, 800, 1344) and pads == [(800, 1344, 3)] * 2
    assert float(b[:, :, :, 1333:].abs().sum()) == 0                              # zero pad
    assert torch.equal(b[1, :, :, :1333], b[0, :, :, :1333].flip(-1))             # flip = mirrored columns
    ref = (img[:5, :7].float().cpu().numpy() - tr.img_means) / tr.img_stds
    assert np.array_equal(b[0, :, :5, :7].permute(1, 2, 0).cpu().numpy(), ref.astype(np.float32))
    with pytest.raises(ValueError):
        ops.collate_images([img.cpu()], (0, 0, 0), (1, 1, 1))
    with pytest.raises(ValueError):
        ops.collate_images([img[:, :, :2].contiguous()], (0, 0, 0), (1, 1, 1))
    with pytest.raises(RuntimeError):
        ops.collate_images([img], (0, 0, 0), (1, 1, 1), batch_hw=(64, 64))
    with pytest.raises(ValueError):
        ops.collate_images([], (0, 0, 0), (1, 1, 1))
