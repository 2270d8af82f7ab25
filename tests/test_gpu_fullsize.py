"""The BASELINE-size network on the GPU: 2 x 3x800x1344 ResNet-50-FPN in bfloat16 and
4 x 3x800x1344 ResNet-101-FPN in float16, forward + backward.

  * schedule equalities — the default schedule (tile choice per layer, nine-tap / grouped weight-gradient launches on
    side streams, branch streams, per-image forward chains) against the plain one (64x64 tiles everywhere, tap-per-tile
    weight gradients, everything on one stream): forward outputs bit for bit, gradients to fp32 summation-order
    accuracy; and the captured hipGraph replayed against the eager step, bit for bit.  A defect that only shows at
    full size — one of the 27 conv shapes no kernel test runs at this size, a missing stream dependency that needs
    long kernels to bite, a graph replay reading a recycled buffer — breaks one of these.
  * in-situ parity — every fused launch of the full-size forward AND backward recomputed on the CPU from the GPU's
    own operands (tests/parity_util.py), <= 1e-3 per launch.
"""
import os

import pytest
import torch

from golden_util import det_tensor, fill_state_dict, rel_l2

pytestmark = pytest.mark.gpu

H, W = 800, 1344
PLAIN = {"TDN_GEMM_CFG": "0", "TDN_BLOCK_FUSE": "0", "TDN_WGRAD9": "0", "TDN_SIDE_STREAM": "0", "TDN_BRANCH": "0", "TDN_IMG_SPLIT_M": "0",
         "TDN_KG_TILES": "0", "TDN_WGRAD_GROUP": "0", "TDN_SPLITK": "0", "TDN_BWD_SPLIT": "0"}


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available()
    import torch_detection_amd as t
    return t


def _net(T, depth, dtype):
    rb, rf = T.ResNet(depth), T.FPN([256, 512, 1024, 2048], 256, 5)
    sdb = fill_state_dict(rb.state_dict(), 50)
    if dtype == torch.float16:   # keep R101's residual stream inside fp16's range (see tests/test_gpu_fp16.py)
        for k in sdb:
            if k.endswith("bn3.weight"):
                sdb[k] = sdb[k] * 0.25
    rb.load_state_dict(sdb)
    rf.load_state_dict(fill_state_dict(rf.state_dict(), 51))
    rb.cuda().train()
    rf.cuda()
    rb.compute_dtype = rf.compute_dtype = dtype
    return rb, rf


@pytest.mark.parametrize("depth,batch,dtype", [(50, 2, torch.bfloat16), (101, 4, torch.float16)],
                         ids=["r50_bf16_b2", "r101_f16_b4"])
def test_full_size_schedules_agree(T, depth, batch, dtype, monkeypatch):
    rb, rf = _net(T, depth, dtype)
    params = list(rb.parameters()) + list(rf.parameters())
    x = det_tensor((batch, 3, H, W), 700, -2, 2).cuda()
    x[:, :, :, 1333:] = 0                                  # the zero right pad of a 1333-wide image
    with torch.no_grad():
        shapes = [tuple(o.shape) for o in rf(rb(x))]
    scale = 2.0 ** -6 if dtype == torch.float16 else 1.0
    cots = [(det_tensor(s, 710 + i, -1, 1) * scale).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
            for i, s in enumerate(shapes)]
    held = {}

    def step():
        for p in params:
            p.grad = None
        o = rf(rb(x))
        torch.autograd.backward(o, cots)
        held["o"] = o

    def run():
        step()
        torch.cuda.synchronize()
        return [t.detach().clone() for t in held["o"]], [p.grad.clone() for p in params]

    # default schedule, twice (run-to-run reproducibility), then without the in-workgroup split-K tiles, then plain
    o_def, g_def = run()
    o_again, g_again = run()
    assert all(torch.equal(a, b) for a, b in zip(o_def, o_again))
    assert all(torch.equal(a, b) for a, b in zip(g_def, g_again))
    monkeypatch.setenv("TDN_KG_TILES", "0")      # neither in-workgroup nor cross-workgroup split-K: every tile shape
    monkeypatch.setenv("TDN_SPLITK", "0")        # then accumulates K in the same order
    o_nokg, g_nokg = run()
    for k, v in PLAIN.items():
        monkeypatch.setenv(k, v)
    o_plain, g_plain = run()
    for k in PLAIN:
        monkeypatch.delenv(k)
    # every tile shape accumulates K in the same order: outputs identical; weight gradients differ only in how the
    # pixel range is cut into fp32 partial sums
    assert all(torch.equal(a, b) for a, b in zip(o_nokg, o_plain))
    worst = max(rel_l2(a, b) for a, b in zip(g_nokg, g_plain))
    assert worst <= 1e-5, worst
    # the split-K launches (two wave groups inside a workgroup, or several workgroups per tile: layer3 / layer4 / top
    # FPN levels) sum K in pieces: same values to fp32 rounding, so a few
    # 16-bit outputs of those layers land on the neighbouring value — and the layers behind them amplify that to the
    # usual distance between two valid 16-bit evaluations of the net (the bound of the end-to-end forward checks)
    assert max(rel_l2(a.float(), b.float()) for a, b in zip(o_def, o_nokg)) <= 2e-2
    # ---- hipGraph replay == eager ----
    held.clear()      # outputs of the eager runs keep their autograd graph alive: GraphedStep would (rightly) refuse
    gs = T.GraphedStep(step, params=params, repack=True)
    assert gs.captured, gs.error
    gs()
    gs()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(held["o"], o_def))
    assert all(torch.equal(p.grad, g) for p, g in zip(params, g_def))


@pytest.mark.parametrize("depth,batch,dtype", [(50, 2, torch.bfloat16), (50, 1, torch.bfloat16), (101, 1, torch.float16)],
                         ids=["r50_bf16_b2", "r50_bf16_b1", "r101_f16_b1"])
def test_full_size_in_situ(T, depth, batch, dtype):
    """Every launch of the 800x1344 forward and backward against the CPU schedule oracle on the GPU's own operands
    (R101 / float16: one image — the CPU side recomputes ~1.4 TFLOP per image).  r50_bf16_b1 is BASELINE config C2 at
    its own batch (one image: the stem and FPN launches get the tile plans of M = 67,200 instead of 134,400)."""
    import json
    import parity_util
    deep = dtype == torch.float16
    res = parity_util.run_teacher_forced(T, depth, (batch, 3, H, W), dtype=dtype, end_to_end=False,
                                         cot_scale=2.0 ** -6 if deep else 1.0, res_gain=0.25 if deep else 1.0)
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/parity_fullsize_r%d_b%d.json" % (depth, batch), "w") as f:
            json.dump(res, f, indent=1, default=float)
    f, b = res["forward_in_situ"], res["backward_in_situ"]
    assert max(f.values()) <= parity_util.FWD_IN_SITU_TOL, f
    for kind in ("dgrad", "dw", "dgamma", "dbeta_or_dbias"):
        assert b[kind][0] <= parity_util.BWD_IN_SITU_TOL, (kind, b[kind])
    assert b["launches"]["dgrad"] >= (52 if depth == 50 else 103) and b["launches"]["wgrad"] >= 61
