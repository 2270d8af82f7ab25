"""GPU: the gradient-bucket path of the HIP backward (weight-gradient kernels writing straight into the flat
all-reduce buffer, param.grad as views, per-bucket RCCL all-reduce on the side stream) on ONE rank.  A single-rank
RCCL group exercises every call the 8-GPU run makes; the N-rank arithmetic is covered on CPU (tests/test_dp_gloo.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from golden_util import det_tensor, fill_state_dict

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("use_gn", [False, True])
def test_bucketed_reducer_single_rank_rccl(use_gn):
    import torch_detection_amd as T
    from torch_detection_amd import dp
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    if use_gn:   # GroupNorm units: dgamma / dbeta come from the GN kernel on the main stream, dw from the side streams
        rb = T.ResNet(18, use_gn=True)
        rf = T.FPN([64, 128, 256, 512], 256, 5, normalize=dict(type="GN"), use_gn=True)
    else:
        rb, rf = T.ResNet(18), T.FPN([64, 128, 256, 512], 256, 5)
    rb.load_state_dict(fill_state_dict(rb.state_dict(), 50))
    rf.load_state_dict(fill_state_dict(rf.state_dict(), 51))
    rb.to(dev).train()
    rf.to(dev)
    x = det_tensor((2, 3, 128, 128), 700, -2, 2).to(dev)

    def run():
        outs = rf(rb(x))
        cots = [det_tensor(tuple(o.shape), 710 + i, -1, 1).to(dev).to(o.dtype) for i, o in enumerate(outs)]
        torch.autograd.backward(outs, cots)

    run()
    params = list(rb.named_parameters()) + list(rf.named_parameters())
    ref = {id(p): p.grad.detach().clone() for _, p in params}
    for _, p in params:
        p.grad = None

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=dev)
    try:
        red = dp.attach_reducer([rf, rb], bucket_bytes=8 << 20)
        assert len(red.buckets) >= 3
        flat_ptr = red.flat.data_ptr()
        for step in range(2):   # second step: gradients are overwritten in place, not accumulated
            run()
            red.finish()
            torch.cuda.synchronize()
            for n, p in params:
                assert p.grad is not None, n
                lo, hi = flat_ptr, flat_ptr + red.flat.numel() * 4
                assert lo <= p.grad.data_ptr() < hi, "%s.grad is not a view of the flat bucket buffer" % n
                assert torch.equal(p.grad, ref[id(p)]), (n, step)
    finally:
        dist.destroy_process_group()


def test_reducer_path_gradients_vs_cpu_oracle():
    """The gradient-sink path against the ORACLE, not against the plain HIP path: with a reducer attached (weight-
    gradient kernels writing into the flat bucket buffer, per-producer-stream events, RCCL all-reduce of every bucket
    on the comm stream), every weight-gradient member of the backward pass is recomputed on the CPU by
    oracle/sched_ref.py from the GPU's own operands of that launch and compared with what sits in the bucket views
    after ``finish()`` — bound 1e-3 (tests/parity_util.backward_in_situ), the figure of the plain path."""
    import parity_util
    import torch_detection_amd as T
    from oracle import sched_ref as S
    from torch_detection_amd import dp, functional as HF
    dev = torch.device("cuda", 0)
    depth = 50
    rb, rf = T.ResNet(depth), T.FPN([256, 512, 1024, 2048], 256, 5)
    sdb = fill_state_dict(rb.state_dict(), 50)
    sdf = fill_state_dict(rf.state_dict(), 51)
    rb.load_state_dict(sdb)
    rf.load_state_dict(sdf)
    rb.to(dev).train()
    rf.to(dev)
    x = det_tensor((2, 3, 96, 128), 700, -2, 2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=dev)
    try:
        red = dp.attach_reducer([rf, rb], bucket_bytes=8 << 20)
        launches = []
        outs = rf(rb(x.to(dev)))
        cots = [det_tensor(tuple(o.shape), 710 + i, -1, 1).to(dev).to(o.dtype) for i, o in enumerate(outs)]
        HF.DEBUG_BWD = launches
        try:
            torch.autograd.backward(outs, cots)
            red.finish()
            torch.cuda.synchronize()
        finally:
            HF.DEBUG_BWD = None
        # the recorded (dw, dgamma, dbeta) of every member ARE views of the reducer's flat buffer
        lo, hi = red.flat.data_ptr(), red.flat.data_ptr() + red.flat.numel() * 4
        nw = 0
        for rec in launches:
            if rec[0] == 'wgrad':
                nw += 1
                assert all(lo <= t.data_ptr() < hi for t in rec[5] if t is not None)
        assert nw == 61
        sch = S.Sched(sdb, sdf, depth, 5, quant=torch.bfloat16)
        res = parity_util.backward_in_situ(sch, rb, rf, launches, x)
        assert res["launches"]["wgrad"] == 61 and res["launches"]["dgrad"] > 0, res["launches"]
        for kind in ("dgrad", "dw", "dgamma", "dbeta_or_dbias"):
            assert res[kind][0] <= parity_util.BWD_IN_SITU_TOL, (kind, res[kind])
    finally:
        dist.destroy_process_group()


def test_reducer_wait_pattern_passes_the_capture_rule():
    """The N > 1 step adds producer -> comm-stream and comm-stream -> origin waits (dp.GradReducer._launch / finish);
    they go through streams.py, so the capture rule (no mutual waits between forked streams) polices them.  World size 1
    through the real reducer path (what bench.py --force-reducer runs): the capture must succeed and replay."""
    import torch_detection_amd as T
    from torch_detection_amd import dp
    from torch_detection_amd.graph import GraphedStep
    dev = torch.device("cuda", 0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=dev)
    try:
        rb, rf = T.ResNet(50), T.FPN([256, 512, 1024, 2048], 256, 5)
        rb.load_state_dict(fill_state_dict(rb.state_dict(), 60))
        rf.load_state_dict(fill_state_dict(rf.state_dict(), 61))
        rb.to(dev).train()
        rf.to(dev)
        x = det_tensor((2, 3, 128, 160), 720, -2, 2).to(dev)
        red = dp.attach_reducer([rf, rb], bucket_bytes=8 << 20)
        assert red.enabled and red.use_streams and len(red.buckets) >= 3
        params = [p for p in list(rb.parameters()) + list(rf.parameters()) if p.requires_grad]

        def step():
            outs = rf(rb(x))
            torch.autograd.backward([o.float().sum() for o in outs])
            red.finish()

        gs = GraphedStep(step, warmup=2, verbose=True)
        assert gs.captured, "capture with the reducer fell back to eager: %r" % (gs.error,)
        gs()
        torch.cuda.synchronize()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in params)
    finally:
        dist.destroy_process_group()
