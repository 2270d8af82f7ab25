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
