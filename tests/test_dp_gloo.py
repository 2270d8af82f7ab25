"""CPU, world_size = 2 over gloo: the bucketed flat-buffer gradient exchange (torch_detection_amd/dp.py) gives every
rank the same result as a single process on the concatenated batch (sum over images / world).  The per-rank
gradients come from the CPU oracle; on the GPU box the same GradReducer is driven by the HIP backward
(functional.unit_wgrad sinks) with RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import det_tensor, fill_state_dict, rel_l2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grads(sdb, sdf, x, cots):
    from oracle import torch_ref as O
    _, g = O.resnet_fpn_fwd_bwd(sdb, sdf, x, 18, cots)
    return g


def _worker(rank, world, port, bucket_bytes, out_dir, comm_dtype=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import torch_detection_amd as T
    from torch_detection_amd import dp
    sdb = fill_state_dict(T.ResNet(18).state_dict(), 50)
    sdf = fill_state_dict(T.FPN([64, 128, 256, 512], 256, 5).state_dict(), 51)
    xall = det_tensor((world, 3, 64, 64), 900, -2, 2)
    shapes = [(1, 256, 16, 16), (1, 256, 8, 8), (1, 256, 4, 4), (1, 256, 2, 2), (1, 256, 1, 1)]
    sl = dp.shard_for_rank(world, rank, world)
    cots = [det_tensor(s, 910 + i, -1, 1) for i, s in enumerate(shapes)]
    g = _grads(sdb, sdf, xall[sl], cots)
    names = sorted(g)[::-1]          # any fixed "ready" order
    red = dp.GradReducer([g[k].numel() for k in names], "cpu", bucket_bytes=bucket_bytes, comm_dtype=comm_dtype)
    assert len(red.buckets) > 1
    for step in range(2):            # two steps: the reducer re-arms itself
        for i, k in enumerate(names):
            red.views[i].copy_(g[k].reshape(-1))
            red.mark_ready(i)
        red.finish()
    torch.save({k: red.views[i].clone().view(g[k].shape) for i, k in enumerate(names)},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1 << 20, 8 << 20])
def test_two_rank_average_equals_single_process(tmp_path, bucket_bytes):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), bucket_bytes, str(tmp_path)), nprocs=world, join=True)
    import torch_detection_amd as T
    sdb = fill_state_dict(T.ResNet(18).state_dict(), 50)
    sdf = fill_state_dict(T.FPN([64, 128, 256, 512], 256, 5).state_dict(), 51)
    xall = det_tensor((world, 3, 64, 64), 900, -2, 2)
    shapes = [(1, 256, 16, 16), (1, 256, 8, 8), (1, 256, 4, 4), (1, 256, 2, 2), (1, 256, 1, 1)]
    cots = [det_tensor(s, 910 + i, -1, 1).expand(world, -1, -1, -1).contiguous() for i, s in enumerate(shapes)]
    torch.set_num_threads(4)
    full = _grads(sdb, sdf, xall, cots)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"), weights_only=True)
    for k in full:
        assert torch.equal(r0[k], r1[k]), k                      # every rank holds the same reduced gradient
        assert rel_l2(r0[k], full[k] / world) <= 1e-5, k         # == single-process gradient / world


def test_two_rank_bf16_wire(tmp_path):
    """comm_dtype=bfloat16: half the bytes per bucket, every rank still ends with the same fp32 values, equal to the
    single-process gradient / world to bfloat16 accuracy (two roundings: the operands and the sum)."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), 1 << 20, str(tmp_path), torch.bfloat16), nprocs=world, join=True)
    import torch_detection_amd as T
    sdb = fill_state_dict(T.ResNet(18).state_dict(), 50)
    sdf = fill_state_dict(T.FPN([64, 128, 256, 512], 256, 5).state_dict(), 51)
    xall = det_tensor((world, 3, 64, 64), 900, -2, 2)
    shapes = [(1, 256, 16, 16), (1, 256, 8, 8), (1, 256, 4, 4), (1, 256, 2, 2), (1, 256, 1, 1)]
    cots = [det_tensor(s, 910 + i, -1, 1).expand(world, -1, -1, -1).contiguous() for i, s in enumerate(shapes)]
    torch.set_num_threads(4)
    full = _grads(sdb, sdf, xall, cots)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"), weights_only=True)
    for k in full:
        assert r0[k].dtype == torch.float32 and torch.equal(r0[k], r1[k]), k
        assert rel_l2(r0[k], full[k] / world) <= 8e-3, k         # 2^-8 per rounding


def test_three_rank_bf16_wire_average_is_scaled_in_fp32(tmp_path):
    """A world size that is not a power of two: the 1 / world average of the 16-bit wire format must be taken in fp32
    on the widened sum (one rounding), not in the 16-bit type (bf16 1.0 / 3 -> 0.33398 instead of 0.33333)."""
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), 1 << 20, str(tmp_path), torch.bfloat16), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"), weights_only=True)
    r2 = torch.load(os.path.join(str(tmp_path), "rank2.pt"), weights_only=True)
    third = torch.tensor(1.0 / 3.0, dtype=torch.float32)
    for k in r0:
        assert torch.equal(r0[k], r2[k]), k
        wire = (r0[k] * 3.0).bfloat16().float()          # the bf16 sum the wire carried
        assert torch.equal(r0[k], wire * third), k       # == fp32(sum) * fp32(1 / 3), bit for bit


def test_bf16_wire_halves_the_bucket_bytes():
    from torch_detection_amd import dp
    n = [1 << 18] * 8                                            # 8 MiB of fp32 gradients
    assert len(dp.GradReducer(n, "cpu", bucket_bytes=2 << 20).buckets) == 4
    assert len(dp.GradReducer(n, "cpu", bucket_bytes=2 << 20, comm_dtype=torch.bfloat16).buckets) == 2


def test_unproduced_slots_are_zeroed_not_reduced_again():
    """A slot nobody wrote this step (frozen stage) must not be reduced with last step's averaged contents."""
    from torch_detection_amd import dp
    red = dp.GradReducer([8, 8, 8], "cpu", bucket_bytes=1 << 20)
    for i in range(3):
        red.views[i].fill_(float(i + 1))
        red.mark_ready_n(0, 1, None, [i])
    red.finish()
    assert [float(v[0]) for v in red.views] == [1.0, 2.0, 3.0]
    red.views[0].fill_(5.0)
    red.mark_ready_n(0, 1, None, [0])
    red.finish()
    assert [float(v[0]) for v in red.views] == [5.0, 0.0, 0.0]


def test_reducer_bucket_layout():
    from torch_detection_amd import dp
    red = dp.GradReducer([10, 100, 1000, 5000, 7], "cpu", bucket_bytes=4096)
    assert all(o % 64 == 0 for o in red.offsets)
    assert [v.numel() for v in red.views] == [10, 100, 1000, 5000, 7]
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == red.flat.numel()
    assert sum(b[2] for b in red.buckets) == 5
    for i in range(5):
        red.mark_ready(i)
    red.finish()
    with pytest.raises(RuntimeError):
        for _ in range(2):
            red.mark_ready(0)
            red.mark_ready(0)
    assert dp.shard_for_rank(16, 3, 8) == slice(6, 8)


def _worker_modes(rank, world, port, out_dir):
    """The three launch points of a complete bucket's all-reduce (TDN_DP_LAUNCH) reduce the same values; 'late' holds a
    complete bucket back until the backward schedule reports its second-to-last stage, 'finish' until finish()."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_detection_amd import dp
    numels = [300000, 200000, 100000, 50000]
    vals = [det_tensor((n,), 40 + i + 10 * rank, -1, 1) for i, n in enumerate(numels)]
    res = {}
    for mode in ("ready", "late", "finish", "2"):
        os.environ["TDN_DP_LAUNCH"] = mode
        red = dp.GradReducer(numels, "cpu", bucket_bytes=1 << 20)
        assert len(red.buckets) >= 2 and red.bucket_of[0] == 0 and red.bucket_of[1] != 0
        red.views[0].copy_(vals[0])
        red.mark_ready(0)                       # bucket 0 is complete now
        assert (0 in red._launched) == (mode == "ready"), mode
        red.on_flush_point(1, 4)
        assert (0 in red._launched) == (mode == "ready"), mode
        red.on_flush_point(2, 4)
        assert (0 in red._launched) == (mode in ("ready", "2")), mode
        red.on_flush_point(3, 4)                # the pass enters its last stage
        assert (0 in red._launched) == (mode != "finish"), mode
        for i in range(1, len(numels)):
            red.views[i].copy_(vals[i])
            red.mark_ready(i)
        red.finish()
        res[mode] = [v.clone() for v in red.views]
    os.environ.pop("TDN_DP_LAUNCH")
    for mode in ("late", "finish", "2"):
        assert all(torch.equal(a, b) for a, b in zip(res["ready"], res[mode])), mode
    torch.save(res["late"], os.path.join(out_dir, "modes%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucket_launch_points(tmp_path):
    world = 2
    mp.spawn(_worker_modes, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a = torch.load(os.path.join(str(tmp_path), "modes0.pt"), weights_only=True)
    b = torch.load(os.path.join(str(tmp_path), "modes1.pt"), weights_only=True)
    numels = [300000, 200000, 100000, 50000]
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y)
        want = (det_tensor((numels[i],), 40 + i, -1, 1) + det_tensor((numels[i],), 50 + i, -1, 1)) / 2
        assert torch.allclose(x, want, rtol=0, atol=1e-7)
