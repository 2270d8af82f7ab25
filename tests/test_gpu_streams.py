"""The capture rule of torch_detection_amd/streams.py: inside a hipGraph capture two forked streams may not wait on
each other (hipStreamEndCapture dies with SIGSEGV on such a capture, ROCm 7.2 — diagnosed in round 3 with the
TDN_CHAIN_SYNC knob of functional._blocks_fwd_split); one-way waits and joins through the origin stream are fine.
The assertion turns the crash into a RuntimeError at the offending ``wait``."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mutual_wait_between_forked_streams_is_refused_in_capture():
    from torch_detection_amd import streams
    dev = torch.device("cuda", 0)
    a = torch.zeros(1 << 16, device=dev)
    b = torch.zeros(1 << 16, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    streams.capture_started(side.cuda_stream)      # what graph.GraphedStep does: the origin is the capture's stream
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream(dev)
        assert cur.cuda_stream == side.cuda_stream
        ev = streams.record(cur)
        streams.wait(s1, ev)                 # fork
        streams.wait(s2, ev)
        with torch.cuda.stream(s1):
            a.add_(1)
        with torch.cuda.stream(s2):
            b.add_(2)
        t1, t2 = streams.record(s1), streams.record(s2)
        streams.wait(s2, t1)                 # one way: allowed
        with pytest.raises(RuntimeError, match="may not wait on each other"):
            streams.wait(s1, t2)             # the reverse: refused before it reaches the runtime
        with torch.cuda.stream(s2):
            b.add_(a)
        streams.wait_stream(cur, s1)         # join through the origin
        streams.wait_stream(cur, s2)
        streams.wait(s1, streams.record(cur))    # origin <-> forked stream is the ordinary fork / join
    streams.capture_finished()
    g.replay()
    torch.cuda.synchronize()
    assert float(a[0]) == 1 and float(b[0]) == 3
    # outside a policed capture nothing is checked (eager events have no such restriction)
    streams.wait(s1, streams.record(s2))
    streams.wait(s2, streams.record(s1))
    torch.cuda.synchronize()


def test_graphed_step_captures_with_split_k_scratch(monkeypatch):
    """ADVICE (round 3): with TDN_SPLITK on, the split-K scratch is created during GraphedStep's warm-up, keyed by the
    raw stream; the capture must run on that same stream to find it again (it may not allocate).  captured == True."""
    import torch_detection_amd as T
    from torch_detection_amd import _lib
    from torch_detection_amd.graph import GraphedStep
    monkeypatch.setenv("TDN_SPLITK", "2")
    _lib.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = T.BACKBONES.module_dict["ResNet"](50)
    net.init_weights()
    net.to(dev).train()
    x = torch.randn(1, 3, 128, 160, device=dev)
    params = [p for p in net.parameters() if p.requires_grad]

    def step():
        for p in params:
            p.grad = None
        outs = net(x)
        torch.autograd.backward([o.float().sum() for o in outs])

    gs = GraphedStep(step, warmup=2, verbose=True, params=params)
    assert gs.captured, "capture fell back to eager: %r" % (gs.error,)
    gs()
    torch.cuda.synchronize()
    assert all(p.grad is not None for p in params)
