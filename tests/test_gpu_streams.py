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
    streams.capture_started()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream(dev)
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
    g.replay()
    torch.cuda.synchronize()
    assert float(a[0]) == 1 and float(b[0]) == 3
    # outside a capture nothing is checked (eager events have no such restriction)
    streams.wait(s1, streams.record(s2))
    streams.wait(s2, streams.record(s1))
    torch.cuda.synchronize()
