"""Cross-stream ordering of the fused schedule, in one place.

The schedule forks work onto side streams (weight gradients, independent conv branches, per-image forward chains) and
joins it back with events.  Every such dependency goes through these three functions: they do the PyTorch event call —
which is what orders the launches now and what a hipGraph capture records — and, while a launch plan is being recorded
(``graph.PreparedStep``, ``tdn_plan_*`` in include/tdn.h), tell the library about it so that the plan replays the same
dependency with its own events.

Capture rule (checked here, at capture time).  Inside a hipGraph capture two FORKED streams must not wait on each
other: if side stream A has waited for an event of side stream B, B must not wait for an event of A (and vice versa)
— synchronise them through the capture's origin stream instead (origin waits for both, both wait for the origin).
Diagnosed on ROCm 7.2 / PyTorch 2.10 (round 3, ``TDN_CHAIN_SYNC`` in functional._blocks_fwd_split): with the two
per-image chains each waiting for the other's per-block event the process died with SIGSEGV inside
``hipStreamEndCapture`` (``torch.cuda.CUDAGraph.capture_end``, no message); the same dependencies expressed one-way
(chain 1 waits for chain 0) or through the origin stream capture, instantiate and replay correctly.  The dependency
graph is acyclic in all three cases — what differs is the runtime's bookkeeping of which capturing streams are tied to
which, which becomes mutual.  A mutual wait between a forked stream and the ORIGIN is the ordinary fork / join and is
fine.  The data-parallel step only adds origin <-> comm-stream and producer -> comm-stream waits (dp.GradReducer, routed
through this module), so it stays inside the rule; the assertion below is what would catch a future schedule that does
not.  The rule is an EMPIRICAL workaround for ROCm 7.2 (inferred from one crashing pattern and the patterns that
pass), not a documented runtime contract.

The origin is the stream ``graph.GraphedStep`` captures on (``capture_started(origin)``); a capture started by someone
else (a user's own ``torch.cuda.graph``) is not policed: without a known origin a forked stream cannot be told from
the origin, and a false refusal is worse than no check.  The edge set lives from ``capture_started`` to
``capture_finished`` and is shared by the threads that launch into that capture (the backward pass runs on autograd's
device thread), under a lock.
"""
import torch

from . import _lib

import threading


class _Capture(object):
    """The capture being policed: raw origin stream (None: nothing is policed) and the (waiting raw stream, recording
    raw stream) pairs seen so far between its forked streams."""
    origin = None
    edges = set()
    lock = threading.Lock()


def _state():
    return _Capture


def record(stream):
    """Record an event on ``stream`` (a ``torch.cuda.Stream``); returns a token for ``wait``."""
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev, _lib.load().tdn_plan_event_record(stream.cuda_stream), stream.cuda_stream


def wait(stream, token):
    """Make ``stream`` wait for the event behind ``token``."""
    ev, plan_id, rec_raw = token
    st = _state()
    if st.origin is not None:
        origin = st.origin
        w = stream.cuda_stream
        if w != origin and rec_raw != origin and w != rec_raw:
            with st.lock:
                bad = (rec_raw, w) in st.edges
                st.edges.add((w, rec_raw))
            if bad:
                raise RuntimeError(
                    "streams.wait: inside a graph capture two forked streams may not wait on each other (stream %#x "
                    "already waited for an event of %#x, now the reverse is requested): hipStreamEndCapture crashes on "
                    "mutual waits between side streams — join them through the capture's origin stream instead "
                    "(see torch_detection_amd/streams.py)" % (rec_raw, w))
    stream.wait_event(ev)
    if plan_id >= 0:
        _lib.check(_lib.load().tdn_plan_stream_wait(stream.cuda_stream, plan_id), "tdn_plan_stream_wait")


def wait_stream(dst, src):
    """``dst`` waits for everything enqueued on ``src`` so far."""
    wait(dst, record(src))


def capture_started(origin=None):
    """Called by graph.GraphedStep right before its capture begins: ``origin`` = raw handle of the stream it captures
    on.  Starts a fresh edge set for this thread."""
    st = _state()
    with st.lock:
        st.origin = origin
        st.edges = set()


def capture_finished():
    st = _state()
    with st.lock:
        st.origin = None
        st.edges = set()
