"""Cross-stream ordering of the fused schedule, in one place.

The schedule forks work onto side streams (weight gradients, independent conv branches, per-image forward chains) and
joins it back with events.  Every such dependency goes through these three functions: they do the PyTorch event call —
which is what orders the launches now and what a hipGraph capture records — and, while a launch plan is being recorded
(``graph.PreparedStep``, ``tdn_plan_*`` in include/tdn.h), tell the library about it so that the plan replays the same
dependency with its own events.
"""
import torch

from . import _lib


def record(stream):
    """Record an event on ``stream`` (a ``torch.cuda.Stream``); returns a token for ``wait``."""
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev, _lib.load().tdn_plan_event_record(stream.cuda_stream)


def wait(stream, token):
    """Make ``stream`` wait for the event behind ``token``."""
    ev, plan_id = token
    stream.wait_event(ev)
    if plan_id >= 0:
        _lib.check(_lib.load().tdn_plan_stream_wait(stream.cuda_stream, plan_id), "tdn_plan_stream_wait")


def wait_stream(dst, src):
    """``dst`` waits for everything enqueued on ``src`` so far."""
    wait(dst, record(src))
