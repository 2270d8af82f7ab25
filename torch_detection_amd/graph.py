"""hipGraph capture of a whole training / inference step.

The HIP path issues ~250 small launches per ResNet-50-FPN step; through Python that is ≈7 ms of host time for a
≈5.3 ms GPU step, so an eager loop is host-bound.  All shapes of the hot path are static (padded batches), which is
exactly what a hipGraph wants: capture the step once — forward, backward, the weight-gradient kernels on their side
streams and, under data parallelism, the bucketed RCCL all-reduces on the comm stream — and replay it with one launch.

    step = GraphedStep(lambda: run_one_step())       # run_one_step reads / writes fixed tensors
    step()                                           # replays; falls back to eager calls if capture was refused

Rules (the usual CUDA-graph ones): the callable must use the same tensors every time (copy new data INTO them),
must not synchronise with the host, and parameter gradients must either be views of a persistent buffer
(``dp.attach_reducer``) or be read before the next replay overwrites them.  One more, specific to autograd: no
autograd graph of these parameters built on the DEFAULT stream may still be alive when the step is captured (probe
output shapes under ``torch.no_grad()``): its AccumulateGrad nodes would make the captured backward synchronise with
the default stream, which a capture cannot contain — hipStreamEndCapture crashes on it.
"""
import sys
import time

import torch


class GraphedStep(object):
    """Captures ``fn`` (after ``warmup`` eager calls on a side stream) and replays it.  ``captured`` tells whether the
    graph exists; if capture raised, the error is printed once and every call runs ``fn`` eagerly."""

    def __init__(self, fn, warmup=3, verbose=True, settle=None):
        self.fn = fn
        self.graph = None
        self.error = None
        if settle is None:
            import torch.distributed as dist
            settle = 0.5 if (dist.is_available() and dist.is_initialized()) else 0.0
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if settle > 0:
                # under torch.distributed the process group's watchdog thread polls the events of the warm-up's
                # collectives (every ~100 ms) until it has seen them complete; an event query that lands inside the
                # capture window is an illegal call during capture and aborts the process — let the watchdog drain
                time.sleep(settle)
            g = torch.cuda.CUDAGraph()
            # thread_local: other threads (that watchdog, the allocator's helpers) are not policed during the capture;
            # the launches autograd's device thread makes into the capturing streams are captured either way
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            self.graph = g
        except Exception as e:  # noqa: BLE001 - fall back loudly, never silently
            self.error = e
            if verbose:
                print("torch_detection_amd.graph: hipGraph capture failed (%s: %s); running eager"
                      % (type(e).__name__, e), file=sys.stderr)
            torch.cuda.synchronize()

    @property
    def captured(self):
        return self.graph is not None

    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.fn()
