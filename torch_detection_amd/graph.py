"""hipGraph capture of a whole training / inference step.

The HIP path issues ~250 small launches per ResNet-50-FPN step; through Python that is ≈7 ms of host time for a
≈4.5 ms GPU step, so an eager loop is host-bound.  All shapes of the hot path are static (padded batches), which is
exactly what a hipGraph wants: capture the step once — forward, backward, the weight-gradient kernels on their side
streams and, under data parallelism, the bucketed RCCL all-reduces on the comm stream — and replay it with one launch.

    step = GraphedStep(lambda: run_one_step(), params=model.parameters())   # run_one_step reads / writes fixed tensors
    step()                                           # replays; falls back to eager calls if capture was refused

Rules (the usual CUDA-graph ones): the callable must use the same tensors every time (copy new data INTO them),
must not synchronise with the host, and parameter gradients must either be views of a persistent buffer
(``dp.attach_reducer``) or be read before the next replay overwrites them.

Weights.  With ``repack=True`` (default) every conv unit re-runs its weight-pack / BN-fold kernels inside the capture
(``functional.ConvUnit.refresh``), so the graph reads the live fp32 parameters and running statistics: an optimizer
step between replays — inside or outside the captured callable — is picked up.  ``repack=False`` freezes the packed
16-bit operands at capture time (valid only while the weights do not change: inference, or a forward+backward
benchmark with static weights); ≈ 0.5 ms of pack / fold launches per ResNet-50-FPN step are then not part of the graph.

One rule is specific to autograd: no autograd graph of these parameters that was built OUTSIDE this object may still be
alive when the step is captured (e.g. the outputs of an ordinary forward pass kept in a variable).  Such a graph keeps
the parameters' AccumulateGrad nodes alive, and those are bound to the stream they were created on; the captured
backward would have to synchronise with that stream, which is not part of the capture — hipStreamEndCapture then
fails (observed as a crash inside ``capture_end``).  ``GraphedStep`` checks for it when it is given ``params`` and
falls back to eager execution with an explanatory error instead of attempting the capture.

Under ``torch.distributed`` the process group's watchdog thread polls the events of collectives that are still
pending.  The capture runs with ``capture_error_mode="thread_local"``: only the capturing thread's calls are policed,
so an event query from the watchdog thread during the capture window is legal (in the default "global" mode any
thread's query invalidates the capture — the abort seen once in four runs).  In addition the warm-up's collectives are
drained deterministically first (``ProcessGroup._wait_for_pending_works``), so the watchdog has nothing left to query.
"""
import sys

import torch


def _accumulator(p):
    """The AccumulateGrad node of leaf ``p`` (created if none is alive)."""
    with torch.enable_grad():
        return p.expand_as(p).grad_fn.next_functions[0][0]


def params_with_live_graph(params):
    """Parameters whose AccumulateGrad node is referenced by someone else — i.e. an autograd graph using them is still
    alive.  A marker is left in the node's ``metadata`` dict (which lives as long as the node); if, after dropping our
    own reference, the node handed out next still carries the marker, it is the same node: something else owns it."""
    held = []
    token = object()
    for p in params:
        if not (isinstance(p, torch.Tensor) and p.requires_grad and p.is_leaf):
            continue
        node = _accumulator(p)
        node.metadata['tdn_graph_probe'] = token
        del node
        again = _accumulator(p)
        if again.metadata.get('tdn_graph_probe') is token:
            held.append(p)
            again.metadata.pop('tdn_graph_probe', None)
        del again
    return held


def _drain_process_group():
    """Wait until the NCCL/RCCL watchdog has retired every pending collective (no timing involved)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    torch.cuda.synchronize()
    try:
        dist.distributed_c10d._get_default_group()._wait_for_pending_works()
    except (AttributeError, RuntimeError):
        pass   # binding absent (older torch) or backend without a watchdog: thread_local capture still covers it


class GraphedStep(object):
    """Captures ``fn`` (after ``warmup`` eager calls on a side stream) and replays it.  ``captured`` tells whether the
    graph exists; if capture was refused or raised, the error is kept in ``error``, printed once, and every call runs
    ``fn`` eagerly.  ``params``: the parameters ``fn`` differentiates (checked for live outside autograd graphs, see the
    module docstring).  ``repack``: see "Weights" above."""

    def __init__(self, fn, warmup=3, verbose=True, params=None, repack=True):
        from . import functional as HF
        self.fn = fn
        self.graph = None
        self.error = None
        self.repack = bool(repack)
        prev_repack = HF.REPACK_IN_CAPTURE
        try:
            if params is not None:
                held = params_with_live_graph(list(params))
                if held:
                    raise RuntimeError(
                        "%d parameter(s) are referenced by an autograd graph built outside GraphedStep (outputs of an "
                        "earlier forward pass still alive?); capturing now would tie the captured backward to that "
                        "graph's stream. Drop those tensors or run the earlier pass under torch.no_grad()" % len(held))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            _drain_process_group()
            g = torch.cuda.CUDAGraph()
            HF.REPACK_IN_CAPTURE = self.repack
            # thread_local: other threads (the process group's watchdog, the allocator's helpers) are not policed
            # during the capture; the launches autograd's device thread makes into the capturing streams are captured
            # either way
            # capture on the stream the warm-up ran on: per-(device, stream) state created during warm-up — the
            # split-K scratch of ops.splitk_workspace, the side / chain stream pools of functional.py — is keyed by the
            # raw stream and must be found again inside the capture (nothing may be allocated there)
            from . import streams
            streams.capture_started(side.cuda_stream)
            try:
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    fn()
            finally:
                streams.capture_finished()
            self.graph = g
        except Exception as e:  # noqa: BLE001 - fall back loudly, never silently
            self.error = e
            if verbose:
                print("torch_detection_amd.graph: hipGraph capture not used (%s: %s); running eager"
                      % (type(e).__name__, e), file=sys.stderr)
            torch.cuda.synchronize()
        finally:
            HF.REPACK_IN_CAPTURE = prev_repack

    @property
    def captured(self):
        return self.graph is not None

    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.fn()


class PreparedStep(object):
    """The fallback when a step cannot be captured into a hipGraph: record the library's launches of ONE eager run of
    ``fn`` — streams, kernels, arguments, cross-stream dependencies — into a launch plan (``tdn_plan_*``,
    include/tdn.h) and, on every call, enqueue the whole list again with a single C call.  The eager path spends
    ~30 us of host time per launch in the operator layer (~6.7 ms per ResNet-50-FPN step, for ~4.4 ms of GPU work);
    the plan spends the ~3 us of the runtime's launch call.

    Ordering and device: a replay enqueues on the RAW streams of the recorded run (the stream that was current while
    recording, and the library's side / chain streams), not on the caller's current stream — work the caller enqueued
    elsewhere must be ordered against it explicitly; ``tdn_plan_run`` selects the recording device for the duration
    of the call.  Only launches made by the recording thread are part of the plan (others are counted by
    ``tdn_plan_stats``'s return value).

    Same contract as ``GraphedStep``: ``fn`` must use the same tensors every time and must not synchronise; its
    results (``param.grad``, whatever it stores) are the tensors of the recorded run, overwritten by each replay —
    they live in a private memory pool owned by this object.  Only the library's own launches are replayed: a step
    that also runs PyTorch GPU ops (a loss, an optimizer, a ``.contiguous()`` copy) cannot be prepared — that is
    checked when ``params`` are given: after recording, the gradients are poisoned, the plan is run once, and every
    gradient must come back bit-identical to the eager run, else ``prepared`` is False and calls run ``fn`` eagerly.
    Weights: the plan re-runs the fold / pack launches of the recorded step (``functional.REPACK_IN_CAPTURE`` is
    honoured only for hipGraph capture; here the units are simply invalidated before the recording so that their
    preparation launches are part of the plan when ``repack`` is set)."""

    def __init__(self, fn, warmup=2, params=None, repack=True, verbose=True, modules=()):
        from . import _lib, functional as HF
        self.fn = fn
        self.plan = None
        self.foreign = 0
        self.error = None
        self.pool = None
        self._lib = _lib.load()
        params = list(params) if params is not None else None
        mt = torch.autograd.is_multithreading_enabled()
        try:
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
            # the backward pass must issue its launches from this thread, in program order (the recorder keeps ONE list)
            torch.autograd.set_multithreading_enabled(False)
            if repack:
                HF.invalidate_packed(*modules)
            self.pool = torch.cuda.MemPool()
            _lib.check(self._lib.tdn_plan_begin(), "tdn_plan_begin")
            try:
                with torch.cuda.use_mem_pool(self.pool):
                    fn()
            finally:
                self.plan = self._lib.tdn_plan_end()
                # replays never go through autograd: the setting is only needed while recording
                torch.autograd.set_multithreading_enabled(mt)
            if not self.plan:
                _lib.check(-1, "tdn_plan_end")
            torch.cuda.synchronize()
            self.stats()
            if self.foreign > 0:
                # launches another thread made while the plan was recorded are counted, not kept: a replay of this plan
                # would silently skip them
                raise RuntimeError("%d launch(es) of the recorded step came from another thread and are not in the "
                                   "plan (recording binds to the recording thread)" % self.foreign)
            if params is not None:
                ref = [p.grad.clone() if p.grad is not None else None for p in params]
                for p in params:
                    if p.grad is not None:
                        p.grad.fill_(float("nan"))
                _lib.check(self._lib.tdn_plan_run(self.plan), "tdn_plan_run")
                torch.cuda.synchronize()
                bad = sum(1 for p, r in zip(params, ref) if r is not None and not torch.equal(p.grad, r))
                if bad:
                    raise RuntimeError("%d gradient(s) differ between the recorded run and its replay: the step does "
                                       "GPU work outside the library's launches" % bad)
        except Exception as e:  # noqa: BLE001 - fall back loudly, never silently
            self.error = e
            if self.plan:
                self._lib.tdn_plan_free(self.plan)
            self.plan = None
            if verbose:
                print("torch_detection_amd.graph: launch plan not used (%s: %s); running eager"
                      % (type(e).__name__, e), file=sys.stderr)
            torch.cuda.synchronize()
        finally:
            torch.autograd.set_multithreading_enabled(mt)

    @property
    def prepared(self):
        return self.plan is not None

    def stats(self):
        """(launches, event records, stream waits) of the plan."""
        import ctypes
        out = (ctypes.c_int32 * 3)()
        self.foreign = int(self._lib.tdn_plan_stats(self.plan, out))
        return tuple(out)

    def __call__(self):
        if self.plan is not None:
            rc = self._lib.tdn_plan_run(self.plan)
            if rc != 0:
                from . import _lib
                _lib.check(rc, "tdn_plan_run")
        else:
            self.fn()

    def close(self):
        if self.plan is not None:
            torch.cuda.synchronize()
            self._lib.tdn_plan_free(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
