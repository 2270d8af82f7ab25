"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce (RCCL over xGMI) overlapped with
backward on a side HIP stream.

The reference contains no communication at all (only ``dist.get_world_size()/get_rank()`` in
datasets/loader/dataset_sampler.py:98,103); what it does fix is the sharding model — one process per GPU,
each rank takes its own ``samples_per_gpu`` slice (dataset_sampler.py:169-172, build_dataloader.py:27-30) and,
because BN runs on running statistics (resnet.py:270-276), the only exchange per step is the parameter
gradient sum.  This module supplies that exchange:

  * all gradients live in ONE flat fp32 buffer laid out in the order the backward pass finishes them
    (FPN -> layer4 -> ... -> stem); the weight-gradient kernels write straight into it (functional.unit_wgrad
    ``sink``), ``param.grad`` are views — no flatten / unflatten copies;
  * the buffer is cut into buckets of ``bucket_bytes``; when the last gradient of a bucket has been enqueued,
    an event is recorded on the compute stream, the comm stream waits on it and issues ``all_reduce`` for that
    bucket — RCCL then runs concurrently with the rest of backward;
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU), so buckets are kept large (default 32 MiB: 4 buckets
    for R50-FPN's 107 MB) and RCCL is left to spread rings/channels over all links;
  * ``finish()`` makes the compute stream wait for the comm stream; the 1/world average happens inside the
    collective (RCCL ``ReduceOp.AVG``), or — gloo, 16-bit wire — in the one pass that touches the reduced buckets;
  * ``comm_dtype=torch.bfloat16`` halves the bytes on the links (53.7 MB instead of 107.4 MB for R50-FPN): each bucket
    is rounded to bfloat16 into a staging buffer on the comm stream, reduced in bfloat16, and written back to the
    fp32 buffer in ``finish()``.  The sum then carries bfloat16 rounding (8 significant bits) — an option for
    link-bound configurations, off by default.

Semantics of the gradient views: with a reducer attached the kernels OVERWRITE ``param.grad`` every backward pass
(beta = 0) — gradient accumulation over several backward passes, or a module applied twice in one step, is not
supported in this mode (use the plain autograd path, which accumulates as usual).
"""
import os
import weakref

import torch
import torch.distributed as dist

from . import functional, streams


class GradReducer(object):
    """Flat-buffer bucketed all-reduce.  ``numels``: gradient sizes in the order they become ready."""

    def __init__(self, numels, device, bucket_bytes=32 << 20, group=None, average=True, dtype=torch.float32,
                 comm_dtype=None):
        self.device = torch.device(device)
        self.group = group
        self.average = average
        if comm_dtype not in (None, torch.float32, torch.bfloat16, torch.float16):
            raise ValueError('comm_dtype must be None, float32, bfloat16 or float16')
        self.comm_dtype = None if comm_dtype in (None, dtype) else comm_dtype
        self.enabled = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.enabled else 1
        # 64-element (256 B) alignment of every slot keeps kernels' float4 stores aligned
        self.offsets, off = [], 0
        for n in numels:
            self.offsets.append(off)
            off += (int(n) + 63) // 64 * 64
        self.numels = [int(n) for n in numels]
        self.flat = torch.zeros(max(off, 64), dtype=dtype, device=self.device)
        self.views = [self.flat[o:o + n] for o, n in zip(self.offsets, self.numels)]
        self.comm_buf = torch.zeros_like(self.flat, dtype=self.comm_dtype) if self.comm_dtype is not None else None
        # bucket sizes are counted in bytes ON THE WIRE
        esize = self.comm_buf.element_size() if self.comm_buf is not None else self.flat.element_size()
        # buckets: consecutive slots, closed when they reach bucket_bytes
        self.bucket_of, self.buckets = [], []   # buckets: [start_off, end_off, nslots]
        cur_start, cur_slots = 0, 0
        for i, (o, n) in enumerate(zip(self.offsets, self.numels)):
            self.bucket_of.append(len(self.buckets))
            cur_slots += 1
            end = o + (n + 63) // 64 * 64
            if (end - cur_start) * esize >= bucket_bytes or i == len(self.numels) - 1:
                self.buckets.append([cur_start, end, cur_slots])
                cur_start, cur_slots = end, 0
        self.use_streams = self.device.type == 'cuda'
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.use_streams else None
        # RCCL divides inside the collective (ReduceOp.AVG): no extra pass over the 107 MB buffer in finish().  gloo has
        # no AVG, and 16-bit wire sums are averaged while they are widened back to fp32.  Also taken with ONE rank (AVG
        # over one rank is the identity): the single-rank RCCL tests and `bench.py --force-reducer` then run exactly the
        # collective the 8-GPU step captures.
        self._avg_in_collective = bool(self.enabled and self.average and self.use_streams and
                                       self.comm_dtype is None and dist.get_backend(group) == 'nccl')
        # When a complete bucket's all-reduce is launched (TDN_DP_LAUNCH):
        #   ready   at its last gradient — the most overlap on paper; but inside the captured step the collective nodes
        #           then sit in the middle of the backward pass and the runtime's 4-lane graph executor
        #           serialises weight-gradient groups and dgrad chains behind them: 437 img/s on ONE rank, where the
        #           collective itself is a copy, against 527 without a reducer
        #   late    (default) when the backward pass enters its last stage (functional.FLUSH_HOOKS: all but one stage's
        #           weight-gradient groups launched): three of R50-FPN's four 32 MiB buckets start there and overlap
        #           layer1 and the tail of the pass, the last one follows from finish(): 509 img/s on one rank
        #   finish  all from finish() (no overlap): 505-510;   <n>: at the n-th stage (1, 2: as bad as ready)
        self.launch_mode = os.environ.get('TDN_DP_LAUNCH', 'late')
        self.defer = self.launch_mode != 'ready'
        self._pending = None
        self._works = []
        self.reset()

    def on_flush_point(self, n, nstages):
        """functional.FLUSH_HOOKS: the backward pass has launched the weight-gradient groups of n of its nstages stages."""
        if not self.defer or self.launch_mode == 'finish':
            return
        at = int(self.launch_mode) if self.launch_mode.isdigit() else max(1, nstages - 1)
        if n >= at:
            self.flush_deferred()

    def flush_deferred(self):
        """Launch the all-reduce of every complete bucket whose launch was put off (``defer``)."""
        todo, self._deferred = self._deferred, []
        for b in todo:
            self._launch(b)

    def reset(self):
        self._deferred = []
        self._pending = [b[2] for b in self.buckets]
        self._works = []
        self._launched = []                            # buckets reduced this step, in launch order
        self._produced = set()                         # slots written this step (the producers report them)
        self._unreported = set()                       # buckets with a producer that did not say which slots it wrote
        self._producers = [{} for _ in self.buckets]   # raw stream -> stream that wrote into each bucket this step

    def mark_ready(self, slot, stream=None):
        """Slot's gradient has been enqueued on ``stream`` (default: the current stream); launch its bucket's
        all-reduce if complete."""
        self.mark_ready_n(self.bucket_of[slot], 1, stream, slots=(slot,))

    def mark_ready_n(self, b, count, stream=None, slots=None):
        """``count`` slots of bucket ``b`` have been enqueued on ``stream`` (one call per conv unit from the backward
        schedule: a unit's weight / affine gradients are consecutive slots, normally of one bucket).  ``slots``: which
        ones — lets ``finish()`` zero the slots nobody wrote this step before it reduces a partly filled bucket.
        Without ``slots`` the bucket is marked as having unreported producers: its stale slots cannot be told apart and
        ``finish()`` refuses to reduce it partly filled."""
        self._pending[b] -= count
        if slots is not None:
            self._produced.update(slots)
        else:
            self._unreported.add(b)
        if self.use_streams and stream is not None:
            self._producers[b][stream.cuda_stream] = stream
        if self._pending[b] == 0:
            if self.defer:
                self._deferred.append(b)
            else:
                self._launch(b)
        elif self._pending[b] < 0:
            raise RuntimeError('GradReducer: bucket %d got more gradients than it has slots in one step '
                               '(call reset()/finish())' % b)

    def _launch(self, b):
        start, end, _ = self.buckets[b]
        buf = self.flat[start:end]
        if not self.enabled:
            return
        self._launched.append(b)
        wire = self.comm_buf[start:end] if self.comm_buf is not None else buf
        if self.use_streams:
            # the bucket's gradients were produced on several streams (the weight-gradient kernels rotate over a
            # pool of side streams): the comm stream waits for every one of them — and for the current stream, which
            # covers buckets flushed from finish() and gradients written by the main stream
            # (through streams.py: the capture rule — no mutual waits between forked streams — is asserted there, and a
            # launch plan being recorded learns of the dependency)
            for st in list(self._producers[b].values()) + [torch.cuda.current_stream(self.device)]:
                streams.wait_stream(self.comm_stream, st)
            with torch.cuda.stream(self.comm_stream):
                if wire is not buf:
                    wire.copy_(buf)            # fp32 -> 16-bit, round to nearest even, on the comm stream
                w = dist.all_reduce(wire, op=dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM,
                                    group=self.group, async_op=True)
        else:
            if wire is not buf:
                wire.copy_(buf)
            w = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(w)

    def finish(self):
        """Wait for every bucket (compute stream waits on the comm stream), average, and re-arm."""
        missing = [i for i, p in enumerate(self._pending) if p > 0]
        if missing:
            # gradients that were never produced this step (e.g. frozen stages): their slots still hold the last
            # step's already averaged values — zero every slot of a partly filled bucket that no producer reported,
            # unconditionally, then reduce
            bad = [b for b in missing if b in self._unreported]
            if bad:
                raise RuntimeError('GradReducer.finish(): buckets %s are partly filled and a producer did not report '
                                   'its slots (mark_ready_n(..., slots=...)): stale gradients cannot be zeroed' % bad)
            for s_, (o, n) in enumerate(zip(self.offsets, self.numels)):
                if self.bucket_of[s_] in missing and s_ not in self._produced:
                    self.flat[o:o + n].zero_()
            for b in missing:
                self._launch(b)
        self.flush_deferred()
        for w in self._works:
            w.wait()
        if self.use_streams and self.enabled:
            streams.wait_stream(torch.cuda.current_stream(self.device), self.comm_stream)
        scale = 1.0 / self.world if (self.average and self.world > 1 and not self._avg_in_collective) else None
        if self.comm_buf is not None and self.enabled:
            # 16-bit sums back into the fp32 gradient views, the 1/world average folded into the same pass
            for b in self._launched:
                start, end, _ = self.buckets[b]
                if scale is None:
                    self.flat[start:end].copy_(self.comm_buf[start:end])
                else:
                    # widen first, scale in fp32: torch.mul(16-bit, scalar, out=fp32) computes in the 16-bit type and
                    # rounds the product once more (bf16 1.0 / 3 -> 0.33398) for world sizes that are not powers of two
                    self.flat[start:end].copy_(self.comm_buf[start:end]).mul_(scale)
        elif scale is not None:
            # only the reduced buckets carry a sum; one pass over them (no pass at all when the collective averaged)
            for b in (self._launched if self.enabled else range(len(self.buckets))):
                start, end, _ = self.buckets[b]
                self.flat[start:end].mul_(scale)
        self.reset()


def attach_reducer(modules, bucket_bytes=32 << 20, group=None, average=True, dtype=None, comm_dtype=None):
    """Wire ``GradReducer`` into HIP-path modules (given in the order their backward runs, e.g. [fpn, resnet]).

    Every conv unit's weight/affine gradients get a slot in the flat buffer (in backward-completion order),
    ``param.grad`` is pre-assigned to the slot view, and the unit reports readiness to the reducer from inside
    the backward schedule.  Returns the reducer; call ``reducer.finish()`` after ``backward()``.
    """
    units = []
    for m in modules:
        net = m.hip_net(dtype)   # the units of the compute dtype the forward will use (default: compute_dtype)
        us = net.units()
        # FPN finishes fpn_convs then laterals; SeqNet finishes last block first, stem last
        order = list(us) if hasattr(net, 'lat') else list(reversed(us))
        if hasattr(net, 'lat'):
            order = list(net.fpn) + list(net.lat)
        units += order
    numels, owners = [], []
    for u in units:
        if u.bias_and_norm:
            raise NotImplementedError('gradient sinks for a conv with both a bias and a norm are not implemented')
        for p in u.params():
            numels.append(p.numel())
            owners.append((u, p))
    dev = units[0].conv.weight.device
    red = GradReducer(numels, dev, bucket_bytes, group, average, comm_dtype=comm_dtype)
    slot = 0
    for u in units:
        ps = u.params()
        views = red.views[slot:slot + len(ps)]
        slots = list(range(slot, slot + len(ps)))
        slot += len(ps)
        w = ps[0]
        if u.is_stem:
            wview = views[0].view(w.shape)
        else:
            wview = views[0].view(w.shape) if u.k == 1 else views[0].view(u.Cout, u.k, u.k, u.Cin // u.groups).permute(0, 3, 1, 2)
        w.grad = wview
        for p, v in zip(ps[1:], views[1:]):
            p.grad = v.view(p.shape)
        u.sink = (views[0].view(w.shape) if u.is_stem else views[0], views[1] if len(ps) > 1 else None,
                  views[2] if len(ps) > 2 else None)

        per_bucket = {}
        for s_ in slots:
            per_bucket.setdefault(red.bucket_of[s_], []).append(s_)
        per_bucket = tuple(per_bucket.items())

        def _cb(unit, stream=None, _pb=per_bucket, _red=red):
            for b_, ss_ in _pb:
                _red.mark_ready_n(b_, len(ss_), stream, ss_)
        u.on_grads = _cb
    # the backward schedule tells the reducer when a stage's weight-gradient groups have been launched (held weakly:
    # a dropped reducer drops out of the hook list)
    ref = weakref.ref(red)

    def _hook(n, nstages, _ref=ref):
        r = _ref()
        if r is not None:
            r.on_flush_point(n, nstages)
    functional.FLUSH_HOOKS[:] = [h for h in functional.FLUSH_HOOKS if getattr(h, '_ref', lambda: None)() is not None]
    _hook._ref = ref
    functional.FLUSH_HOOKS.append(_hook)
    return red


def shard_for_rank(num_samples, rank, world):
    """Rank-contiguous slice of a batch, the sharding rule of DistributedGroupSampler
    (dataset_sampler.py:169-172): rank r takes samples [r*n, (r+1)*n)."""
    n = num_samples // world
    return slice(rank * n, (rank + 1) * n)
