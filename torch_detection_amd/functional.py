"""Fused forward/backward schedules of the hot path as ``torch.autograd.Function``s over the HIP kernels.

One Function covers a whole module (all of ResNet, all of FPN, or a single block / ConvModule when those are
called on their own).  Inside a Function the backward is written out explicitly so that everything the
reference leaves to autograd as separate passes is fused into GEMM epilogues:

  forward   conv -> BN(eval, folded) -> (+ residual | + nearest-2x upsample) -> ReLU       one kernel
  backward  dgrad -> (+ residual-branch grad | + external grad | + 2x2 sum-pool) -> ReLU mask of the producer
            wgrad -> split-K reduce -> BN gamma/beta or bias grads                             two kernels

Reference call sites replaced: BasicBlock.forward resnet.py:42-59, Bottleneck.forward resnet.py:97-119,
ResNet.forward resnet.py:253-268, ConvModule.forward layers.py:122-135, FPN.forward fpn.py:88-125.
BN is folded as the per-channel affine the reference's default ``bn_eval=True`` makes it (resnet.py:270-276);
its gamma/beta still get gradients (``bn_frozen=False``).
"""
import contextlib
import os
import threading

import torch
import torch.nn as nn

from . import _lib, ops, streams
from .ops import ADD_NONE, ADD_SAME, ADD_SUMPOOL2, ADD_UP2X

# Test instrumentation: when set to a dict, the forward passes drop references to the tensors they save for
# backward into it (keys 'seq', 'fpn') so parity tests can teacher-force the CPU schedule oracle with them.
DEBUG_CAPTURE = None
# Likewise for the backward: when set to a list, every dgrad launch appends
#   ('dgrad', unit, g, in_hw, addend, addend_mode, mask_src, dx)
# and every weight-gradient member ('wgrad', unit, x_in, g, img_hw, grads) — the operands and results of that launch,
# so a test can recompute each launch on the CPU from the GPU's own inputs (tests/parity_util.py).
DEBUG_BWD = None


# ---------------------------------------------------------------------------------------------------
# prepared conv units (packed weights + folded affine), cached per module and refreshed by version
# ---------------------------------------------------------------------------------------------------
class ConvUnit(object):
    """One conv (+BN | +bias) of the reference, prepared for the GEMM kernels."""

    def __init__(self, conv, bn, relu=False, dtype=torch.bfloat16):
        ops.dtype_code(dtype)   # validates
        self.dtype = dtype      # element type of the packed operands = compute dtype of every launch using them
        if not isinstance(conv, nn.Conv2d):
            raise TypeError('expected nn.Conv2d, got %s' % type(conv))
        kh, kw = conv.kernel_size
        self.conv, self.bn, self.relu = conv, bn, relu
        self.act6 = False       # activation is nn.ReLU6 (ConvModule(activation='relu6')): extra upper clamp at 6
        self.k, self.stride, self.pad = kh, conv.stride[0], conv.padding[0]
        self.Cin, self.Cout = conv.in_channels, conv.out_channels
        self.is_stem = (kh == 7)
        # grouped conv (ResNeXt conv2, resnext.py:26-28,82-83): block-diagonal form over 64-channel blocks
        self.groups = conv.groups
        if conv.groups != 1:
            cpg = conv.in_channels // conv.groups
            if conv.in_channels != conv.out_channels or conv.in_channels % 64 or cpg > 64 or 64 % cpg or kh == 7:
                raise NotImplementedError('grouped convolution %d -> %d with %d groups is not on the HIP path (needs '
                                          'equal channel counts, a multiple of 64, channels per group dividing 64)'
                                          % (conv.in_channels, conv.out_channels, conv.groups))
        # dilation: only as conv3x3_group builds it (padding = dilation, layers.py:20-32); the C ABI reads the dilation
        # of a 3x3 conv from its padding
        if conv.dilation != (1, 1) and not (kh == 3 and conv.dilation == conv.padding and conv.dilation[0] <= 32):
            raise NotImplementedError('dilated convolution is only on the HIP path as a 3x3 with padding = dilation')
        if kh != kw or conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1]:
            raise NotImplementedError('only square kernels / strides / paddings are supported')
        if self.is_stem:
            if (self.Cin, self.stride, self.pad) != (3, 2, 3) or self.Cout % 64:
                raise NotImplementedError('7x7 conv is only supported as the ResNet stem (3->64k, stride 2, pad 3)')
        else:
            if kh not in (1, 3) or self.stride not in (1, 2) or self.pad != (kh // 2) * conv.dilation[0]:
                raise NotImplementedError('conv %dx%d stride %d pad %d is not on the HIP path' %
                                          (kh, kw, self.stride, self.pad))
            if self.Cin % 64 or self.Cout % 64:
                raise NotImplementedError('HIP conv path needs channel counts that are multiples of 64 '
                                          '(got %d -> %d)' % (self.Cin, self.Cout))
        if bn is not None and not isinstance(bn, (nn.BatchNorm2d, nn.GroupNorm)):
            raise NotImplementedError('only BatchNorm2d (eval mode, folded) and GroupNorm are on the HIP path, got %s'
                                      % type(bn).__name__)
        # GroupNorm (layers.py:50-54): per-sample statistics, so nothing folds into the conv — the unit runs as
        # conv (raw) -> tdn_gn_fwd; ``bn`` keeps holding the norm module either way (its weight / bias are the
        # second and third parameter of the unit)
        self.gn = isinstance(bn, nn.GroupNorm)
        # BatchNorm2d in training mode (batch statistics; ResNet(bn_eval=False), resnet.py:270-276) cannot be folded
        # either: decided at refresh() time from ``bn.training`` and run like GroupNorm (conv raw -> tdn_bn_train_fwd)
        self.bnt = False
        # a conv with BOTH a bias and a norm behind it (ConvModule warns about it, layers.py:84-85, and computes it):
        # eval-mode BN folds the bias into the shift; GroupNorm / training-mode BN get it in the raw conv's epilogue
        self.bias_and_norm = bn is not None and conv.bias is not None
        if self.gn:
            C, G = bn.num_channels, bn.num_groups
            if C != self.Cout or C & (C - 1) or not 64 <= C <= 2048 or not bn.affine:
                raise NotImplementedError('GroupNorm(%d groups, %d channels, affine=%s) is not on the HIP path '
                                          '(channels: a power of two in 64..2048)' % (G, C, bn.affine))
        self.key = None
        self.w_fwd = self.w_dgrad = self.scale = self.shift = self.invstd = self.mean = None
        self._fold = self._shift_b = self._mean_b = None   # persistent buffers behind scale / shift / invstd / mean
        self._layout = None
        self.sink = None      # optional (dw, d_affine0, d_affine1) views owned by a gradient bucket (dp.py)
        self.on_grads = None  # optional callback(unit) fired when this unit's grads have been enqueued

    def params(self):
        p = [self.conv.weight]
        if self.conv.bias is not None:
            p.append(self.conv.bias)
        if self.bn is not None:
            p += [self.bn.weight, self.bn.bias]
        return p

    def _version_key(self):
        w = self.conv.weight
        key = [w.data_ptr(), w._version, w.device, self.dtype, self.bn is not None and self.bn.training]
        if self.bn is not None and not self.gn and not self.bn.training:
            for t in (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var):
                key += [t.data_ptr(), t._version]
            if self.conv.bias is not None:
                key += [self.conv.bias.data_ptr(), self.conv.bias._version]
        elif self.conv.bias is not None:
            key += [self.conv.bias.data_ptr()]
        return tuple(key)

    def refresh(self):
        """(Re)pack weights / fold BN if any source tensor changed since the last call.

        While a hipGraph is being captured (``graph.GraphedStep``, or a caller's own ``torch.cuda.graph``) and
        ``REPACK_IN_CAPTURE`` is set, the pack / fold kernels are ALWAYS enqueued, writing into the unit's existing
        buffers: they become graph nodes that read the live fp32 parameters, so a replay after an optimizer update
        computes with the updated weights.  Otherwise the version key decides; updates that bypass the version
        counter (``p.data.copy_``, raw-pointer kernels) need ``invalidate_packed(module)``."""
        self.bnt = self.bn is not None and not self.gn and self.bn.training
        if self.bnt and (self.bn.momentum is None or not self.bn.affine):
            raise NotImplementedError('training-mode BatchNorm2d needs affine=True and a numeric momentum on the HIP '
                                      'path (cumulative averaging / affine=False are not implemented)')
        w = self.conv.weight
        if not w.is_cuda:
            raise RuntimeError('torch_detection_amd modules run on the MI355X HIP path only: move the module to '
                               'a CUDA/HIP device (no CPU fallback)')
        key = self._version_key()
        capturing = REPACK_IN_CAPTURE and torch.cuda.is_current_stream_capturing()
        if key == self.key and not capturing:
            return self
        pending = getattr(_defer, 'pending', None)
        if pending is not None and w.dtype == torch.float32 and not (
                self.is_stem or self.groups > 1 or self.gn or self.bnt or self.bias_and_norm):
            pending.append((self, key))   # plain conv (+ eval-mode BN): prepared with the others in one launch
            return self
        # same sources as last time (only their contents may differ): overwrite the buffers in place — during capture
        # that is what makes the graph self-contained, and addresses other captured nodes hold stay valid
        layout = (w.device, self.dtype, tuple(w.shape), self.gn, self.bnt, self.bn is not None,
                  self.conv.bias is not None)
        reuse = self.w_fwd is not None and layout == self._layout
        self._layout = layout
        with torch.no_grad():
            if w.dtype != torch.float32:
                raise NotImplementedError('parameters must be float32 (bf16 operands are derived on the fly)')
            if self.gn or self.bnt:
                self.scale = self.invstd = self.mean = None   # gamma / beta are read at launch time
                self.shift = self.conv.bias.detach() if self.conv.bias is not None else None   # z = conv + bias
            elif self.bn is not None:
                fold = self._fold if (reuse and self._fold is not None) else None
                self.scale, self.shift, self.invstd = ops.bn_fold(self.bn.weight, self.bn.bias, self.bn.running_mean,
                                                                  self.bn.running_var, self.bn.eps, out=fold)
                self._fold = self.scale._base if self.scale._base is not None else None
                self.mean = self.bn.running_mean
                if self.conv.bias is not None:
                    # bn(conv + b) = scale * conv + (shift + scale * b);  z - mean = conv - (mean - b)
                    b = self.conv.bias.detach()
                    if reuse and self._shift_b is not None:
                        torch.addcmul(self.shift, self.scale, b, out=self._shift_b)
                        torch.sub(self.bn.running_mean, b, out=self._mean_b)
                    else:
                        self._shift_b = self.shift + self.scale * b
                        self._mean_b = self.bn.running_mean - b
                    self.shift, self.mean = self._shift_b, self._mean_b
            else:
                self.scale = self.invstd = self.mean = None
                self.shift = self.conv.bias.detach() if self.conv.bias is not None else None
            if self.is_stem:
                self.w_fwd = ops.pack_stem_weight(w.detach().contiguous(), self.dtype,
                                                  out=self.w_fwd if reuse else None)
                self.w_dgrad = None
            elif self.groups > 1:
                self.w_fwd, self.w_dgrad = ops.pack_gconv_weight(
                    w, self.groups, self.scale, True, self.dtype,
                    out=(self.w_fwd, self.w_dgrad) if reuse and self.w_dgrad is not None else None)
            else:
                self.w_fwd, self.w_dgrad = ops.pack_conv_weight(
                    w, self.scale, True, self.dtype,
                    out=(self.w_fwd, self.w_dgrad) if reuse and self.w_dgrad is not None else None)
        self.key = key
        return self


# Units refreshed inside ``with batched_refresh():`` (the nets' builders) are collected and their fold + pack work is
# enqueued as ONE grouped launch per 30 units when the block exits (ops.prepare_group) — a training step re-derives
# the operands of all 53 + 8 convs of ResNet-50-FPN after every optimizer update.
_defer = threading.local()


class batched_refresh(object):
    def __enter__(self):
        self.outer = getattr(_defer, 'pending', None)
        if self.outer is None:
            _defer.pending = []
        return self

    def __exit__(self, *exc):
        if self.outer is None:
            pending, _defer.pending = _defer.pending, None
            if exc[0] is None:
                _flush_refresh(pending)
        return False


def _flush_refresh(pending):
    by_dtype = {}
    for u, key in pending:
        w = u.conv.weight
        O, I, kh, kw = w.shape
        layout = (w.device, u.dtype, tuple(w.shape), False, False, u.bn is not None, u.conv.bias is not None)
        reuse = u.w_fwd is not None and u.w_dgrad is not None and layout == u._layout
        u._layout = layout
        if not reuse:
            u.w_fwd = torch.empty(O, kh, kw, I, dtype=u.dtype, device=w.device)
            u.w_dgrad = torch.empty(I, kh, kw, O, dtype=u.dtype, device=w.device)
            u._fold = torch.empty(3, O, dtype=torch.float32, device=w.device) if u.bn is not None else None
        bn = None
        if u.bn is not None:
            if u._fold is None:
                u._fold = torch.empty(3, O, dtype=torch.float32, device=w.device)
            bn = (u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var, u.bn.eps)
            u.scale, u.shift, u.invstd = u._fold[0], u._fold[1], u._fold[2]
            u.mean = u.bn.running_mean
        else:
            u.scale = u.invstd = u.mean = None
            u.shift = u.conv.bias.detach() if u.conv.bias is not None else None
        by_dtype.setdefault(u.dtype, []).append((w, bn, u.w_fwd, u.w_dgrad, u._fold if bn is not None else None))
        u.key = key
    with torch.no_grad():
        for dtype, entries in by_dtype.items():
            ops.prepare_group(entries, dtype)


# Whether units re-run their pack / fold kernels inside a hipGraph capture (see ConvUnit.refresh).  GraphedStep sets
# it for the duration of its capture (``repack=True``, the default).
REPACK_IN_CAPTURE = True


def invalidate_packed(*modules):
    """Forget the packed operands / folded BN constants of every conv unit under ``modules``: the next forward
    re-derives them from the fp32 parameters.  Needed after parameter or running-statistics updates that do not bump
    the tensors' version counters (``p.data.copy_()``, ``p.data.mul_()``, kernels writing through raw pointers);
    ordinary in-place updates under ``torch.no_grad()`` (what optimizers do) are picked up automatically."""
    for m in modules:
        for sub in m.modules():
            for u in sub.__dict__.get('_hip_units', {}).values():
                u.key = None


class StagedImages(object):
    """A batch already in the stem kernel's input layout: ``xp`` (N, H+6, W+8, 4) bf16/fp16 with zero halo
    (``ops.stage_image`` / ``ops.collate_images(staged=True)``), ``hw`` = (H, W) of the padded batch.
    ``ResNet.forward`` accepts it in place of the float32 (N, 3, H, W) tensor and skips its own staging launch."""

    def __init__(self, xp, hw):
        H, W = hw
        if xp.dim() != 4 or tuple(xp.shape[1:]) != (H + 6, W + 8, 4) or xp.dtype not in (torch.bfloat16, torch.float16):
            raise ValueError('StagedImages: xp %s %s does not match batch size %s' % (tuple(xp.shape), xp.dtype, (H, W)))
        self.xp, self.hw = xp, (int(H), int(W))

    @property
    def dtype(self):
        return self.xp.dtype

    @property
    def shape(self):
        return (self.xp.shape[0], 3, self.hw[0], self.hw[1])


def pick_dtype(module, x):
    """Compute dtype of a forward call: the input's own 16-bit float dtype when it has one (fp16 in -> fp16 compute,
    like ``model.half()(x.half())`` on the reference), else the module's ``compute_dtype`` attribute (default
    bfloat16).  Parameters stay float32 master copies either way."""
    t = x[0] if isinstance(x, (tuple, list)) else x
    if t.dtype in (torch.float16, torch.bfloat16):   # includes StagedImages
        return t.dtype
    return getattr(module, 'compute_dtype', torch.bfloat16)


def prepare_unit(owner, name, conv, bn, relu=False, dtype=torch.bfloat16):
    """Cached ConvUnit stored on ``owner`` (a module) under ``(name, dtype)``; refreshed if parameters changed."""
    cache = owner.__dict__.setdefault('_hip_units', {})
    u = cache.get((name, dtype))
    if u is None or u.conv is not conv or u.bn is not bn:
        u = ConvUnit(conv, bn, relu, dtype)
        cache[(name, dtype)] = u
    u.relu = relu
    return u.refresh()


def _hw(t):
    return (t.shape[1], t.shape[2])


def _as_nchw(t):
    return t.permute(0, 3, 1, 2)


# Per-call side storage of the GroupNorm units: the raw conv output z and the (mean, rstd) table of every GN unit of
# one forward call, consumed by the matching backward call.  The autograd node that is executing (forward or backward)
# installs its dict here for the duration of the call — thread-local, because backward runs on autograd's thread.
_gn_tls = threading.local()


class gn_scope(object):
    def __init__(self, store):
        self.store = store

    def __enter__(self):
        self.prev = getattr(_gn_tls, 'store', None)
        _gn_tls.store = self.store
        return self.store

    def __exit__(self, *exc):
        _gn_tls.store = self.prev
        # the per-backward dz cache (_gn_dz) must not outlive the call that filled it: a second backward through a
        # retained graph may hand in a different g at the same address
        for k in [k for k in self.store if isinstance(k, tuple) and len(k) == 2 and k[1] == 'dz']:
            del self.store[k]
        return False


def _gn_store(u):
    st = getattr(_gn_tls, 'store', None)
    if st is None:
        raise RuntimeError('GroupNorm unit used outside an autograd node of functional.py')
    return st


def unit_fwd(u, x, addend=None, addend_mode=ADD_NONE, relu=None, out=None):
    relu = u.relu if relu is None else relu
    if relu and u.act6:
        relu = 2                 # nn.ReLU6: the conv epilogue clamps at 6 as well
    if out is not None and (u.gn or u.bnt or u.groups > 1):
        raise NotImplementedError('caller-provided outputs: plain conv (+ folded BN) units only')
    if not (u.gn or u.bnt):
        if u.groups > 1:
            return ops.gconv2d_fwd(x, u.w_fwd, u.groups, u.k, u.stride, u.pad, u.scale, u.shift, addend, addend_mode,
                                   relu)
        return ops.conv2d_fwd(x, u.w_fwd, u.k, u.stride, u.pad, u.scale, u.shift, addend, addend_mode, relu, out=out)
    if u.groups > 1:
        z = ops.gconv2d_fwd(x, u.w_fwd, u.groups, u.k, u.stride, u.pad, None, u.shift)
    else:
        z = ops.conv2d_fwd(x, u.w_fwd, u.k, u.stride, u.pad, None, u.shift)
    if addend is None:
        addend_mode = ADD_SAME
    y, stats = _dyn_norm_fwd(u, z, addend, relu, addend_mode)
    _gn_store(u)[u] = (z, stats)
    return y


def _dyn_norm_fwd(u, z, addend, relu, addend_mode=ADD_SAME):
    """GroupNorm, or BatchNorm2d with batch statistics (running statistics and num_batches_tracked updated like
    nn.BatchNorm2d.forward does), + addend + ReLU on a raw conv output."""
    if u.gn:
        return ops.gn_fwd(z, u.bn.weight, u.bn.bias, u.bn.num_groups, u.bn.eps, addend, relu, addend_mode)
    bn = u.bn
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    out = ops.bn_train_fwd(z, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, addend, relu, addend_mode)
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return out


def _gn_dz(u, g):
    """dL/dz of a GroupNorm unit from g = dL/dy (computed once per backward, shared by its wgrad and dgrad), with
    the affine gradients written to the unit's sink (dp.py) or to fresh tensors."""
    st = _gn_store(u)
    hit = st.get((u, 'dz'))
    if hit is not None and hit[0] == g.data_ptr():
        return hit[1]
    z, stats = st[u]
    dg = db = None
    if u.sink is not None:
        dg, db = u.sink[1], u.sink[2]
    if u.gn:
        dz, dg, db = ops.gn_bwd(g, z, stats, u.bn.weight, u.bn.num_groups, dg, db)
    else:
        dz, dg, db = ops.bn_train_bwd(g, z, stats, u.bn.weight, dg, db)
    st[(u, 'dz')] = (g.data_ptr(), dz, dg, db)
    return dz


def unit_dgrad(u, g, in_hw, addend=None, addend_mode=ADD_NONE, mask_src=None, out=None):
    if addend is None:
        addend_mode = ADD_NONE
    if out is not None and (u.gn or u.bnt or u.groups > 1):
        raise NotImplementedError('caller-provided outputs: plain conv (+ folded BN) units only')
    if u.gn or u.bnt:
        g = _gn_dz(u, g)
    if u.groups > 1:
        return ops.gconv2d_dgrad(g, u.w_dgrad, u.groups, in_hw, u.k, u.stride, u.pad, addend, addend_mode, mask_src)
    dx = ops.conv2d_dgrad(g, u.w_dgrad, in_hw, u.k, u.stride, u.pad, addend, addend_mode, mask_src, out=out)
    if DEBUG_BWD is not None:
        DEBUG_BWD.append(('dgrad', u, g, tuple(in_hw), addend, addend_mode, mask_src, dx))
    return dx


# Weight-gradient kernels run on a side HIP stream: within a backward pass they depend only on tensors that
# already exist (the saved forward input and the activation gradient g), while the dgrad chain that produces the
# next g is the critical path.  Most per-layer launches of this network cannot fill 256 CUs on their own, so
# letting wgrad(l) overlap dgrad(l), dgrad(l-1), ... recovers idle CUs.  TDN_SIDE_STREAM=0 disables it.
_side_streams = {}
_side_rr = [0]


def _num_side_streams():
    return max(0, int(os.environ.get('TDN_SIDE_STREAM', '8')))


def _side_stream(device):
    """Next side stream (round robin over TDN_SIDE_STREAM streams, default 8; 0 disables) for this device /
    main stream.  Keep the count a multiple of 4: the runtime deals streams onto 4 hardware queues, and whole-step
    A/B runs with 5 or 7 streams were 5-8 % slower than with 4 or 8 (8: equal to +1.6 % depending on the box);
    raising GPU_MAX_HW_QUEUES to 8 or 16 cost 35 %."""
    n = _num_side_streams()
    if n == 0:
        return None
    key = (device.index, torch._C._cuda_getCurrentRawStream(device.index))
    pool = _side_streams.get(key)
    if pool is None or len(pool) != n:
        pool = [torch.cuda.Stream(device=device) for _ in range(n)]
        _side_streams[key] = pool
    _side_rr[0] = (_side_rr[0] + 1) % n
    return pool[_side_rr[0]]


# Tensors read by side-stream kernels are kept referenced here until the main stream has joined the side streams:
# their memory then cannot be handed out again before those kernels have run.  (Not ``Tensor.record_stream``: the
# allocator's deferred-free events do not mix with hipGraph capture — capture_end crashed with them.)
_side_refs = {}


def join_side_stream(device):
    """Make the current stream wait for the side streams' weight-gradient kernels (end of a backward pass)."""
    key = (device.index, torch._C._cuda_getCurrentRawStream(device.index))
    pool = _side_streams.get(key)
    if pool:
        cur = torch.cuda.current_stream(device)
        for st in pool:
            streams.wait_stream(cur, st)
    used = _branch_used.pop(key, None)
    if used:
        cur = torch.cuda.current_stream(device)
        for st in used:
            streams.wait_stream(cur, st)
    _side_refs.pop(key, None)


# Independent branches of the forward / dgrad chain (a block's downsample conv beside conv1 -> conv2, its dgrad beside
# the conv3 -> conv2 dgrads, the small FPN output convs beside the lateral chain, the lateral dgrads beside the output
# convs' dgrad chain) run on two streams of their own — not the weight-gradient pool, where they would queue behind long
# wgrad kernels.  Most of these launches cannot fill 256 CUs alone.  TDN_BRANCH=0 keeps everything on the main stream.
_branch_streams = {}
_branch_used = {}     # streams forked since the last join (a capture must not wait on a stream it never forked)
_branch_rr = [0]


class branch(object):
    """``with branch(dev, unit, keep=(tensors...)) as br: y = unit_fwd(...)`` runs the enclosed library launches on a
    branch stream that first waits for everything enqueued on the current stream; ``br.join()`` makes the current
    stream wait for them.  Outputs are allocated on the current stream as usual (only the kernels move), ``keep``
    lists the operands whose memory must outlive the branch kernels.  Units with a run-time norm (GroupNorm, BN in
    training mode) involve PyTorch ops on the current stream and stay inline."""

    def __init__(self, device, unit=None, keep=()):
        self.side = None
        n = int(os.environ.get('TDN_BRANCH', '2'))
        if n <= 0 or (unit is not None and (unit.gn or unit.bnt)):
            return
        key = (device.index, torch._C._cuda_getCurrentRawStream(device.index))
        pool = _branch_streams.get(key)
        if pool is None or len(pool) != n:
            pool = [torch.cuda.Stream(device=device) for _ in range(n)]
            _branch_streams[key] = pool
        _branch_rr[0] = (_branch_rr[0] + 1) % n
        self.side = pool[_branch_rr[0]]
        self.key, self.keep, self.device = key, keep, device

    def __enter__(self):
        if self.side is not None:
            streams.wait_stream(self.side, torch.cuda.current_stream(self.device))
            _side_refs.setdefault(self.key, []).extend(self.keep)
            _branch_used.setdefault(self.key, set()).add(self.side)
            self.prev = _lib.set_stream_override(self.side.cuda_stream)
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            _lib.set_stream_override(self.prev)
            self.done = streams.record(self.side)
        return False

    def join(self):
        if self.side is not None:
            streams.wait(torch.cuda.current_stream(self.device), self.done)


def join_branches(device):
    """End of a forward pass: nothing of it may still be running on a branch stream; operands kept alive for the
    branches are released."""
    key = (device.index, torch._C._cuda_getCurrentRawStream(device.index))
    used = _branch_used.pop(key, None)
    if used:
        cur = torch.cuda.current_stream(device)
        for st in used:
            streams.wait_stream(cur, st)
    _side_refs.pop(key, None)


def _t9_eligible(u, x_in):
    """Mirror of the library's choice of the nine-tap kernel (csrc/conv_wgrad.hip: item_geometry) — only used to put
    those members into a launch group (and side stream) of their own, never for correctness."""
    return (u.k == 3 and u.stride == 1 and u.pad == 1 and u.groups == 1 and not u.is_stem and u.Cout % 64 == 0
            and u.Cin % 64 == 0 and x_in.shape[2] >= 8 and
            (u.Cout % 128 == 0 or os.environ.get('TDN_WGRAD9_64', '1') != '0'))


class WgradQueue(object):
    """Weight-gradient work of one backward pass, collected and launched per GROUP (a ResNet stage, the FPN's output
    convs, its laterals): ``add`` allocates the outputs and records one member, ``flush`` enqueues the members as
    grouped launches (ops.wgrad_group) on side streams — the nine-tap-eligible 3x3 convs as one group, everything
    else as another — after an event on the current stream (every recorded ``g`` exists by then).

    Within a backward pass the weight gradients depend only on tensors that already exist (the saved forward input
    and the activation gradient g), while the dgrad chain that produces the next g is the critical path; a layer's
    weight gradient alone cannot fill 256 CUs without cutting its pixel range into many fp32 partial slabs, a stage's
    can.  TDN_WGRAD_GROUP=0 flushes after every member (one launch group per layer)."""

    def __init__(self, device):
        self.device = device
        self.members = []      # (unit, item, side-stream group tag)
        self.per_layer = os.environ.get('TDN_WGRAD_GROUP', '1') == '0'
        # called at the start of every flush: the owner's way of making the CURRENT stream see every recorded g.
        # The per-image dgrad chains write g on streams of their own; a flush that only orders the side stream
        # behind the main stream would race with them (per-layer flushes happen in the middle of a block).
        self.pre_flush = None

    def add(self, u, x_in, g, img_hw=None):
        """Gradients aligned with ``u.params()`` (written when the group is flushed).  With a gradient sink attached
        (dp.py) the kernels write straight into the bucket views and ``None`` is returned for autograd
        (``param.grad`` already aliases the views)."""
        sink = u.sink
        dev = g.device
        dg = db = None
        gn_affine = None
        dyn = u.gn or u.bnt
        if dyn:
            g = _gn_dz(u, g)                       # dL/dz; the affine gradients come from the norm kernel
            gn_affine = _gn_store(u)[(u, 'dz')][2:]
        if sink is not None:
            dw, d0, d1 = sink
            if dyn:
                pass                                # dgamma / dbeta were written into the sink by _gn_dz
            elif u.bn is not None:
                dg, db = d0, d1
            else:
                db = d0
        else:
            # outputs are allocated on the main stream (their consumers live there); only the kernels move
            dw = torch.empty((u.Cout, 3, 7, 7) if u.is_stem else (u.Cout, u.k, u.k, u.Cin // u.groups),
                             dtype=torch.float32, device=dev)
            if u.bn is not None and not dyn:
                dg = torch.empty(u.Cout, dtype=torch.float32, device=dev)
            if (u.bn is not None and not dyn) or u.conv.bias is not None:
                db = torch.empty(u.Cout, dtype=torch.float32, device=dev)
        want_db = db is not None
        if u.is_stem:
            it, dw, dg, db = ops.stem_conv_wgrad_item(x_in, g, u.w_fwd, img_hw, u.scale, u.mean, u.invstd, dw, dg, db,
                                                      want_dbeta=want_db)
            dw_view = dw
        else:
            dw4 = dw.view(u.Cout, u.k, u.k, u.Cin // u.groups)
            if u.groups > 1:
                it, dw4, dg, db = ops.gconv2d_wgrad_item(x_in, g, u.w_fwd, u.groups, u.k, u.stride, u.pad, u.scale,
                                                         u.mean, u.invstd, dw4, dg, db, want_dbeta=want_db)
            else:
                it, dw4, dg, db = ops.conv2d_wgrad_item(x_in, g, u.w_fwd, u.k, u.stride, u.pad, u.scale, u.mean,
                                                        u.invstd, dw4, dg, db, want_dbeta=want_db)
            # 1x1: [Cout,1,1,Cin] is byte-identical to the contiguous OIHW parameter -> view, so autograd's
            # layout contract holds and AccumulateGrad does not copy
            dw_view = dw4.view(u.Cout, u.Cin // u.groups, 1, 1) if u.k == 1 else dw4.permute(0, 3, 1, 2)
        # every tensor a member's raw pointers refer to stays referenced until the side streams have been joined
        # (the kernels run later, on another stream: nothing they touch may be recycled before; see _side_refs)
        key = (dev.index, torch._C._cuda_getCurrentRawStream(dev.index))
        _side_refs.setdefault(key, []).extend(t for t in (x_in, g, dw, dg, db, u.w_fwd, u.scale, u.mean, u.invstd)
                                              if t is not None)
        self.members.append((u, it, 1 if _t9_eligible(u, x_in) else 0))
        if self.per_layer:
            self.flush()
        if DEBUG_BWD is not None and not dyn:
            DEBUG_BWD.append(('wgrad', u, x_in, g, img_hw, (dw_view, dg, db)))
        if sink is not None:
            return [None] * len(u.params())
        if u.bias_and_norm:
            # dyn: db = sum of dL/dz = the bias gradient.  Folded eval-mode BN: dL/dz = g * scale, so the bias
            # gradient is scale * dbeta — left as None here and filled in by the caller once the side stream has
            # been joined
            return [dw_view, db, gn_affine[0], gn_affine[1]] if dyn else [dw_view, None, dg, db]
        if dyn:
            return [dw_view, gn_affine[0], gn_affine[1]]
        if u.bn is not None:
            return [dw_view, dg, db]
        if u.conv.bias is not None:
            return [dw_view, db]
        return [dw_view]

    def flush(self):
        """Enqueue everything recorded so far."""
        if not self.members:
            return
        if self.pre_flush is not None:
            self.pre_flush()
        members, self.members = self.members, []
        dev = self.device
        ev = None
        for tag in (1, 0):
            grp = [(u, it) for u, it, t in members if t == tag]
            if not grp:
                continue
            side = _side_stream(dev)
            prev = None
            if side is not None:
                if ev is None:
                    ev = streams.record(torch.cuda.current_stream(dev))   # every g recorded so far is ready there
                streams.wait(side, ev)
                prev = _lib.set_stream_override(side.cuda_stream)
            try:
                ops.wgrad_group([it for _, it in grp], grp[0][0].dtype, dev)
            finally:
                if side is not None:
                    _lib.set_stream_override(prev)
            for u, _ in grp:
                if u.on_grads is not None:
                    u.on_grads(u, side)


def unit_wgrad(u, x_in, g, img_hw=None, queue=None):
    """Weight / affine gradients of one unit: recorded in ``queue`` (launched at its next flush), or launched now."""
    if queue is not None:
        return queue.add(u, x_in, g, img_hw)
    q = WgradQueue(g.device)
    grads = q.add(u, x_in, g, img_hw)
    q.flush()
    return grads


# ---------------------------------------------------------------------------------------------------
# single ConvModule
# ---------------------------------------------------------------------------------------------------
class ConvUnitFunction(torch.autograd.Function):
    """ConvModule.forward (layers.py:122-135) for conv(+bias | +BN eval)(+ReLU)."""

    @staticmethod
    def forward(ctx, unit, x, *params):
        ctx.gn_saved = {}
        with gn_scope(ctx.gn_saved):
            return ConvUnitFunction._forward(ctx, unit, x, *params)

    @staticmethod
    def _forward(ctx, unit, x, *params):
        xh = ops.to_nhwc_bf16(x, unit.dtype)
        y = unit_fwd(unit, xh)       # ReLU / ReLU6 fused (relu code 2 for nn.ReLU6, see unit_fwd)
        ctx.unit, ctx.xh, ctx.y = unit, xh, y
        return _as_nchw(y)

    @staticmethod
    def backward(ctx, dy):
        with gn_scope(ctx.gn_saved):
            return ConvUnitFunction._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        u = ctx.unit
        g = ops.to_nhwc_bf16(dy, u.dtype)
        if u.relu:
            g = ops.act_mask(g, ctx.y, 6.0) if u.act6 else ops.add_relu_mask(g, None, ctx.y)
        grads = unit_wgrad(u, ctx.xh, g)
        dx = _as_nchw(unit_dgrad(u, g, _hw(ctx.xh))) if ctx.needs_input_grad[1] else None
        join_side_stream(g.device)
        if u.bias_and_norm and grads[1] is None and u.sink is None:
            grads[1] = u.scale * grads[3]          # eval-mode BN behind a biased conv (see unit_wgrad)
        return (None, dx) + tuple(grads)


class PreActUnit(object):
    """ConvModule(activate_last=False) (layers.py:129-134): [norm on the INPUT] -> [ReLU | ReLU6] -> conv (+bias).
    ``conv`` is the prepared conv(+bias) unit, ``norm`` the nn.BatchNorm2d / nn.GroupNorm over in_channels or None,
    ``act`` 0 none / 1 ReLU / 2 ReLU6."""

    def __init__(self, conv_unit, norm, act):
        self.conv, self.norm, self.act = conv_unit, norm, act
        if norm is not None and not isinstance(norm, (nn.BatchNorm2d, nn.GroupNorm)):
            raise NotImplementedError('pre-activation norm %s is not on the HIP path' % type(norm).__name__)
        if isinstance(norm, nn.GroupNorm):
            C = norm.num_channels
            if C & (C - 1) or not 64 <= C <= 2048 or not norm.affine:
                raise NotImplementedError('GroupNorm(%d channels) is not on the HIP path (a power of two in 64..2048)'
                                          % C)
        if isinstance(norm, nn.BatchNorm2d) and (not norm.affine or not norm.track_running_stats):
            raise NotImplementedError('pre-activation BatchNorm2d needs affine=True and running statistics')

    def params(self):
        p = self.conv.params()
        if self.norm is not None:
            p = p + [self.norm.weight, self.norm.bias]
        return p


class PreActConvFunction(torch.autograd.Function):
    """norm -> activate -> conv of ConvModule(activate_last=False) as one autograd node."""

    @staticmethod
    def forward(ctx, pu, x, *params):
        u, norm, act = pu.conv, pu.norm, pu.act
        xh = ops.to_nhwc_bf16(x, u.dtype)
        stats = fold = None
        if isinstance(norm, nn.GroupNorm):
            h, stats = ops.gn_fwd(xh, norm.weight, norm.bias, norm.num_groups, norm.eps, None, act)
            kind = 'gn'
        elif norm is not None and norm.training:
            if norm.momentum is None:
                raise NotImplementedError('cumulative-average BatchNorm2d is not on the HIP path')
            h, stats = ops.bn_train_fwd(xh, norm.weight, norm.bias, norm.running_mean, norm.running_var,
                                        norm.momentum, norm.eps, None, act)
            if norm.num_batches_tracked is not None:
                norm.num_batches_tracked.add_(1)
            kind = 'bnt'
        elif norm is not None:
            scale, shift, invstd = ops.bn_fold(norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps)
            fold = (scale, invstd, norm.running_mean)
            h = ops.channel_affine_fwd(xh, scale, shift, act)
            kind = 'bn'
        else:
            h = ops.add_relu_mask(xh, None, xh) if act >= 1 else xh
            kind = 'none'
        if act == 2 and kind == 'none':
            h = ops.clamp_max_(h, 6.0)       # exact: h holds 16-bit input values, min(x, 6) needs no rounding
        y = unit_fwd(u, h)
        ctx.pu, ctx.kind, ctx.xh, ctx.h, ctx.stats, ctx.fold = pu, kind, xh, h, stats, fold
        return _as_nchw(y)

    @staticmethod
    def backward(ctx, dy):
        pu, kind, xh, h = ctx.pu, ctx.kind, ctx.xh, ctx.h
        u, norm, act = pu.conv, pu.norm, pu.act
        g = ops.to_nhwc_bf16(dy, u.dtype)
        grads = unit_wgrad(u, h, g)
        norm_grads = []
        dx = None
        if ctx.needs_input_grad[1] or norm is not None:
            gh = unit_dgrad(u, g, _hw(h))
            if act >= 1:
                gh = ops.act_mask(gh, h, 6.0 if act == 2 else float('inf'))
            if kind == 'gn':
                dx, dg, db = ops.gn_bwd(gh, xh, ctx.stats, norm.weight, norm.num_groups)
                norm_grads = [dg, db]
            elif kind == 'bnt':
                dx, dg, db = ops.bn_train_bwd(gh, xh, ctx.stats, norm.weight)
                norm_grads = [dg, db]
            elif kind == 'bn':
                scale, invstd, mean = ctx.fold
                dx, dg, db = ops.channel_affine_bwd(gh, xh, scale, mean, invstd)
                norm_grads = [dg, db]
            else:
                dx = gh
        join_side_stream(g.device)
        dx = _as_nchw(dx) if (dx is not None and ctx.needs_input_grad[1]) else None
        return (None, dx) + tuple(grads) + tuple(norm_grads)


# ---------------------------------------------------------------------------------------------------
# residual blocks / ResNet
# ---------------------------------------------------------------------------------------------------
class BlockSpec(object):
    """Prepared BasicBlock (u3 is None) or Bottleneck of the reference (resnet.py:9-119)."""

    def __init__(self, kind, u1, u2, u3, ud, stride):
        self.kind, self.u1, self.u2, self.u3, self.ud, self.stride = kind, u1, u2, u3, ud, stride

    def units(self):
        return [u for u in (self.u1, self.u2, self.u3, self.ud) if u is not None]


class SeqNet(object):
    """Prepared straight-line program: optional stem, residual blocks, and which block outputs are returned."""

    def __init__(self, stem, blocks, out_blocks):
        self.stem, self.blocks, self.out_blocks = stem, blocks, list(out_blocks)
        self.dtype = (stem or blocks[0].u1).dtype

    def units(self):
        us = [self.stem] if self.stem is not None else []
        for b in self.blocks:
            us += b.units()
        return us

    def params(self):
        ps = []
        for u in self.units():
            ps += u.params()
        return ps


def _block_fusable(b, C4):
    """Stride-1 Bottleneck without a downsample branch, plain convs (+ folded eval-mode BN), ReLU activations, and a
    one-launch kernel for its width in this build (csrc/conv_block.hip): conv1 -> conv2 -> conv3 + residual run as
    ONE launch, forward (ops.bottleneck_fwd) and input-gradient chain (ops.bottleneck_dgrad).  TDN_BLOCK_FUSE=0
    keeps the per-conv launches (A/B runs, and the reference side of tests/test_gpu_block.py)."""
    knob = os.environ.get('TDN_BLOCK_FUSE', '1')     # 0: off; 1: every width the build has; 64 / 128: up to that width
    if b.kind != 'bottleneck' or b.ud is not None or b.stride != 1 or knob == '0':
        return False
    if knob not in ('', '1') and b.u1.Cout > int(knob):
        return False
    u1, u2, u3 = b.u1, b.u2, b.u3
    for u in (u1, u2, u3):
        if u.gn or u.bnt or u.groups != 1 or u.act6 or u.bias_and_norm or u.is_stem or u.stride != 1:
            return False
    C = u1.Cout
    if not (u1.k == 1 and u3.k == 1 and u2.k == 3 and u2.pad == 1 and u1.Cin == C4 and u3.Cout == C4 and
            C4 == 4 * C and u2.Cin == C and u2.Cout == C and u3.Cin == C):
        return False
    return ops.bottleneck_supported(1, 1, C)


def _block_head_fusable(b, Cin):
    """The stage's first Bottleneck where it keeps the resolution (layer1.0, resnet.py:130-136: a 1x1 conv + BN
    downsample because inplanes != 4 * planes): conv1 -> conv2 -> conv3 + residual run as one launch whose residual is
    the downsample branch — computed inside the launch, forward and backward (default, TDN_BLOCK_HEAD=2; 2f: forward
    only) or by a launch of its own (TDN_BLOCK_HEAD=1): ops.bottleneck_head_fwd / _dgrad.  TDN_BLOCK_HEAD=0 (or
    TDN_BLOCK_FUSE=0) keeps the per-conv launches.  One box, interleaved: 0 -> 510.7, 1 -> 518.1, 2f -> 520.0,
    2 -> 523.9 img/s."""
    if b.kind != 'bottleneck' or b.ud is None or b.stride != 1:
        return False
    if os.environ.get('TDN_BLOCK_FUSE', '1') == '0' or os.environ.get('TDN_BLOCK_HEAD', '1') == '0':
        return False
    u1, u2, u3, ud = b.u1, b.u2, b.u3, b.ud
    for u in (u1, u2, u3, ud):
        if u.gn or u.bnt or u.groups != 1 or u.act6 or u.bias_and_norm or u.is_stem or u.stride != 1:
            return False
    C = u1.Cout
    if not (u1.k == 1 and u3.k == 1 and ud.k == 1 and u2.k == 3 and u2.pad == 1 and u1.Cin == Cin and ud.Cin == Cin and
            u3.Cout == 4 * C and ud.Cout == 4 * C and u2.Cin == C and u2.Cout == C and u3.Cin == C):
        return False
    return ops.bottleneck_head_supported(1, 1, Cin, C)


# set by SeqNetFunction.forward for the duration of the call: will a backward pass follow (does any input / parameter of
# the node need a gradient)?  Inside autograd.Function.forward grad mode is always off, so it cannot be asked there.
_WANT_BWD = [False]


def _block_bits_on():
    """ReLU bit planes for the one-launch blocks (default on): the forward launch also writes h1 > 0, h2 > 0 and x > 0
    as 1-bit planes and the backward launch reads those instead of the 16-bit tensors (1/16 of the mask bytes: 37 % of
    that launch's HBM traffic).  TDN_BLOCK_BITS=0: 16-bit mask sources."""
    return os.environ.get('TDN_BLOCK_BITS', '1') != '0'


def _block_fwd(x, b, bufs=None):
    """One residual block.  ``bufs`` = caller-provided (h1, h2, out, res[, bits]) outputs — one image's slices of batch
    tensors when the images of a batch run as separate chains (ImageSplit); everything then stays on the routed
    stream (no branch stream for the downsample conv)."""
    res, br = x, None
    o1 = o2 = o3 = ores = bits = None
    if bufs is not None:
        o1, o2, o3, ores = bufs[:4]
        bits = bufs[4] if len(bufs) > 4 else None
    if _block_fusable(b, x.shape[3]):
        u1, u2, u3 = b.u1, b.u2, b.u3
        if bufs is None and _block_bits_on() and _WANT_BWD[0]:
            bits = ops.bottleneck_bit_planes(x.shape[0], x.shape[1], x.shape[2], u1.Cout, x.device)
        h1, h2, out = ops.bottleneck_fwd(x, u1.w_fwd, u2.w_fwd, u3.w_fwd,
                                         (u1.scale, u1.shift, u2.scale, u2.shift, u3.scale, u3.shift),
                                         outs=(o1, o2, o3) if bufs is not None else None, bits=bits)
        if bufs is None and bits is not None:
            h1._tdn_bits = bits       # travels with the saved activation to the backward launch
        return out, (x, h1, h2, out)
    if _block_head_fusable(b, x.shape[3]):
        u1, u2, u3 = b.u1, b.u2, b.u3
        down = None
        if os.environ.get('TDN_BLOCK_HEAD', '2') == '1':
            res = unit_fwd(b.ud, x, relu=False, out=ores)
        else:
            res, down = None, (b.ud.w_fwd, b.ud.scale, b.ud.shift)
        if bufs is None and _block_bits_on() and _WANT_BWD[0]:
            bits = ops.bottleneck_bit_planes(x.shape[0], x.shape[1], x.shape[2], u1.Cout, x.device)[:2]
        h1, h2, out = ops.bottleneck_head_fwd(x, u1.w_fwd, u2.w_fwd, u3.w_fwd,
                                              (u1.scale, u1.shift, u2.scale, u2.shift, u3.scale, u3.shift), res,
                                              outs=(o1, o2, o3) if bufs is not None else None, bits=bits, down=down)
        if bufs is None and bits is not None:
            h1._tdn_bits = bits
        return out, (x, h1, h2, out)
    if b.ud is not None:
        if bufs is not None:
            res = unit_fwd(b.ud, x, relu=False, out=ores)
        else:
            with branch(x.device, b.ud, (x,)) as br:    # the downsample conv runs beside conv1 (-> conv2)
                res = unit_fwd(b.ud, x, relu=False)
    if b.kind == 'bottleneck':
        h1 = unit_fwd(b.u1, x, relu=True, out=o1)
        h2 = unit_fwd(b.u2, h1, relu=True, out=o2)
        if br is not None:
            br.join()
        out = unit_fwd(b.u3, h2, res, ADD_SAME, True, out=o3)
        return out, (x, h1, h2, out)
    h1 = unit_fwd(b.u1, x, relu=True, out=o1)
    if br is not None:
        br.join()
    out = unit_fwd(b.u2, h1, res, ADD_SAME, True, out=o3)
    return out, (x, h1, None, out)


# Small-M stages (layer3 / layer4 at the BASELINE batch of 2: 8,400 and 2,100 pixels) cannot fill 256 CUs from one
# launch — 132 ... 528 workgroups with short K loops, each launch paying its own ramp, tail and dependency gap.  The
# images of a batch are independent (eval-mode BN), so from the first block whose batch has at most TDN_IMG_SPLIT_M
# pixels on, the forward chain runs once per image, each chain on a stream of its own: the launches of one image fill
# the gaps of the other's.  Outputs are the image slices of ordinary batch tensors (backward is unchanged).
_split_streams = {}


def _img_split_m():
    return int(os.environ.get('TDN_IMG_SPLIT_M', '300000'))


def _splittable(b):
    return all(not (u.gn or u.bnt) and u.groups == 1 for u in b.units())


def _blocks_fwd_split(blocks, cur, pre=None):
    """Forward of ``blocks`` on batch ``cur`` (N >= 2) as N per-image chains on N streams; returns (out, saved).
    ``pre(a, e)``: optional producer of cur[a:e], launched at the head of that image range's chain (the stem)."""
    dev = cur.device
    N = cur.shape[0]
    key = (dev.index, torch._C._cuda_getCurrentRawStream(dev.index))
    # two chains whatever the batch: at 4 images per GPU four chains measured worse than two (R101 fp16: 311 vs 350
    # img/s with the backward chains on; R50: 493 vs 541)
    ways = max(2, min(N, int(os.environ.get('TDN_IMG_SPLIT_WAYS', '2'))))
    cuts = [N * i // ways for i in range(ways + 1)]      # contiguous image ranges, one chain each
    pool = _split_streams.get(key)
    if pool is None or len(pool) < ways:
        pool = [torch.cuda.Stream(device=dev) for _ in range(ways)]
        _split_streams[key] = pool
    # batch tensors of every block, allocated on the current stream
    bufs, x = [], cur
    for b in blocks:
        H, W = x.shape[1], x.shape[2]

        def new(hh, ww, c):
            return torch.empty(N, hh, ww, c, dtype=x.dtype, device=dev)
        if b.kind == 'bottleneck':
            h1 = new(ops.conv_out_size(H, b.u1.k, b.u1.stride, b.u1.pad),
                     ops.conv_out_size(W, b.u1.k, b.u1.stride, b.u1.pad), b.u1.Cout)
            h2 = new(ops.conv_out_size(h1.shape[1], b.u2.k, b.u2.stride, b.u2.pad),
                     ops.conv_out_size(h1.shape[2], b.u2.k, b.u2.stride, b.u2.pad), b.u2.Cout)
            out = new(h2.shape[1], h2.shape[2], b.u3.Cout)
        else:
            h1 = new(ops.conv_out_size(H, b.u1.k, b.u1.stride, b.u1.pad),
                     ops.conv_out_size(W, b.u1.k, b.u1.stride, b.u1.pad), b.u1.Cout)
            h2 = None
            out = new(h1.shape[1], h1.shape[2], b.u2.Cout)
        res = None
        if b.ud is not None and not (_block_head_fusable(b, x.shape[3]) and os.environ.get('TDN_BLOCK_HEAD', '2') != '1'):
            res = new(out.shape[1], out.shape[2], out.shape[3])     # (a head block computes the branch in its launch)
        bits = None
        if _block_bits_on() and _WANT_BWD[0] and _block_fusable(b, x.shape[3]):
            bits = ops.bottleneck_bit_planes(N, H, W, b.u1.Cout, dev)
            h1._tdn_bits = bits
        elif _block_bits_on() and _WANT_BWD[0] and _block_head_fusable(b, x.shape[3]):
            bits = ops.bottleneck_bit_planes(N, H, W, b.u1.Cout, dev)[:2]
            h1._tdn_bits = bits
        bufs.append((h1, h2, out, res, bits))
        x = out
    ev = streams.record(torch.cuda.current_stream(dev))
    for i in range(ways):
        streams.wait(pool[i], ev)
    # Launch order: block by block, alternating between the chains — the order in which the eager path and the
    # launch-plan executor hand the launches to the GPU.  (A captured hipGraph is replayed by the runtime branch by
    # branch whatever the capture order: the second chain starts as soon as the host has submitted the first chain's
    # nodes — a few hundred microseconds unprofiled, ~0.7 ms under rocprofv3, which is what its timelines show.)
    xs = [cur[cuts[i]:cuts[i + 1]] for i in range(ways)]
    # TDN_CHAIN_SYNC=n (diagnostic, default off): every n blocks each chain waits for the other chains' progress up to
    # that block (a per-block cross-join, meant to make a replayed graph interleave the chains).  Mode 'cross' (mutual
    # waits between the two forked streams) is the pattern that crashed hipStreamEndCapture in round 2; inside a
    # GraphedStep capture streams.wait now refuses it with a RuntimeError (the step then runs eager) — to reproduce the
    # crash itself, capture with a plain torch.cuda.graph (not policed).
    chain_sync = int(os.environ.get('TDN_CHAIN_SYNC', '0'))
    if pre is not None:
        for i in range(ways):
            prev = _lib.set_stream_override(pool[i].cuda_stream)
            try:
                pre(cuts[i], cuts[i + 1])
            finally:
                _lib.set_stream_override(prev)
    for bi_, (b, (h1, h2, out, res, bits)) in enumerate(zip(blocks, bufs)):
        if chain_sync > 0 and bi_ > 0 and bi_ % chain_sync == 0:
            mode = os.environ.get('TDN_CHAIN_SYNC_MODE', 'cross')
            if mode == 'cross':        # every chain waits for every other chain's event of this point
                toks = [streams.record(pool[i]) for i in range(ways)]
                for i in range(ways):
                    for j in range(ways):
                        if i != j:
                            streams.wait(pool[i], toks[j])
            elif mode == 'oneway':     # chain i waits for chain i - 1 only
                toks = [streams.record(pool[i]) for i in range(ways)]
                for i in range(1, ways):
                    streams.wait(pool[i], toks[i - 1])
            else:                      # 'main': join into the main stream and fork again
                main_ = torch.cuda.current_stream(dev)
                for i in range(ways):
                    streams.wait_stream(main_, pool[i])
                ev_ = streams.record(main_)
                for i in range(ways):
                    streams.wait(pool[i], ev_)
        for i in range(ways):
            a, e_ = cuts[i], cuts[i + 1]
            prev = _lib.set_stream_override(pool[i].cuda_stream)
            try:
                xs[i], _ = _block_fwd(xs[i], b, (h1[a:e_], h2[a:e_] if h2 is not None else None, out[a:e_],
                                                 res[a:e_] if res is not None else None,
                                                 tuple(t[a:e_] for t in bits) if bits is not None else None))
            finally:
                _lib.set_stream_override(prev)
    main = torch.cuda.current_stream(dev)
    for i in range(ways):
        streams.wait_stream(main, pool[i])
    saved, x = [], cur
    for (h1, h2, out, res, bits) in bufs:
        saved.append((x, h1, h2, out))
        x = out
    return x, saved


def _block_dgrad_fused(b, saved, g, mask_src, outs=None, bits=None):
    """The three input gradients of a fusable block in one launch (g2, g1, dx); with DEBUG_BWD set the launch is
    recorded as the three dgrad launches it replaces — their operands and results all exist in HBM — so the in-situ
    parity checks (tests/parity_util.py) recompute every conv of the fused launch like any other."""
    x, h1, h2, out = saved
    u1, u2, u3 = b.u1, b.u2, b.u3
    if bits is None:
        bits = getattr(h1, '_tdn_bits', None)
    if bits is not None and mask_src is not None and mask_src.data_ptr() != x.data_ptr():
        bits = None                    # the third plane is x > 0: only valid when the block's input is the mask source
    if bits is not None and mask_src is None:
        bits = None                    # no mask on dx at all (first block of a net without a stem): 16-bit path
    g2, g1, dx = ops.bottleneck_dgrad(g, u3.w_dgrad, u2.w_dgrad, u1.w_dgrad, (h2, h1, mask_src), outs=outs, bits=bits)
    if DEBUG_BWD is not None:
        DEBUG_BWD.append(('dgrad', u3, g, _hw(h2), None, ADD_NONE, h2, g2))
        DEBUG_BWD.append(('dgrad', u2, g2, _hw(h1), None, ADD_NONE, h1, g1))
        DEBUG_BWD.append(('dgrad', u1, g1, _hw(x), g, ADD_SAME, mask_src, dx))
    return g2, g1, dx


def _head_ds_in_launch():
    """TDN_BLOCK_HEAD=2 (default): the head block's downsample branch is computed inside its launch, forward and
    backward.  The in-situ parity recorder (DEBUG_BWD) wants the downsample dgrad's result as a tensor: separate launch."""
    return os.environ.get('TDN_BLOCK_HEAD', '2') not in ('1', '2f') and DEBUG_BWD is None


def _block_head_dgrad_fused(b, saved, g, outs=None, bits=None, t_out=None):
    """Input gradients of a head block (_block_head_fusable): the downsample conv's dgrad as a launch of its own, then
    g2, g1 and dx = conv1^T(g1) + t in one launch.  Returns (g2, g1, dx, t)."""
    x, h1, h2, out = saved
    u1, u2, u3 = b.u1, b.u2, b.u3
    if bits is None:
        bits = getattr(h1, '_tdn_bits', None)
    if _head_ds_in_launch():
        # downsample^T(g) accumulated inside the launch, beside conv3^T(g): g is read once, t never exists in HBM
        t, down = None, b.ud.w_dgrad
    else:
        t, down = unit_dgrad(b.ud, g, _hw(x), out=t_out), None
    g2, g1, dx = ops.bottleneck_head_dgrad(g, u3.w_dgrad, u2.w_dgrad, u1.w_dgrad, (h2, h1), t, outs=outs,
                                           bits=bits[:2] if bits is not None else None, down=down)
    if DEBUG_BWD is not None:
        DEBUG_BWD.append(('dgrad', u3, g, _hw(h2), None, ADD_NONE, h2, g2))
        DEBUG_BWD.append(('dgrad', u2, g2, _hw(h1), None, ADD_NONE, h1, g1))
        DEBUG_BWD.append(('dgrad', u1, g1, _hw(x), t, ADD_SAME, None, dx))
    return g2, g1, dx, t


def _block_bwd(b, saved, g, extra, mask_src, need_dx, wq=None):
    """g: gradient w.r.t. the block's pre-ReLU output, already masked by (out > 0).
    extra: external gradient w.r.t. the block INPUT to fold in (e.g. the FPN's gradient of a stage output).
    mask_src: if given, the returned dx is masked by (mask_src > 0) — i.e. it already is the masked ``g`` of the
    block that produced this block's input.  Returns (dx | None, {unit: grads})."""
    x, h1, h2, out = saved
    grads = {}
    t, br = None, None
    if need_dx and extra is None and _block_fusable(b, x.shape[3]):
        g2, g1, dx = _block_dgrad_fused(b, saved, g, mask_src)
        grads[b.u3] = unit_wgrad(b.u3, h2, g, queue=wq)
        grads[b.u2] = unit_wgrad(b.u2, h1, g2, queue=wq)
        grads[b.u1] = unit_wgrad(b.u1, x, g1, queue=wq)
        return dx, grads
    if need_dx and extra is None and mask_src is None and _block_head_fusable(b, x.shape[3]):
        g2, g1, dx, _ = _block_head_dgrad_fused(b, saved, g)
        grads[b.u3] = unit_wgrad(b.u3, h2, g, queue=wq)
        grads[b.u2] = unit_wgrad(b.u2, h1, g2, queue=wq)
        grads[b.u1] = unit_wgrad(b.u1, x, g1, queue=wq)
        grads[b.ud] = unit_wgrad(b.ud, x, g, queue=wq)
        return dx, grads
    if need_dx and b.ud is not None:
        with branch(g.device, b.ud, (g, extra) if extra is not None else (g,)) as br:
            t = unit_dgrad(b.ud, g, _hw(x), extra, ADD_SAME)   # beside the conv3 -> conv2 dgrads of the main path
    if b.kind == 'bottleneck':
        grads[b.u3] = unit_wgrad(b.u3, h2, g, queue=wq)
        g2 = unit_dgrad(b.u3, g, _hw(h2), mask_src=h2)
        grads[b.u2] = unit_wgrad(b.u2, h1, g2, queue=wq)
        g1 = unit_dgrad(b.u2, g2, _hw(h1), mask_src=h1)
    else:
        grads[b.u2] = unit_wgrad(b.u2, h1, g, queue=wq)
        g1 = unit_dgrad(b.u2, g, _hw(h1), mask_src=h1)
    grads[b.u1] = unit_wgrad(b.u1, x, g1, queue=wq)
    if b.ud is not None:
        grads[b.ud] = unit_wgrad(b.ud, x, g, queue=wq)
    dx = None
    if need_dx:
        if b.ud is not None:
            br.join()
        elif extra is not None:
            t = ops.add_relu_mask(g, extra, None)
        else:
            t = g
        dx = unit_dgrad(b.u1, g1, _hw(x), t, ADD_SAME, mask_src)
    return dx, grads


def _block_bwd_chains(b, saved, g, extra, mask_src, wq, pool, cuts):
    """_block_bwd with the dgrad launches of every image range issued on that range's stream (see
    SeqNetFunction._backward): outputs are slices of batch tensors allocated here, weight gradients are queued on the
    batch tensors as usual.  Needs need_dx, and `extra` only where the block has a downsample conv."""
    x, h1, h2, out = saved
    grads = {}
    dev = g.device

    def new_like(t):
        return torch.empty_like(t)

    bott = b.kind == 'bottleneck'
    g2 = new_like(h2) if bott else None
    g1 = new_like(h1)
    dx = new_like(x)
    fused = extra is None and _block_fusable(b, x.shape[3])
    head = extra is None and mask_src is None and _block_head_fusable(b, x.shape[3])
    t = g
    if b.ud is not None:
        t = new_like(x) if not (head and _head_ds_in_launch()) else None
    bits_all = getattr(h1, '_tdn_bits', None) if (fused or head) else None
    for i, st in enumerate(pool[:len(cuts) - 1]):
        a, e = cuts[i], cuts[i + 1]
        prev = _lib.set_stream_override(st.cuda_stream)
        try:
            gi = g[a:e]
            if fused:
                _block_dgrad_fused(b, (x[a:e], h1[a:e], h2[a:e], None), gi,
                                   mask_src[a:e] if mask_src is not None else None,
                                   outs=(g2[a:e], g1[a:e], dx[a:e]),
                                   bits=tuple(t[a:e] for t in bits_all) if bits_all is not None else None)
                continue
            if head:
                _block_head_dgrad_fused(b, (x[a:e], h1[a:e], h2[a:e], None), gi, outs=(g2[a:e], g1[a:e], dx[a:e]),
                                        bits=tuple(t_[a:e] for t_ in bits_all) if bits_all is not None else None,
                                        t_out=t[a:e] if t is not None else None)
                continue
            if b.ud is not None:
                unit_dgrad(b.ud, gi, _hw(x), extra[a:e] if extra is not None else None, ADD_SAME, out=t[a:e])
            if bott:
                unit_dgrad(b.u3, gi, _hw(h2), mask_src=h2[a:e], out=g2[a:e])
                unit_dgrad(b.u2, g2[a:e], _hw(h1), mask_src=h1[a:e], out=g1[a:e])
            else:
                unit_dgrad(b.u2, gi, _hw(h1), mask_src=h1[a:e], out=g1[a:e])
            unit_dgrad(b.u1, g1[a:e], _hw(x), t[a:e], ADD_SAME, mask_src[a:e] if mask_src is not None else None,
                       out=dx[a:e])
        finally:
            _lib.set_stream_override(prev)
    if bott:
        grads[b.u3] = unit_wgrad(b.u3, h2, g, queue=wq)
        grads[b.u2] = unit_wgrad(b.u2, h1, g2, queue=wq)
    else:
        grads[b.u2] = unit_wgrad(b.u2, h1, g, queue=wq)
    grads[b.u1] = unit_wgrad(b.u1, x, g1, queue=wq)
    if b.ud is not None:
        grads[b.ud] = unit_wgrad(b.ud, x, g, queue=wq)
    # nothing of this block may be recycled before the chains have run: t is only referenced here
    key = (dev.index, torch._C._cuda_getCurrentRawStream(dev.index))
    _side_refs.setdefault(key, []).extend(v for v in (t, g, g1, g2, dx) if v is not None)
    return dx, grads


def _stem_fwd(u, xp, hw):
    """conv7x7/s2 + norm + ReLU of the stem (resnet.py:254-257): BN folded into the conv epilogue, or GroupNorm."""
    if not (u.gn or u.bnt):
        return ops.stem_conv_fwd(xp, u.w_fwd, hw, u.scale, u.shift, True)
    z = ops.stem_conv_fwd(xp, u.w_fwd, hw, None, None, False)
    s, stats = _dyn_norm_fwd(u, z, None, True)
    _gn_store(u)[u] = (z, stats)
    return s


def _stem_pool_split(u, xp, hw):
    """The one-launch stem + pool per image range, at the head of the per-image forward chains (TDN_STEM_SPLIT, default
    1): returns (pooled output, indices, pre) with the batch tensors allocated here and ``pre(a, e)`` the launch that
    fills images a..e — or None where the one-launch stem does not apply."""
    if (u.gn or u.bnt or u.Cout != 64 or u.scale is None or u.shift is None or DEBUG_CAPTURE is not None or
            os.environ.get('TDN_STEM_FUSED', '1') == '0' or os.environ.get('TDN_STEM_SPLIT', '1') == '0'):
        return None
    H, W = hw
    N = xp.shape[0]
    Ho, Wo = ops.conv_out_size(H // 2, 3, 2, 1), ops.conv_out_size(W // 2, 3, 2, 1)
    y = torch.empty(N, Ho, Wo, u.Cout, dtype=xp.dtype, device=xp.device)
    idx = torch.empty(N, Ho, Wo, u.Cout, dtype=torch.uint8, device=xp.device)

    def pre(a, e):
        ops.stem_pool_fwd(xp[a:e], u.w_fwd, hw, u.scale, u.shift, out=(y[a:e], idx[a:e]))
    return y, idx, pre


def _stem_pool_fwd(u, xp, hw):
    """Stem + max pool (resnet.py:254-258): (stem activation or None, pooled output, window indices).  The plain stem —
    64 channels, eval-mode BN folded into the conv — runs as ONE launch that never writes its full-size activation
    (ops.stem_pool_fwd, bit-identical to the two launches); TDN_STEM_FUSED=0, a parity test's DEBUG_CAPTURE (it looks
    at the stem activation) and every other stem variant take the two launches."""
    if (not (u.gn or u.bnt) and u.Cout == 64 and u.scale is not None and u.shift is not None and
            DEBUG_CAPTURE is None and os.environ.get('TDN_STEM_FUSED', '1') != '0'):
        y, idx = ops.stem_pool_fwd(xp, u.w_fwd, hw, u.scale, u.shift)
        return None, y, idx
    s = _stem_fwd(u, xp, hw)
    y, idx = ops.maxpool3x3s2_fwd(s)
    return s, y, idx


# Called by SeqNetFunction._backward behind every weight-gradient group launch of a stage, with the number of stages
# whose groups have been launched so far in this pass (1: the last stage's, ...) and the number of stages:
# dp.GradReducer hangs the launch of its deferred bucket all-reduces on it.
FLUSH_HOOKS = []


class SeqNetFunction(torch.autograd.Function):
    """ResNet.forward (resnet.py:253-268) — or a single residual block — as one autograd node."""

    @staticmethod
    def forward(ctx, net, x, *params):
        ctx.gn_saved = {}
        prev = _WANT_BWD[0]
        _WANT_BWD[0] = any(ctx.needs_input_grad)
        try:
            with gn_scope(ctx.gn_saved):
                return SeqNetFunction._forward(ctx, net, x, *params)
        finally:
            _WANT_BWD[0] = prev

    @staticmethod
    def _forward(ctx, net, x, *params):
        st = {}
        s = None
        stem_pre = None
        if net.stem is not None and isinstance(x, StagedImages):
            if x.dtype != net.dtype:
                raise RuntimeError('staged images are %s but the net computes in %s' % (x.dtype, net.dtype))
            xp, (H, W) = x.xp, x.hw
            s, cur, idx, stem_pre = SeqNetFunction._stem(net, xp, (H, W))
            st.update(xp=xp, y=cur, idx=idx, img_hw=(H, W))   # the stem's own output is not kept: see maxpool3x3s2_bwd
        elif net.stem is not None:
            if x.dim() != 4 or x.shape[1] != 3:
                raise RuntimeError('ResNet expects an (N,3,H,W) image batch, got %s' % (tuple(x.shape),))
            img = x if x.dtype == torch.float32 else x.float()
            H, W = img.shape[2], img.shape[3]
            xp = ops.stage_image(img, net.dtype)
            s, cur, idx, stem_pre = SeqNetFunction._stem(net, xp, (H, W))
            st.update(xp=xp, y=cur, idx=idx, img_hw=(H, W))   # the stem's own output is not kept: see maxpool3x3s2_bwd
        else:
            cur = ops.to_nhwc_bf16(x, net.dtype)
        saved, outs = [], []
        nb = len(net.blocks)
        bi = 0
        split_m = _img_split_m()
        while bi < nb:
            b = net.blocks[bi]
            # pixels of this block's output batch (stride on conv2 / conv1: the block works at the reduced size)
            m_out = cur.shape[0] * -(-cur.shape[1] // b.stride) * -(-cur.shape[2] // b.stride)
            if split_m > 0 and cur.shape[0] >= 2 and m_out <= split_m and \
                    all(_splittable(bb) for bb in net.blocks[bi:]):
                join_branches(cur.device)
                cur, svs = _blocks_fwd_split(net.blocks[bi:], cur, pre=stem_pre)
                stem_pre = None
                for j, sv in enumerate(svs):
                    saved.append(sv)
                    if bi + j in net.out_blocks:
                        outs.append(sv[3])
                bi = nb
                break
            if stem_pre is not None:       # the chains do not start here after all: the whole batch's stem now
                stem_pre(0, cur.shape[0])
                stem_pre = None
            cur, sv = _block_fwd(cur, b)
            saved.append(sv)
            if bi in net.out_blocks:
                outs.append(cur)
            bi += 1
        if stem_pre is not None:
            stem_pre(0, cur.shape[0])
        if not net.blocks:
            outs.append(cur)
        join_branches(cur.device)
        ctx.net, ctx.st, ctx.saved, ctx.dev = net, st, saved, cur.device
        if DEBUG_CAPTURE is not None:      # parity tests look at the stem's activation too; the step itself drops it
            DEBUG_CAPTURE['seq'] = (dict(st, s=s), saved)
        return tuple(_as_nchw(o) for o in outs)

    @staticmethod
    def _stem(net, xp, hw):
        """(stem activation | None, pooled output, window indices, pre | None): with ``pre`` the pooled output is still
        to be produced — per image range, at the head of the per-image chains (_blocks_fwd_split) — which is only done
        when those chains start at the first block."""
        N = xp.shape[0]
        H, W = hw
        will_split = (_img_split_m() > 0 and N >= 2 and net.blocks and
                      N * (H // 4) * (W // 4) // (net.blocks[0].stride ** 2) <= _img_split_m() and
                      all(_splittable(bb) for bb in net.blocks))
        if will_split:
            r = _stem_pool_split(net.stem, xp, hw)
            if r is not None:
                return None, r[0], r[1], r[2]
        s, cur, idx = _stem_pool_fwd(net.stem, xp, hw)
        return s, cur, idx, None

    @staticmethod
    def backward(ctx, *douts):
        with gn_scope(ctx.gn_saved):
            return SeqNetFunction._backward(ctx, *douts)

    @staticmethod
    def _backward(ctx, *douts):
        net, st, saved = ctx.net, ctx.st, ctx.saved
        ext = {}
        for bi, d in zip(net.out_blocks, douts):
            if d is not None:
                ext[bi] = ops.to_nhwc_bf16(d, net.dtype)
        unit_grads = {}
        need_net_dx = ctx.needs_input_grad[1] and net.stem is None
        g = None
        wq = WgradQueue(ctx.dev)
        flush_every = int(os.environ.get('TDN_WGRAD_FLUSH', '0'))
        since_flush = 0
        nflush = 0
        nstages = sum(1 for b_ in net.blocks if b_.ud is not None)
        # Per-image dgrad chains (default; TDN_BWD_SPLIT=0 turns them off): like the forward's, the dgrad launches of
        # each image range go to that range's stream, block by block in alternation — the stretches of the backward
        # pass where a chain of small dgrad kernels had the GPU to itself become two half-size chains side by side
        # (443 -> 459 img/s).  The weight-gradient groups still see batch tensors, so the main stream joins the chains
        # before every group is launched.
        chains = None
        nimg = saved[0][0].shape[0] if saved else 0
        if os.environ.get('TDN_BWD_SPLIT', '1') != '0' and nimg >= 2 and (net.stem is not None or need_net_dx) and \
                all(_splittable(b_) for b_ in net.blocks) and \
                all(net.blocks[k + 1].ud is not None for k in ext if k + 1 < len(net.blocks)):
            dev = ctx.dev
            key = (dev.index, torch._C._cuda_getCurrentRawStream(dev.index))
            ways = max(2, min(nimg, int(os.environ.get('TDN_IMG_SPLIT_WAYS', '2'))))
            pool = _split_streams.get(key)
            if pool is None or len(pool) < ways:
                pool = [torch.cuda.Stream(device=dev) for _ in range(ways)]
                _split_streams[key] = pool
            chains = (pool[:ways], [nimg * i // ways for i in range(ways + 1)])
        main = torch.cuda.current_stream(ctx.dev)

        def join_chains():
            for st_ in chains[0]:
                streams.wait_stream(main, st_)

        started = False
        if chains is not None:
            # a flush from inside a block (TDN_WGRAD_GROUP=0 flushes per layer) must see the chains' g tensors too
            wq.pre_flush = lambda: join_chains() if started else None
        for bi in reversed(range(len(net.blocks))):
            b, sv = net.blocks[bi], saved[bi]
            if g is None:
                e = ext.get(bi)
                if e is None:
                    continue  # nothing flows into this block's output
                g = ops.add_relu_mask(e, None, sv[3])
            extra = ext.get(bi - 1) if bi > 0 else None
            mask_src = saved[bi - 1][3] if bi > 0 else None
            need_dx = bi > 0 or net.stem is not None or need_net_dx
            if chains is not None and need_dx:
                if not started:        # the chains start behind everything the main stream has produced so far
                    ev0 = streams.record(main)
                    for st_ in chains[0]:
                        streams.wait(st_, ev0)
                    started = True
                g, gr = _block_bwd_chains(b, sv, g, extra, mask_src, wq, chains[0], chains[1])
            else:
                if chains is not None and started:
                    join_chains()      # this block runs on the main stream and reads what the chains produced
                g, gr = _block_bwd(b, sv, g, extra, mask_src, need_dx, wq)
            unit_grads.update(gr)
            since_flush += 1
            # Weight gradients are launched as groups: at the first block of a stage (resnet.py:130-136; the last one
            # the backward pass reaches) and, inside long stages, every TDN_WGRAD_FLUSH blocks.  One group per stage
            # is the most efficient launch, but it only becomes available when the stage's dgrad chain has ended: the
            # timeline then alternates between stretches where a chain of small dgrad kernels has the GPU to itself
            # (layer3: ~300 us) and bursts of weight-gradient work, and ends in a tail of weight gradients with nothing
            # left beside them.  Half-stage groups keep both kinds of work on the GPU throughout.
            if b.ud is not None or (flush_every > 0 and since_flush >= flush_every):
                if chains is not None and started:
                    join_chains()
                wq.flush()
                since_flush = 0
                nflush += 1
                for hook in FLUSH_HOOKS:
                    hook(nflush, nstages)
        if chains is not None and started:
            join_chains()
        dx_in = None
        if net.stem is not None:
            if g is not None:
                H, W = st['img_hw']
                ds = ops.maxpool3x3s2_bwd(g, st['idx'], (H // 2, W // 2), pooled=st['y'])
                unit_grads[net.stem] = unit_wgrad(net.stem, st['xp'], ds, (H, W), queue=wq)
            # the image itself gets no gradient (the reference never needs one; SURVEY §8(d): -5.06 GFLOP)
        elif need_net_dx and g is not None:
            dx_in = _as_nchw(g)
        wq.flush()
        flat = []
        for u in net.units():
            flat += unit_grads.get(u, [None] * len(u.params()))
        join_side_stream(ctx.dev)
        return (None, dx_in) + tuple(flat)


# ---------------------------------------------------------------------------------------------------
# FPN
# ---------------------------------------------------------------------------------------------------
class FPNNet(object):
    def __init__(self, lat, fpn, start_level, backbone_end_level, num_outs, add_extra_convs, num_ins):
        self.lat, self.fpn = lat, fpn
        self.start_level, self.backbone_end_level = start_level, backbone_end_level
        self.num_outs, self.add_extra_convs, self.num_ins = num_outs, add_extra_convs, num_ins
        self.dtype = lat[0].dtype

    def units(self):
        return list(self.lat) + list(self.fpn)

    def params(self):
        ps = []
        for u in self.units():
            ps += u.params()
        return ps


class FPNFunction(torch.autograd.Function):
    """FPN.forward (fpn.py:88-125): laterals, top-down nearest-2x add (fused into the lateral epilogue),
    3x3 output convs, extra levels by stride-2 subsampling (fpn.py:114-116) or stride-2 convs (fpn.py:118-124)."""

    @staticmethod
    def forward(ctx, net, *args):
        ctx.gn_saved = {}
        with gn_scope(ctx.gn_saved):
            return FPNFunction._forward(ctx, net, *args)

    @staticmethod
    def _forward(ctx, net, *args):
        inputs = args[:net.num_ins]
        nlat = len(net.lat)
        xs = [ops.to_nhwc_bf16(inputs[i + net.start_level], net.dtype) for i in range(nlat)]
        lat, outs = [None] * nlat, [None] * nlat
        for i in reversed(range(nlat)):
            if i == nlat - 1:
                lat[i] = unit_fwd(net.lat[i], xs[i])
            else:
                lat[i] = unit_fwd(net.lat[i], xs[i], lat[i + 1], ADD_UP2X)
            if i > 0:
                # the coarse levels' output convs (small launches) run beside the rest of the top-down chain
                with branch(lat[i].device, net.fpn[i], (lat[i],)):
                    outs[i] = unit_fwd(net.fpn[i], lat[i])
            else:
                outs[i] = unit_fwd(net.fpn[i], lat[i])
        join_branches(lat[0].device)
        extra_in = []   # inputs of the extra stride-2 convs (RetinaNet style), for backward
        if net.num_outs > nlat:
            if not net.add_extra_convs:
                for _ in range(net.num_outs - nlat):
                    outs.append(ops.subsample2_fwd(outs[-1]))
            else:
                orig = ops.to_nhwc_bf16(inputs[net.backbone_end_level - 1], net.dtype)
                outs.append(unit_fwd(net.fpn[nlat], orig))
                extra_in.append(orig)
                for i in range(nlat + 1, net.num_outs):
                    # F.relu(outs[-1], inplace=True) in the reference also rewrites the returned level (fpn.py:124)
                    outs[-1] = ops.add_relu_mask(outs[-1], None, outs[-1])
                    outs.append(unit_fwd(net.fpn[i], outs[-1]))
                    extra_in.append(outs[-2])
        ctx.net, ctx.xs, ctx.lat, ctx.extra_in = net, xs, lat, extra_in
        if DEBUG_CAPTURE is not None:
            DEBUG_CAPTURE['fpn'] = (xs, lat)
        ctx.out_hw = [_hw(o) for o in outs]
        ctx.out_meta = (outs[0].shape[0], outs[0].shape[3], outs[0].device)
        return tuple(_as_nchw(o) for o in outs)

    @staticmethod
    def backward(ctx, *douts):
        with gn_scope(ctx.gn_saved):
            return FPNFunction._backward(ctx, *douts)

    @staticmethod
    def _backward(ctx, *douts):
        net, xs, lat = ctx.net, ctx.xs, ctx.lat
        nlat = len(net.lat)
        N, C, dev = ctx.out_meta
        d = [ops.to_nhwc_bf16(t, net.dtype) if t is not None else None for t in douts]

        def zeros(i):
            h, w = ctx.out_hw[i]
            return torch.zeros(N, h, w, C, dtype=net.dtype, device=dev)

        unit_grads = {}
        dx = [None] * net.num_ins
        wq = WgradQueue(dev)
        if net.num_outs > nlat:
            if not net.add_extra_convs:
                for j in range(net.num_outs - 1, nlat - 1, -1):
                    if d[j] is not None:
                        d[j - 1] = ops.subsample2_bwd(d[j], ctx.out_hw[j - 1], d[j - 1])
            else:
                for j in range(net.num_outs - 1, nlat - 1, -1):
                    gj = d[j] if d[j] is not None else zeros(j)
                    xin = ctx.extra_in[j - nlat]
                    unit_grads[net.fpn[j]] = unit_wgrad(net.fpn[j], xin, gj, queue=wq)
                    if j > nlat:
                        # input was relu(out[j-1]) written back in place: mask by the (ReLU'd) saved tensor
                        d[j - 1] = unit_dgrad(net.fpn[j], gj, _hw(xin), d[j - 1], ADD_SAME, xin)
                    else:
                        k = net.backbone_end_level - 1
                        if ctx.needs_input_grad[1 + k]:
                            dx[k] = unit_dgrad(net.fpn[j], gj, _hw(xin))
        dL = [None] * nlat
        # the output convs' weight gradients need nothing this pass still has to compute (d and the saved laterals):
        # one group, launched before the dgrad chain starts; the laterals' follow as a second group behind it
        gis = [d[i] if d[i] is not None else zeros(i) for i in range(nlat)]
        for i in range(nlat):
            unit_grads[net.fpn[i]] = unit_wgrad(net.fpn[i], lat[i], gis[i], queue=wq)
        wq.flush()
        for i in range(nlat):
            gi = gis[i]
            dL[i] = unit_dgrad(net.fpn[i], gi, _hw(lat[i]), dL[i - 1] if i > 0 else None, ADD_SUMPOOL2)
            # level i's lateral gradients need only dL[i]: they run beside the remaining (coarser, smaller) output
            # convs' dgrad chain; the coarsest one, whose result the backbone's backward starts from, stays inline
            unit_grads[net.lat[i]] = unit_wgrad(net.lat[i], xs[i], dL[i], queue=wq)
            k = i + net.start_level
            if ctx.needs_input_grad[1 + k]:
                if i < nlat - 1:
                    with branch(dev, net.lat[i], (dL[i], dx[k]) if dx[k] is not None else (dL[i],)):
                        t = unit_dgrad(net.lat[i], dL[i], _hw(xs[i]), dx[k], ADD_SAME)
                else:
                    t = unit_dgrad(net.lat[i], dL[i], _hw(xs[i]), dx[k], ADD_SAME)
                dx[k] = t
        wq.flush()
        flat = []
        for u in net.units():
            flat += unit_grads.get(u, [None] * len(u.params()))
        join_side_stream(dev)
        return (None,) + tuple(_as_nchw(t) if t is not None else None for t in dx) + tuple(flat)


# ---------------------------------------------------------------------------------------------------
# PAFPN bottom-up path (pafpn.py:127-131) + stride-2 extra levels (pafpn.py:136-138)
# ---------------------------------------------------------------------------------------------------
class PAPathNet(object):
    def __init__(self, pa1, pa2, num_extra):
        self.pa1, self.pa2, self.num_extra = pa1, pa2, num_extra
        self.dtype = pa1[0].dtype

    def units(self):
        us = []
        for a, b in zip(self.pa1, self.pa2):
            us += [a, b]
        return us

    def params(self):
        ps = []
        for u in self.units():
            ps += u.params()
        return ps


class PAPathFunction(torch.autograd.Function):
    """N_0 = P_0;  N_i = pa_convs2[i-1](P_i + pa_convs1[i-1](N_{i-1}))  (pafpn.py:129-131), then
    ``num_extra`` levels by stride-2 subsampling of the last one.  Without an activation the ``P_i +`` is fused
    into the stride-2 conv's epilogue; with ReLU (which acts before the add) it is a separate add kernel."""

    @staticmethod
    def forward(ctx, net, *args):
        ctx.gn_saved = {}
        with gn_scope(ctx.gn_saved):
            return PAPathFunction._forward(ctx, net, *args)

    @staticmethod
    def _forward(ctx, net, *args):
        n = len(net.pa1) + 1
        P = [ops.to_nhwc_bf16(t, net.dtype) for t in args[:n]]
        outs, t_saved, s_saved = [P[0]], [], []
        for i in range(1, n):
            u1, u2 = net.pa1[i - 1], net.pa2[i - 1]
            if u1.relu:
                t = unit_fwd(u1, outs[i - 1])
                s_in = ops.add_relu_mask(t, P[i], None)
            else:
                t = None
                s_in = unit_fwd(u1, outs[i - 1], P[i], ADD_SAME)
            t_saved.append(t)
            s_saved.append(s_in)
            outs.append(unit_fwd(u2, s_in))
        for _ in range(net.num_extra):
            outs.append(ops.subsample2_fwd(outs[-1]))
        ctx.net, ctx.outs, ctx.t_saved, ctx.s_saved, ctx.n = net, outs, t_saved, s_saved, n
        return tuple(_as_nchw(o) for o in outs)

    @staticmethod
    def backward(ctx, *douts):
        with gn_scope(ctx.gn_saved):
            return PAPathFunction._backward(ctx, *douts)

    @staticmethod
    def _backward(ctx, *douts):
        net, outs, n = ctx.net, ctx.outs, ctx.n
        dev = outs[0].device
        d = [ops.to_nhwc_bf16(t, net.dtype) if t is not None else None for t in douts]
        for j in range(len(outs) - 1, n - 1, -1):       # fold the subsampled extra levels back
            if d[j] is not None:
                d[j - 1] = ops.subsample2_bwd(d[j], _hw(outs[j - 1]), d[j - 1])
        unit_grads = {}
        dP = [None] * n
        wq = WgradQueue(dev)

        def masked(g_, y_, u_):   # backward of the unit's activation from its saved output: ReLU, or ReLU6 (0 < y < 6)
            return ops.act_mask(g_, y_, 6.0) if u_.act6 else ops.add_relu_mask(g_, None, y_)

        for i in range(n - 1, 0, -1):
            u1, u2 = net.pa1[i - 1], net.pa2[i - 1]
            g = d[i] if d[i] is not None else torch.zeros_like(outs[i])
            if u2.relu:
                g = masked(g, outs[i], u2)
            unit_grads[u2] = unit_wgrad(u2, ctx.s_saved[i - 1], g, queue=wq)
            ds = unit_dgrad(u2, g, _hw(ctx.s_saved[i - 1]))
            dP[i] = ds
            dt = masked(ds, ctx.t_saved[i - 1], u1) if u1.relu else ds
            unit_grads[u1] = unit_wgrad(u1, outs[i - 1], dt, queue=wq)
            d[i - 1] = unit_dgrad(u1, dt, _hw(outs[i - 1]), d[i - 1], ADD_SAME)
        dP[0] = d[0]
        wq.flush()
        flat = []
        for u in net.units():
            flat += unit_grads.get(u, [None] * len(u.params()))
        join_side_stream(dev)
        return (None,) + tuple(_as_nchw(t) if t is not None else None for t in dP) + tuple(flat)
