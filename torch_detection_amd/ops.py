"""Host-side operator layer: torch tensors in, C-ABI calls out (libtdn.so, include/tdn.h).

Activations are NHWC bfloat16 (default) or float16 tensors of shape (N, H, W, C), contiguous; the element type of
a call is the one of its activation operand and every other 16-bit operand must match it.  Every function validates
shapes on the host before launching (a mis-shaped launch can fault the GPU), enqueues on PyTorch's
current HIP stream, and never synchronises.  There is no non-HIP fallback.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ADD_NONE, ADD_SAME, ADD_SUMPOOL2, ADD_UP2X, TDN_BF16, TDN_F16, Epilogue  # noqa: F401

BF16 = torch.bfloat16
F16 = torch.float16
_CODES = {BF16: TDN_BF16, F16: TDN_F16}


def dtype_code(dtype):
    """tdn dtype code (include/tdn.h) of a torch 16-bit float dtype."""
    try:
        return _CODES[dtype]
    except KeyError:
        raise ValueError("compute dtype must be torch.bfloat16 or torch.float16, got %s" % (dtype,))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _chk_dev(t, name):
    """Launches go to the CURRENT device's stream (``_lib.stream_ptr``): an operand living on another GPU would be
    touched from the wrong stream / device context — refuse instead of computing garbage."""
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError("%s is on %s but the current device is cuda:%d: select the tensor's device first "
                           "(torch.cuda.set_device / with torch.cuda.device(...))" %
                           (name, t.device, torch.cuda.current_device()))


def _chk_act(t, name, C=None, dtype=None):
    if t.dtype not in _CODES or not t.is_cuda or t.dim() != 4 or not t.is_contiguous():
        raise ValueError("%s must be a contiguous CUDA bfloat16/float16 NHWC tensor, got %s %s %s" %
                         (name, t.dtype, t.device, tuple(t.shape)))
    _chk_dev(t, name)
    if dtype is not None and t.dtype != dtype:
        raise ValueError("%s is %s but the call computes in %s" % (name, t.dtype, dtype))
    if C is not None and t.shape[3] != C:
        raise ValueError("%s: expected %d channels, got %d" % (name, C, t.shape[3]))


def _chk_vec(t, name, C):
    if t is None:
        return
    if t.dtype != torch.float32 or not t.is_cuda or t.numel() != C or not t.is_contiguous():
        raise ValueError("%s must be a contiguous CUDA float32 vector of %d elements" % (name, C))


def conv_out_size(h, k, stride, pad):
    """Output size of the "same" convs of the hot path: 1x1 with pad 0, 3x3 with pad = dilation (conv3x3_group builds
    them with padding = dilation, layers.py:20-32), 3x3/s2 max pool."""
    d = pad if k == 3 and pad > 1 else 1
    return (h + 2 * pad - (d * (k - 1) + 1)) // stride + 1


def _relu_code(relu):
    """0 none / 1 ReLU / 2 ReLU6 from a bool or one of those codes (True is 1)."""
    return 2 if (relu == 2 and relu is not True) else (1 if relu else 0)


_splitk_ws = {}
SPLITK_WS_BYTES = 96 << 20     # tickets + fp32 partial tiles of the largest split launch (R50/R101-FPN: <= 40 MB)


def splitk_workspace(device):
    """Scratch for cross-workgroup split-K launches on the routed stream: one persistent buffer per (device, stream),
    ticket area zeroed once (the kernels leave it zero).  Allocated outside any graph capture / private pool on first
    use — warm-up runs see to that."""
    key = (device.index, _lib.stream_ptr())
    buf = _splitk_ws.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("split-K scratch for this stream does not exist yet: run the step once outside the "
                               "capture (warm-up) so that it is allocated and zeroed there")
        buf = torch.empty(SPLITK_WS_BYTES, dtype=torch.uint8, device=device)
        buf[:_lib.SPLITK_TICKET_BYTES].zero_()
        # the zeroing ran on torch's current stream, the launches that use the tickets go to the ROUTED stream (a
        # chain / branch / side stream): order them once, here, instead of on every launch
        torch.cuda.current_stream(device).synchronize()
        _splitk_ws[key] = buf
    return buf


def make_epilogue(Cout, Ho, Wo, scale=None, shift=None, addend=None, addend_mode=ADD_NONE, relu=False,
                  mask_src=None, N=None, out_f32=False, dtype=None, device=None):
    _chk_vec(scale, "scale", Cout)
    _chk_vec(shift, "shift", Cout)
    ep = Epilogue()
    if device is not None and os.environ.get("TDN_SPLITK", "0") not in ("", "0"):
        # cross-workgroup split-K is off by default (csrc/conv_igemm.hip: splitk_for): no scratch is handed over,
        # and none is allocated, unless it has been switched on
        ws = splitk_workspace(device)
        ep.splitk_ws, ep.splitk_ws_bytes = ws.data_ptr(), ws.numel()
    ep.scale = scale.data_ptr() if scale is not None else None
    ep.shift = shift.data_ptr() if shift is not None else None
    ep.relu = _relu_code(relu)
    ep.out_f32 = 1 if out_f32 else 0
    ep.addend_mode = ADD_NONE
    if addend is not None and addend_mode != ADD_NONE:
        _chk_act(addend, "addend", Cout, dtype)
        exp = {ADD_SAME: (Ho, Wo), ADD_UP2X: (Ho // 2, Wo // 2), ADD_SUMPOOL2: (Ho * 2, Wo * 2)}[addend_mode]
        if tuple(addend.shape[1:3]) != exp or (addend_mode == ADD_UP2X and (Ho % 2 or Wo % 2)) or \
                (N is not None and addend.shape[0] != N):
            raise RuntimeError("epilogue addend of spatial size %s does not match output %s (mode %d)" %
                               (tuple(addend.shape[1:3]), (Ho, Wo), addend_mode))
        ep.addend = addend.data_ptr()
        ep.addend_mode = addend_mode
        ep.addend_h, ep.addend_w = addend.shape[1], addend.shape[2]
    if mask_src is not None:
        _chk_act(mask_src, "mask_src", Cout, dtype)
        if tuple(mask_src.shape[1:3]) != (Ho, Wo) or (N is not None and mask_src.shape[0] != N):
            raise RuntimeError("mask_src shape %s does not match output" % (tuple(mask_src.shape),))
        ep.mask_src = mask_src.data_ptr()
    return ep


def bn_fold(gamma, beta, mean, var, eps, out=None):
    """(scale, shift, invstd) of an eval-mode BatchNorm2d (layers.py:50-54).  ``out``: a (3, C) float32 buffer to
    write into (the three results are its rows) instead of a fresh one."""
    C = gamma.numel()
    for t, n in ((gamma, "gamma"), (beta, "beta"), (mean, "running_mean"), (var, "running_var")):
        _chk_vec(t.detach(), n, C)
    if out is None:
        out = torch.empty(3, C, dtype=torch.float32, device=gamma.device)
    elif tuple(out.shape) != (3, C) or out.dtype != torch.float32 or not out.is_contiguous() or not out.is_cuda:
        raise ValueError("bn_fold: out must be a contiguous CUDA float32 (3, %d) tensor" % C)
    _lib.check(_lib.load().tdn_bn_fold(_ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), float(eps), C,
                                       _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _lib.stream_ptr()), "tdn_bn_fold")
    return out[0], out[1], out[2]


def _pack_out(out, shapes, dtype, name):
    """Persistent pack buffers handed back in by the caller: must be what would have been allocated."""
    for t, shp in zip(out, shapes):
        if t is None or tuple(t.shape) != tuple(shp) or t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("%s: out buffers do not match %s %s" % (name, dtype, [tuple(s) for s in shapes]))
    return out


def pack_conv_weight(w, scale=None, want_dgrad=True, dtype=BF16, out=None):
    """fp32 OIHW (any strides) -> (w_fwd [O][kh][kw][I], w_dgrad [I][kh][kw][O] with scale folded), both `dtype`.
    ``out`` = (w_fwd, w_dgrad) buffers of an earlier call to overwrite in place."""
    w = w.detach()
    if w.dtype != torch.float32 or not w.is_cuda or w.dim() != 4:
        raise ValueError("weight must be a CUDA float32 4-D tensor")
    O, I, kh, kw = w.shape
    _chk_vec(scale, "scale", O)
    if out is not None:
        w_fwd, w_dg = _pack_out(out, [(O, kh, kw, I), (I, kh, kw, O)], dtype, "pack_conv_weight")
    else:
        w_fwd = torch.empty(O, kh, kw, I, dtype=dtype, device=w.device)
        w_dg = torch.empty(I, kh, kw, O, dtype=dtype, device=w.device) if want_dgrad else None
    s = w.stride()
    _lib.check(_lib.load().tdn_pack_conv_weight(_ptr(w), s[0], s[1], s[2], s[3], O, I, kh, kw, _ptr(scale),
                                                _ptr(w_fwd), _ptr(w_dg), dtype_code(dtype), _lib.stream_ptr()),
               "tdn_pack_conv_weight")
    return w_fwd, w_dg


def prepare_group(entries, dtype=BF16):
    """BN fold + weight pack of many plain conv units in one grouped launch (``tdn_prepare_group``).

    entries: list of (w fp32 OIHW parameter (any strides), bn or None, w_fwd, w_dgrad, fold) with bn = (gamma, beta,
    running_mean, running_var, eps) and the outputs preallocated: w_fwd [O,kh,kw,I], w_dgrad [I,kh,kw,O] in ``dtype``,
    fold float32 (3, O) (None without bn).  Bit-identical to bn_fold + pack_conv_weight per entry."""
    n = len(entries)
    if n == 0:
        return
    arr = (_lib.PrepItem * n)()
    for it, (w, bn, w_fwd, w_dg, fold) in zip(arr, entries):
        w = w.detach()
        if w.dtype != torch.float32 or not w.is_cuda or w.dim() != 4:
            raise ValueError("weight must be a CUDA float32 4-D tensor")
        O, I, kh, kw = w.shape
        _pack_out((w_fwd, w_dg), [(O, kh, kw, I), (I, kh, kw, O)], dtype, "prepare_group")
        it.w, it.w_fwd, it.w_dgrad = w.data_ptr(), w_fwd.data_ptr(), w_dg.data_ptr()
        it.s_o, it.s_i, it.s_h, it.s_w = w.stride()
        it.Cout, it.Cin, it.kh, it.kw = O, I, kh, kw
        if bn is not None:
            gamma, beta, mean, var, eps = bn
            for t, nm in ((gamma, "gamma"), (beta, "beta"), (mean, "running_mean"), (var, "running_var")):
                _chk_vec(t.detach(), nm, O)
            if fold is None or tuple(fold.shape) != (3, O) or fold.dtype != torch.float32 or not fold.is_contiguous():
                raise ValueError("prepare_group: fold must be a contiguous float32 (3, %d) tensor" % O)
            it.gamma, it.beta, it.mean, it.var = gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr()
            it.fold, it.eps = fold.data_ptr(), float(eps)
    _lib.check(_lib.load().tdn_prepare_group(arr, n, dtype_code(dtype), _lib.stream_ptr()), "tdn_prepare_group")


def pack_stem_weight(w, dtype=BF16, out=None):
    w = w.detach()
    if w.dtype != torch.float32 or not w.is_cuda or tuple(w.shape[1:]) != (3, 7, 7) or not w.is_contiguous():
        raise ValueError("stem weight must be a contiguous CUDA float32 [Cout,3,7,7] tensor")
    O = w.shape[0]
    if out is not None:
        (out,) = _pack_out((out,), [(O, 7, 8, 4)], dtype, "pack_stem_weight")
    else:
        out = torch.empty(O, 7, 8, 4, dtype=dtype, device=w.device)
    _lib.check(_lib.load().tdn_pack_stem_weight(_ptr(w), O, _ptr(out), dtype_code(dtype), _lib.stream_ptr()),
               "tdn_pack_stem_weight")
    return out


def _out_buffer(out, shape, dtype, name):
    """Caller-provided output (e.g. one image's slice of a batch tensor): must be exactly what would be allocated."""
    if tuple(out.shape) != tuple(shape) or out.dtype != dtype or not out.is_cuda or not out.is_contiguous():
        raise ValueError("%s: out must be a contiguous CUDA %s tensor of shape %s, got %s %s" %
                         (name, dtype, tuple(shape), out.dtype, tuple(out.shape)))
    return out


def conv2d_fwd(x, w_fwd, k, stride, pad, scale=None, shift=None, addend=None, addend_mode=ADD_NONE, relu=False,
               out_f32=False, out=None):
    _chk_act(x, "x")
    N, H, W, Cin = x.shape
    Cout = w_fwd.shape[0]
    if w_fwd.dtype != x.dtype or tuple(w_fwd.shape) != (Cout, k, k, Cin) or not w_fwd.is_contiguous():
        raise ValueError("w_fwd must be %s [Cout,k,k,Cin] contiguous, got %s %s" %
                         (x.dtype, w_fwd.dtype, tuple(w_fwd.shape)))
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    odt = torch.float32 if out_f32 else x.dtype
    y = torch.empty(N, Ho, Wo, Cout, dtype=odt, device=x.device) if out is None else \
        _out_buffer(out, (N, Ho, Wo, Cout), odt, "conv2d_fwd")
    ep = make_epilogue(Cout, Ho, Wo, scale, shift, addend, addend_mode, relu, None, N, out_f32, x.dtype, x.device)
    _lib.check(_lib.load().tdn_conv2d_fwd(_ptr(x), _ptr(w_fwd), _ptr(y), N, H, W, Cin, Cout, k, stride, pad,
                                          ctypes.byref(ep), dtype_code(x.dtype), _lib.stream_ptr()), "tdn_conv2d_fwd")
    return y


def conv2d_dgrad(g, w_dgrad, in_hw, k, stride, pad, addend=None, addend_mode=ADD_NONE, mask_src=None,
                 out_f32=False, out=None):
    """dx (N,H,W,Cin) from g (N,Ho,Wo,Cout); epilogue: + addend, then ReLU mask by mask_src > 0."""
    _chk_act(g, "g")
    N, Ho, Wo, Cout = g.shape
    H, W = in_hw
    Cin = w_dgrad.shape[0]
    if w_dgrad.dtype != g.dtype or tuple(w_dgrad.shape) != (Cin, k, k, Cout) or not w_dgrad.is_contiguous():
        raise ValueError("w_dgrad must be %s [Cin,k,k,Cout] contiguous, got %s %s" %
                         (g.dtype, w_dgrad.dtype, tuple(w_dgrad.shape)))
    if (Ho, Wo) != (conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)):
        raise RuntimeError("dgrad: g spatial size %s inconsistent with input %s" % ((Ho, Wo), (H, W)))
    odt = torch.float32 if out_f32 else g.dtype
    dx = torch.empty(N, H, W, Cin, dtype=odt, device=g.device) if out is None else \
        _out_buffer(out, (N, H, W, Cin), odt, "conv2d_dgrad")
    ep = make_epilogue(Cin, H, W, None, None, addend, addend_mode, False, mask_src, N, out_f32, g.dtype, g.device)
    _lib.check(_lib.load().tdn_conv2d_dgrad(_ptr(g), _ptr(w_dgrad), _ptr(dx), N, H, W, Cin, Cout, k, stride, pad,
                                            ctypes.byref(ep), dtype_code(g.dtype), _lib.stream_ptr()), "tdn_conv2d_dgrad")
    return dx


def bottleneck_supported(H, W, C, stride=1, dilation=1):
    """Does this build have a one-launch kernel for a stride-1 Bottleneck with C mid channels?"""
    return bool(_lib.load().tdn_bottleneck_supported(int(H), int(W), int(C), int(stride), int(dilation)))


def bottleneck_bit_planes(N, H, W, C, device):
    """Empty ReLU bit planes of one bottleneck (1 bit per element; include/tdn.h: tdn_bottleneck_args.bits1..3):
    (h1 > 0, h2 > 0) of C channels and (x > 0) of 4C channels, as int32 words [N][H][W][channels / 32]."""
    mk = lambda ch: torch.empty(N, H, W, ch // 32, dtype=torch.int32, device=device)
    return mk(C), mk(C), mk(4 * C)


def _bottleneck_args(name, a, w1, w2, w3, affine, masks, outs, bits=None, head=None):
    """``head``: None, 'fwd' (a has C channels, out3 4C) or 'bwd' (a has 4C channels, out3 C) — the layer1.0 geometry."""
    _chk_act(a, "in")
    N, H, W, CA = a.shape
    C = CA if head == 'fwd' else CA // 4
    C4 = 4 * C
    if head != 'fwd' and C * 4 != CA:
        raise ValueError("%s: %d input channels are not 4 x the mid channels" % (name, CA))
    C3 = C if head == 'bwd' else C4          # channels of out3
    for w, shp, nm in ((w1, (C, 1, 1, CA), "w1"), (w2, (C, 3, 3, C), "w2"), (w3, (C3, 1, 1, C), "w3")):
        if w.dtype != a.dtype or tuple(w.shape) != shp or not w.is_contiguous() or not w.is_cuda:
            raise ValueError("%s: %s must be %s %s contiguous, got %s %s" % (name, nm, a.dtype, shp, w.dtype,
                                                                              tuple(w.shape)))
    o1, o2, o3 = outs if outs is not None else (None, None, None)
    o1 = torch.empty(N, H, W, C, dtype=a.dtype, device=a.device) if o1 is None else _out_buffer(o1, (N, H, W, C), a.dtype, name)
    o2 = torch.empty(N, H, W, C, dtype=a.dtype, device=a.device) if o2 is None else _out_buffer(o2, (N, H, W, C), a.dtype, name)
    o3 = torch.empty(N, H, W, C3, dtype=a.dtype, device=a.device) if o3 is None else _out_buffer(o3, (N, H, W, C3), a.dtype, name)
    args = _lib.BottleneckArgs()
    args.in_, args.w1, args.w2, args.w3 = a.data_ptr(), w1.data_ptr(), w2.data_ptr(), w3.data_ptr()
    if affine is not None:
        for i, (v, n) in enumerate(zip(affine, (C, C, C, C, C4, C4))):
            _chk_vec(v, "affine[%d]" % i, n)
        sc1, sh1, sc2, sh2, sc3, sh3 = affine
        for fld, v in (("scale1", sc1), ("shift1", sh1), ("scale2", sc2), ("shift2", sh2), ("scale3", sc3),
                       ("shift3", sh3)):
            setattr(args, fld, v.data_ptr() if v is not None else None)
    if masks is not None:
        for fld, m, n in zip(("mask1", "mask2", "mask3"), masks, (C, C, C4)):
            if m is not None:
                _chk_act(m, fld, n, a.dtype)
                if tuple(m.shape[:3]) != (N, H, W):
                    raise ValueError("%s: %s has shape %s, expected (%d, %d, %d, %d)" % (name, fld, tuple(m.shape), N, H, W, n))
                setattr(args, fld, m.data_ptr())
    args.out1, args.out2, args.out3 = o1.data_ptr(), o2.data_ptr(), o3.data_ptr()
    args.N, args.H, args.W, args.C = N, H, W, C
    if bits is not None:
        for fld, b, ch in zip(("bits1", "bits2", "bits3"), bits, (C, C, C4)):
            if b is None:
                continue
            if b.dtype != torch.int32 or tuple(b.shape) != (N, H, W, ch // 32) or not b.is_contiguous() or not b.is_cuda:
                raise ValueError("%s: %s must be a contiguous CUDA int32 tensor of shape %s, got %s %s" %
                                 (name, fld, (N, H, W, ch // 32), b.dtype, tuple(b.shape)))
            setattr(args, fld, b.data_ptr())
    return args, (o1, o2, o3)


def bottleneck_fwd(x, w1, w2, w3, affine, outs=None, bits=None):
    """Stride-1 Bottleneck (resnet.py:97-119) in one launch: returns (h1, h2, out).  ``affine`` = (scale1, shift1,
    scale2, shift2, scale3, shift3) of the folded BNs; w_k = w_fwd packs of conv1 / conv2 / conv3.  ``bits`` = optional
    (b1, b2, b3) from ``bottleneck_bit_planes``: the kernel also writes h1 > 0, h2 > 0 and x > 0 as bit planes — the
    backward call reads those instead of the 16-bit tensors."""
    args, o = _bottleneck_args("bottleneck_fwd", x, w1, w2, w3, affine, None, outs, bits)
    _lib.check(_lib.load().tdn_bottleneck_fwd(ctypes.byref(args), dtype_code(x.dtype), _lib.stream_ptr()),
               "tdn_bottleneck_fwd")
    return o


def bottleneck_dgrad(g, w3d, w2d, w1d, masks, outs=None, bits=None):
    """Input-gradient chain of the same block in one launch: returns (g2, g1, dx) with g2 = mask(h2) . conv3^T(g),
    g1 = mask(h1) . conv2^T(g2), dx = mask(x) . (conv1^T(g1) + g).  ``masks`` = (h2, h1, x | None); w_kd = w_dgrad packs
    of conv3 / conv2 / conv1.  ``bits`` = the (b1, b2, b3) bit planes the forward call wrote: used INSTEAD of ``masks``."""
    C = g.shape[3] // 4
    if tuple(w3d.shape) != (C, 1, 1, 4 * C) or tuple(w1d.shape) != (4 * C, 1, 1, C):
        raise ValueError("bottleneck_dgrad: w_dgrad packs have shapes %s / %s" % (tuple(w3d.shape), tuple(w1d.shape)))
    if bits is not None and any(b is None for b in bits):
        raise ValueError("bottleneck_dgrad: all three bit planes or none")
    args, o = _bottleneck_args("bottleneck_dgrad", g, w3d, w2d, w1d, None, None if bits is not None else masks, outs, bits)
    _lib.check(_lib.load().tdn_bottleneck_dgrad(ctypes.byref(args), dtype_code(g.dtype), _lib.stream_ptr()),
               "tdn_bottleneck_dgrad")
    return o


def bottleneck_head_supported(H, W, Cin, C, stride=1, dilation=1):
    """Does this build have a one-launch kernel for a stage's first Bottleneck (1x1 downsample branch, Cin inputs)?"""
    return bool(_lib.load().tdn_bottleneck_head_supported(int(H), int(W), int(Cin), int(C), int(stride), int(dilation)))


def _head_args(args, addend, shape, dtype, name):
    _chk_act(addend, "addend", shape[3], dtype)
    if tuple(addend.shape) != tuple(shape):
        raise ValueError("%s: addend has shape %s, expected %s" % (name, tuple(addend.shape), tuple(shape)))
    h = _lib.BottleneckHeadArgs()
    h.b = args
    h.addend = addend.data_ptr()
    return h


def bottleneck_head_fwd(x, w1, w2, w3, affine, res=None, outs=None, bits=None, down=None):
    """First Bottleneck of layer1 (resnet.py:130-136: 1x1 downsample because inplanes != 4 * planes; stride 1) in one
    launch: x [N,H,W,C]; returns (h1, h2, out) with out = relu(bn3(conv3(h2)) + downsample(x)).  The downsample
    branch is either ``res`` [N,H,W,4C] (conv + BN, a launch of its own) or computed inside the launch from
    ``down`` = (w_fwd [4C,1,1,C], scale, shift).  ``bits`` = optional (b1, b2) planes."""
    N, H, W, C = x.shape
    if (res is None) == (down is None):
        raise ValueError("bottleneck_head_fwd: give either res or down")
    bits3 = (bits[0], bits[1], None) if bits is not None else None
    args, o = _bottleneck_args("bottleneck_head_fwd", x, w1, w2, w3, affine, None, outs, bits3, head='fwd')
    if res is not None:
        h = _head_args(args, res, (N, H, W, 4 * C), x.dtype, "bottleneck_head_fwd")
    else:
        wd, scd, shd = down
        if wd.dtype != x.dtype or tuple(wd.shape) != (4 * C, 1, 1, C) or not wd.is_contiguous() or not wd.is_cuda:
            raise ValueError("bottleneck_head_fwd: downsample weights must be %s %s contiguous" % (x.dtype, (4 * C, 1, 1, C)))
        h = _lib.BottleneckHeadArgs()
        h.b = args
        h.wd = wd.data_ptr()
        for fld, v in (("scale_d", scd), ("shift_d", shd)):
            if v is not None:
                _chk_vec(v, fld, 4 * C)
                setattr(h, fld, v.data_ptr())
    _lib.check(_lib.load().tdn_bottleneck_head_fwd(ctypes.byref(h), dtype_code(x.dtype), _lib.stream_ptr()),
               "tdn_bottleneck_head_fwd")
    return o


def bottleneck_head_dgrad(g, w3d, w2d, w1d, masks, t=None, outs=None, bits=None, down=None):
    """Input-gradient chain of that block in one launch: g [N,H,W,4C]; returns (g2, g1, dx) with
    dx = conv1^T(g1) + downsample^T(g) (no mask: x comes from the max pool).  The downsample conv's input gradient is
    either ``t`` [N,H,W,C] (a launch of its own) or computed inside the launch from ``down`` = its w_dgrad pack
    [C,1,1,4C].  ``masks`` = (h2, h1); ``bits`` = the (b1, b2) planes of the forward call, used instead."""
    N, H, W, C4 = g.shape
    C = C4 // 4
    if (t is None) == (down is None):
        raise ValueError("bottleneck_head_dgrad: give either t or down")
    if bits is not None and any(b is None for b in bits[:2]):
        raise ValueError("bottleneck_head_dgrad: both bit planes or none")
    bits3 = (bits[0], bits[1], None) if bits is not None else None
    m3 = None if bits is not None else (masks[0], masks[1], None)
    args, o = _bottleneck_args("bottleneck_head_dgrad", g, w3d, w2d, w1d, None, m3, outs, bits3, head='bwd')
    if t is not None:
        h = _head_args(args, t, (N, H, W, C), g.dtype, "bottleneck_head_dgrad")
    else:
        if down.dtype != g.dtype or tuple(down.shape) != (C, 1, 1, C4) or not down.is_contiguous() or not down.is_cuda:
            raise ValueError("bottleneck_head_dgrad: downsample w_dgrad must be %s %s contiguous" % (g.dtype, (C, 1, 1, C4)))
        h = _lib.BottleneckHeadArgs()
        h.b = args
        h.wd = down.data_ptr()
    _lib.check(_lib.load().tdn_bottleneck_head_dgrad(ctypes.byref(h), dtype_code(g.dtype), _lib.stream_ptr()),
               "tdn_bottleneck_head_dgrad")
    return o


_ws_cache = {}
_ws_retired = []   # outgrown workspaces stay referenced: a side stream may still be running kernels that use them


def _workspace(nbytes, device):
    """One grow-only scratch buffer per (device, launch stream); all users are ordered on that stream."""
    key = (device.index, _lib.stream_ptr())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


WGRAD_CONV, WGRAD_STEM, WGRAD_GCONV = 0, 1, 2


def wgrad_item(kind, x, g, w_fwd, scale, mean, invstd, dw, dgamma, dbeta, beta, N, H, W, Cin, Cout, k, stride, pad,
               groups=1):
    """One member of a grouped weight-gradient launch (``tdn_wgrad_item``); the tensors must stay referenced by the
    caller until the launch has been enqueued AND has run (they are raw pointers here)."""
    it = _lib.WgradItem()
    it.x, it.g, it.w_fwd = x.data_ptr(), g.data_ptr(), w_fwd.data_ptr()
    it.scale = scale.data_ptr() if scale is not None else None
    it.mean = mean.data_ptr() if mean is not None else None
    it.invstd = invstd.data_ptr() if invstd is not None else None
    it.dw = dw.data_ptr()
    it.dgamma = dgamma.data_ptr() if dgamma is not None else None
    it.dbeta = dbeta.data_ptr() if dbeta is not None else None
    it.beta = float(beta)
    it.kind = kind
    it.N, it.H, it.W, it.Cin, it.Cout, it.k, it.stride, it.pad, it.groups = N, H, W, Cin, Cout, k, stride, pad, groups
    return it


def wgrad_group(items, dtype, device):
    """Launch the weight / affine gradients of all ``items`` (``wgrad_item``) as one group on the routed stream:
    tdn_wgrad_group — a few gradient launches spanning every member + one finalize launch."""
    n = len(items)
    if n == 0:
        return
    arr = (_lib.WgradItem * n)(*items)
    lib = _lib.load()
    code = dtype_code(dtype)
    nbytes = lib.tdn_wgrad_group_workspace(arr, n, code)
    if nbytes < 0:
        _lib.check(-1, "tdn_wgrad_group_workspace")
    ws = _workspace(nbytes, device)
    _lib.check(lib.tdn_wgrad_group(arr, n, _ptr(ws), ws.numel(), code, _lib.stream_ptr()), "tdn_wgrad_group")


def wgrad_group_plan(items, dtype=BF16):
    """Host-only: (per_item rows of 8 ints, totals of 4 ints) — see tdn_wgrad_group_plan in include/tdn.h."""
    n = len(items)
    arr = (_lib.WgradItem * n)(*items)
    per = (ctypes.c_int32 * (8 * n))()
    tot = (ctypes.c_int32 * 4)()
    _lib.check(_lib.load().tdn_wgrad_group_plan(arr, n, dtype_code(dtype), per, tot), "tdn_wgrad_group_plan")
    return [list(per[8 * i:8 * i + 8]) for i in range(n)], list(tot)


def conv2d_wgrad_item(x, g, w_fwd, k, stride, pad, scale=None, mean=None, invstd=None, dw=None, dgamma=None,
                      dbeta=None, beta=0.0, want_dbeta=True):
    """Validate one conv's weight-gradient operands, allocate missing outputs; returns (item, dw, dgamma, dbeta)."""
    _chk_act(x, "x")
    _chk_act(g, "g", None, x.dtype)
    if w_fwd.dtype != x.dtype:
        raise ValueError("wgrad: w_fwd is %s but activations are %s" % (w_fwd.dtype, x.dtype))
    N, H, W, Cin = x.shape
    Cout = g.shape[3]
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    if tuple(g.shape) != (N, Ho, Wo, Cout) or tuple(w_fwd.shape) != (Cout, k, k, Cin):
        raise RuntimeError("wgrad: inconsistent shapes x=%s g=%s w=%s" %
                           (tuple(x.shape), tuple(g.shape), tuple(w_fwd.shape)))
    dev = x.device
    if dw is None:
        dw = torch.empty(Cout, k, k, Cin, dtype=torch.float32, device=dev)
    elif dw.dtype != torch.float32 or dw.numel() != Cout * k * k * Cin:
        raise ValueError("dw has wrong dtype/size")
    if dbeta is None and want_dbeta:
        dbeta = torch.empty(Cout, dtype=torch.float32, device=dev)
    if mean is not None and dgamma is None:
        dgamma = torch.empty(Cout, dtype=torch.float32, device=dev)
    for t, n in ((scale, "scale"), (mean, "mean"), (invstd, "invstd"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _chk_vec(t, n, Cout)
    it = wgrad_item(WGRAD_CONV, x, g, w_fwd, scale, mean, invstd, dw, dgamma, dbeta, beta, N, H, W, Cin, Cout, k,
                    stride, pad)
    return it, dw, dgamma, dbeta


def conv2d_wgrad(x, g, w_fwd, k, stride, pad, scale=None, mean=None, invstd=None, dw=None, dgamma=None,
                 dbeta=None, beta=0.0, want_dbeta=True):
    """Weight + affine grads. dw: fp32 [Cout,k,k,Cin] (written, or accumulated when beta=1).  ``want_dbeta=False``:
    the per-channel sum of g is not wanted (a conv without bias / with GroupNorm) — nothing is allocated for it.
    (A temporary here would be written by the finalize kernel on whatever stream the call is routed to and could be
    recycled by the allocator before that kernel has run.)"""
    it, dw, dgamma, dbeta = conv2d_wgrad_item(x, g, w_fwd, k, stride, pad, scale, mean, invstd, dw, dgamma, dbeta,
                                              beta, want_dbeta)
    wgrad_group([it], x.dtype, x.device)
    return dw, dgamma, dbeta


def stage_image(img, dtype=BF16):
    """NCHW float32 image batch (any strides) -> zero-padded NHWC4 `dtype` (N, H+6, W+8, 4)."""
    if img.dtype != torch.float32 or not img.is_cuda or img.dim() != 4 or img.shape[1] != 3:
        raise ValueError("image batch must be CUDA float32 (N,3,H,W), got %s %s" % (img.dtype, tuple(img.shape)))
    N, _, H, W = img.shape
    xp = torch.empty(N, H + 6, W + 8, 4, dtype=dtype, device=img.device)
    s = img.stride()
    _lib.check(_lib.load().tdn_stage_image(_ptr(img), s[0], s[1], s[2], s[3], N, H, W, _ptr(xp),
                                           dtype_code(dtype), _lib.stream_ptr()), "tdn_stage_image")
    return xp


def stem_conv_fwd(xp, w_stem, hw, scale=None, shift=None, relu=True, out_f32=False):
    H, W = hw
    N = xp.shape[0]
    Cout = w_stem.shape[0]
    if tuple(xp.shape) != (N, H + 6, W + 8, 4) or xp.dtype not in _CODES or not xp.is_contiguous():
        raise ValueError("xp must be the staged image (N,H+6,W+8,4) in bf16/fp16")
    if tuple(w_stem.shape) != (Cout, 7, 8, 4) or w_stem.dtype != xp.dtype:
        raise ValueError("w_stem must be %s [Cout,7,8,4]" % (xp.dtype,))
    if H % 2 or W % 2:
        raise RuntimeError("stem conv needs even H and W (got %dx%d); pad the batch as the reference pipeline does "
                           "(size_divisor, datasets/utils/image.py:326-347)" % (H, W))
    y = torch.empty(N, H // 2, W // 2, Cout, dtype=torch.float32 if out_f32 else xp.dtype, device=xp.device)
    ep = make_epilogue(Cout, H // 2, W // 2, scale, shift, None, ADD_NONE, relu, None, N, out_f32)
    _lib.check(_lib.load().tdn_stem_conv_fwd(_ptr(xp), _ptr(w_stem), _ptr(y), N, H, W, Cout, ctypes.byref(ep),
                                             dtype_code(xp.dtype), _lib.stream_ptr()), "tdn_stem_conv_fwd")
    return y


def stem_pool_fwd(xp, w_stem, hw, scale, shift, out=None):
    """conv7x7/s2 + folded BN + ReLU + MaxPool2d(3, 2, 1) in one launch (``tdn_stem_pool_fwd``): returns the pooled
    output and the window indices, bit-identical to ``maxpool3x3s2_fwd(stem_conv_fwd(...))``; 64 output channels."""
    H, W = hw
    N = xp.shape[0]
    Cout = w_stem.shape[0]
    if tuple(xp.shape) != (N, H + 6, W + 8, 4) or xp.dtype not in _CODES or not xp.is_contiguous():
        raise ValueError("xp must be the staged image (N,H+6,W+8,4) in bf16/fp16")
    if tuple(w_stem.shape) != (Cout, 7, 8, 4) or w_stem.dtype != xp.dtype or Cout != 64:
        raise ValueError("w_stem must be %s [64,7,8,4]" % (xp.dtype,))
    if H % 2 or W % 2:
        raise RuntimeError("stem conv needs even H and W (got %dx%d)" % (H, W))
    for t, nm in ((scale, "scale"), (shift, "shift")):
        if t is None or t.dtype != torch.float32 or t.numel() != Cout or not t.is_contiguous() or t.device != xp.device:
            raise ValueError("stem_pool_fwd: %s must be a contiguous float32 [%d] tensor on the image's device" % (nm, Cout))
    Ho, Wo = conv_out_size(H // 2, 3, 2, 1), conv_out_size(W // 2, 3, 2, 1)
    if out is not None:     # caller-provided buffers (one image range of batch tensors)
        y = _out_buffer(out[0], (N, Ho, Wo, Cout), xp.dtype, "stem_pool_fwd")
        idx = _out_buffer(out[1], (N, Ho, Wo, Cout), torch.uint8, "stem_pool_fwd")
    else:
        y = torch.empty(N, Ho, Wo, Cout, dtype=xp.dtype, device=xp.device)
        idx = torch.empty(N, Ho, Wo, Cout, dtype=torch.uint8, device=xp.device)
    _lib.check(_lib.load().tdn_stem_pool_fwd(_ptr(xp), _ptr(w_stem), _ptr(scale), _ptr(shift), _ptr(y), _ptr(idx), N, H,
                                             W, Cout, dtype_code(xp.dtype), _lib.stream_ptr()), "tdn_stem_pool_fwd")
    return y, idx


def stem_conv_wgrad_item(xp, g, w_stem, hw, scale=None, mean=None, invstd=None, dw=None, dgamma=None, dbeta=None,
                         beta=0.0, want_dbeta=True):
    H, W = hw
    N = xp.shape[0]
    Cout = w_stem.shape[0]
    _chk_act(g, "g", Cout, xp.dtype)
    if w_stem.dtype != xp.dtype:
        raise ValueError("stem wgrad: w_stem is %s but the staged image is %s" % (w_stem.dtype, xp.dtype))
    if tuple(g.shape) != (N, H // 2, W // 2, Cout) or tuple(xp.shape) != (N, H + 6, W + 8, 4):
        raise RuntimeError("stem wgrad: inconsistent shapes")
    dev = g.device
    if dw is None:
        dw = torch.empty(Cout, 3, 7, 7, dtype=torch.float32, device=dev)
    if dbeta is None and want_dbeta:
        dbeta = torch.empty(Cout, dtype=torch.float32, device=dev)
    if mean is not None and dgamma is None:
        dgamma = torch.empty(Cout, dtype=torch.float32, device=dev)
    for t, n in ((scale, "scale"), (mean, "mean"), (invstd, "invstd"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _chk_vec(t, n, Cout)
    it = wgrad_item(WGRAD_STEM, xp, g, w_stem, scale, mean, invstd, dw, dgamma, dbeta, beta, N, H, W, 3, Cout, 7, 2, 3)
    return it, dw, dgamma, dbeta


def stem_conv_wgrad(xp, g, w_stem, hw, scale=None, mean=None, invstd=None, dw=None, dgamma=None, dbeta=None,
                    beta=0.0, want_dbeta=True):
    it, dw, dgamma, dbeta = stem_conv_wgrad_item(xp, g, w_stem, hw, scale, mean, invstd, dw, dgamma, dbeta, beta,
                                                 want_dbeta)
    wgrad_group([it], xp.dtype, g.device)
    return dw, dgamma, dbeta


def maxpool3x3s2_fwd(x):
    _chk_act(x, "x")
    N, H, W, C = x.shape
    Ho, Wo = conv_out_size(H, 3, 2, 1), conv_out_size(W, 3, 2, 1)
    y = torch.empty(N, Ho, Wo, C, dtype=x.dtype, device=x.device)
    idx = torch.empty(N, Ho, Wo, C, dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().tdn_maxpool3x3s2_fwd(_ptr(x), _ptr(y), _ptr(idx), N, H, W, C, dtype_code(x.dtype),
                                                _lib.stream_ptr()), "tdn_maxpool3x3s2_fwd")
    return y, idx


def maxpool3x3s2_bwd(dy, idx, in_hw, mask_src=None, pooled=None):
    """Adjoint of MaxPool2d(3, 2, 1); with a ReLU in front of the pool either ``mask_src`` (the pool's input) or
    ``pooled`` (the pool's output — same result, a quarter of the bytes) supplies the ReLU mask."""
    _chk_act(dy, "dy")
    N, Ho, Wo, C = dy.shape
    H, W = in_hw
    if (Ho, Wo) != (conv_out_size(H, 3, 2, 1), conv_out_size(W, 3, 2, 1)) or tuple(idx.shape) != tuple(dy.shape):
        raise RuntimeError("maxpool bwd: inconsistent shapes")
    if mask_src is not None and pooled is not None:
        raise RuntimeError("maxpool bwd: give mask_src or pooled, not both")
    if mask_src is not None:
        _chk_act(mask_src, "mask_src", C, dy.dtype)
        if tuple(mask_src.shape) != (N, H, W, C):
            raise RuntimeError("maxpool bwd: mask_src shape mismatch")
    dx = torch.empty(N, H, W, C, dtype=dy.dtype, device=dy.device)
    if pooled is not None:
        _chk_act(pooled, "pooled", C, dy.dtype)
        if tuple(pooled.shape) != tuple(dy.shape):
            raise RuntimeError("maxpool bwd: pooled shape mismatch")
        _lib.check(_lib.load().tdn_maxpool3x3s2_relu_bwd(_ptr(dy), _ptr(idx), _ptr(pooled), _ptr(dx), N, H, W, C,
                                                         dtype_code(dy.dtype), _lib.stream_ptr()),
                   "tdn_maxpool3x3s2_relu_bwd")
        return dx
    _lib.check(_lib.load().tdn_maxpool3x3s2_bwd(_ptr(dy), _ptr(idx), _ptr(mask_src), _ptr(dx), N, H, W, C,
                                                dtype_code(dy.dtype), _lib.stream_ptr()), "tdn_maxpool3x3s2_bwd")
    return dx


def subsample2_fwd(x):
    _chk_act(x, "x")
    N, H, W, C = x.shape
    y = torch.empty(N, (H + 1) // 2, (W + 1) // 2, C, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().tdn_subsample2_fwd(_ptr(x), _ptr(y), N, H, W, C, dtype_code(x.dtype), _lib.stream_ptr()),
               "tdn_subsample2_fwd")
    return y


def subsample2_bwd(dy, in_hw, dx_in=None):
    _chk_act(dy, "dy")
    N, Ho, Wo, C = dy.shape
    H, W = in_hw
    if (Ho, Wo) != ((H + 1) // 2, (W + 1) // 2):
        raise RuntimeError("subsample bwd: inconsistent shapes")
    if dx_in is not None:
        _chk_act(dx_in, "dx_in", C, dy.dtype)
        if tuple(dx_in.shape) != (N, H, W, C):
            raise RuntimeError("subsample bwd: dx_in shape mismatch")
    dx = torch.empty(N, H, W, C, dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().tdn_subsample2_bwd(_ptr(dy), _ptr(dx_in), _ptr(dx), N, H, W, C, dtype_code(dy.dtype),
                                              _lib.stream_ptr()), "tdn_subsample2_bwd")
    return dx


def add_relu_mask(a, b=None, mask_src=None):
    _chk_act(a, "a")
    for t in (b, mask_src):
        if t is not None:
            _chk_act(t, "operand", None, a.dtype)
            if t.shape != a.shape:
                raise RuntimeError("add_relu_mask: shape mismatch")
    out = torch.empty_like(a)
    _lib.check(_lib.load().tdn_add_relu_mask(_ptr(a), _ptr(b), _ptr(mask_src), _ptr(out), a.numel(),
                                             dtype_code(a.dtype), _lib.stream_ptr()), "tdn_add_relu_mask")
    return out


def clamp_max_(y, hi):
    """In place y = min(y, hi): the upper clamp of nn.ReLU6 (layers.py:117-118) after a ReLU epilogue."""
    _chk_act(y, "y")
    _lib.check(_lib.load().tdn_clamp_max(_ptr(y), float(hi), y.numel(), dtype_code(y.dtype), _lib.stream_ptr()),
               "tdn_clamp_max")
    return y


def act_mask(g, y, hi=float("inf")):
    """g where 0 < y < hi else 0: backward of ReLU (hi = inf) / ReLU6 (hi = 6) from the activation's saved output."""
    _chk_act(g, "g")
    _chk_act(y, "y", None, g.dtype)
    if y.shape != g.shape:
        raise RuntimeError("act_mask: shape mismatch")
    out = torch.empty_like(g)
    _lib.check(_lib.load().tdn_act_mask(_ptr(g), _ptr(y), _ptr(out), float(hi), g.numel(), dtype_code(g.dtype),
                                        _lib.stream_ptr()), "tdn_act_mask")
    return out


def channel_affine_fwd(x, scale, shift, act=0):
    """act(x * scale[c] + shift[c]) on an NHWC activation: eval-mode BatchNorm2d (+ReLU / ReLU6) in front of a conv
    (ConvModule(activate_last=False), layers.py:129-134).  act: 0 none, 1 ReLU, 2 ReLU6."""
    _chk_act(x, "x")
    C = x.shape[3]
    _chk_vec(scale, "scale", C)
    _chk_vec(shift, "shift", C)
    y = torch.empty_like(x)
    _lib.check(_lib.load().tdn_channel_affine_fwd(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), x.numel() // C, C,
                                                  int(act), dtype_code(x.dtype), _lib.stream_ptr()),
               "tdn_channel_affine_fwd")
    return y


def channel_affine_bwd(g, x, scale, mean, invstd, dgamma=None, dbeta=None):
    """(dx, dgamma, dbeta) of channel_affine_fwd from g = dL/dy already masked by the activation."""
    _chk_act(g, "g")
    _chk_act(x, "x", None, g.dtype)
    C = x.shape[3]
    for t, nm in ((scale, "scale"), (mean, "mean"), (invstd, "invstd")):
        _chk_vec(t, nm, C)
    lib = _lib.load()
    npix = x.numel() // C
    dx = torch.empty_like(x)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    nbytes = lib.tdn_channel_affine_bwd_workspace(npix, C)
    ws = _workspace(nbytes, x.device)
    _lib.check(lib.tdn_channel_affine_bwd(_ptr(g), _ptr(x), _ptr(scale), _ptr(mean), _ptr(invstd), _ptr(dx),
                                          _ptr(dgamma), _ptr(dbeta), 0.0, npix, C, _ptr(ws), ws.numel(),
                                          dtype_code(x.dtype), _lib.stream_ptr()), "tdn_channel_affine_bwd")
    return dx, dgamma, dbeta


def to_nhwc_bf16(x, dtype=BF16):
    """Logical NCHW tensor -> NHWC `dtype` (N,H,W,C). Zero-copy when x already is a permuted NHWC tensor of that
    dtype.  (The name is historical: dtype may be torch.float16.)"""
    if x.dim() != 4 or not x.is_cuda:
        raise ValueError("expected a 4-D CUDA tensor, got %s on %s" % (tuple(x.shape), x.device))
    if x.dtype == dtype:
        xp = x.permute(0, 2, 3, 1)
        if xp.is_contiguous():
            return xp
        # a 16-bit tensor in another memory format (plain NCHW-contiguous cotangents, slices): the library's own
        # transpose, not a PyTorch copy — every launch of the path then is one a recorded launch plan contains
        _chk_dev(x, "x")
        N, C, H, W = x.shape
        out = torch.empty(N, H, W, C, dtype=dtype, device=x.device)
        s = x.stride()
        _lib.check(_lib.load().tdn_nchw16_to_nhwc(_ptr(x), s[0], s[1], s[2], s[3], N, C, H, W, _ptr(out),
                                                  _lib.stream_ptr()), "tdn_nchw16_to_nhwc")
        return out
    if x.dtype != torch.float32:
        x = x.float()
    N, C, H, W = x.shape
    out = torch.empty(N, H, W, C, dtype=dtype, device=x.device)
    s = x.stride()
    _lib.check(_lib.load().tdn_nchw_f32_to_nhwc(_ptr(x), s[0], s[1], s[2], s[3], N, C, H, W, _ptr(out),
                                                dtype_code(dtype), _lib.stream_ptr()), "tdn_nchw_f32_to_nhwc")
    return out


def nhwc_to_nchw_f32(x):
    _chk_act(x, "x")
    N, H, W, C = x.shape
    out = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().tdn_nhwc_to_nchw_f32(_ptr(x), N, C, H, W, _ptr(out), dtype_code(x.dtype),
                                                _lib.stream_ptr()), "tdn_nhwc_to_nchw_f32")
    return out


# ---- box ops ---------------------------------------------------------------------------------------
def _chk_boxes(b, name):
    if b.dtype != torch.float32 or not b.is_cuda or b.dim() != 2 or b.shape[1] != 4 or not b.is_contiguous():
        raise ValueError("%s must be a contiguous CUDA float32 (K,4) tensor" % name)
    _chk_dev(b, name)


def anchor_grid(base_anchors, featmap_size, stride, valid_size=None):
    _chk_boxes(base_anchors, "base_anchors")
    fh, fw = featmap_size
    A = base_anchors.shape[0]
    vh, vw = valid_size if valid_size is not None else (fh, fw)
    dev = base_anchors.device
    anchors = torch.empty(fh * fw * A, 4, dtype=torch.float32, device=dev)
    valid = torch.empty(fh * fw * A, dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().tdn_anchor_grid(_ptr(base_anchors), A, fh, fw, int(stride), int(vh), int(vw),
                                           _ptr(anchors), _ptr(valid), _lib.stream_ptr()), "tdn_anchor_grid")
    return anchors, valid


def anchor_pyramid(bases, featmap_sizes, strides, valid_sizes=None):
    """All levels of an anchor pyramid in ONE launch (``tdn_anchor_pyramid``).  bases: per-level base anchors (A_l, 4)
    on the GPU; returns (anchors (sum, 4) float32, valid (sum,) uint8, per-level row counts) — level after level, the
    rows of a level exactly what ``anchor_grid`` returns for it."""
    n = len(bases)
    if not (n == len(featmap_sizes) == len(strides)) or n == 0 or n > 8 or \
            (valid_sizes is not None and len(valid_sizes) != n):
        raise ValueError("anchor_pyramid takes 1..8 levels with one base / size / stride each")
    arr = (_lib.AnchorLevel * n)()
    counts = []
    for l, (base, (fh, fw), st) in enumerate(zip(bases, featmap_sizes, strides)):
        _chk_boxes(base, "base_anchors[%d]" % l)
        vh, vw = valid_sizes[l] if valid_sizes is not None else (fh, fw)
        arr[l].base_anchors, arr[l].A = base.data_ptr(), base.shape[0]
        arr[l].featH, arr[l].featW, arr[l].stride, arr[l].valid_h, arr[l].valid_w = int(fh), int(fw), int(st), int(vh), int(vw)
        counts.append(int(fh) * int(fw) * base.shape[0])
    dev = bases[0].device
    total = sum(counts)
    anchors = torch.empty(total, 4, dtype=torch.float32, device=dev)
    valid = torch.empty(total, dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().tdn_anchor_pyramid(arr, n, _ptr(anchors), _ptr(valid), _lib.stream_ptr()),
               "tdn_anchor_pyramid")
    return anchors, valid, counts


def bbox_iou_pairwise(a, b):
    _chk_boxes(a, "bboxes1")
    _chk_boxes(b, "bboxes2")
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().tdn_bbox_iou_pairwise(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out),
                                                 _lib.stream_ptr()), "tdn_bbox_iou_pairwise")
    return out


def _f4(vals):
    vals = [float(v) for v in vals]
    if len(vals) != 4:
        raise AssertionError("means / stds must have 4 entries")
    return (ctypes.c_float * 4)(*vals)


def bbox_normalize_(bbox, means, stds):
    """In place: bbox <- (bbox - means) / stds (datasets/utils/bbox.py:118-140). bbox: CUDA float32 (A, 4)."""
    _chk_boxes(bbox, "bbox")
    _lib.check(_lib.load().tdn_bbox_normalize(_ptr(bbox), bbox.shape[0], _f4(means), _f4(stds), _lib.stream_ptr()),
               "tdn_bbox_normalize")
    return bbox


def bbox_denormalize(bbox, means, stds):
    """bbox * stds + means, means/stds tiled over the 4C columns (datasets/utils/bbox.py:143-166)."""
    if bbox.dtype != torch.float32 or not bbox.is_cuda or bbox.dim() != 2 or not bbox.is_contiguous():
        raise ValueError("bbox must be a contiguous CUDA float32 (A, 4C) tensor")
    if bbox.shape[1] % 4:
        raise AssertionError("bbox.shape[1] must be a multiple of 4")
    out = torch.empty_like(bbox)
    _lib.check(_lib.load().tdn_bbox_denormalize(_ptr(bbox), _ptr(out), bbox.shape[0], bbox.shape[1], _f4(means),
                                                _f4(stds), _lib.stream_ptr()), "tdn_bbox_denormalize")
    return out


def nms(boxes, scores, iou_thr):
    _chk_boxes(boxes, "boxes")
    N = boxes.shape[0]
    if scores.dtype != torch.float32 or not scores.is_cuda or scores.numel() != N or not scores.is_contiguous():
        raise ValueError("scores must be a contiguous CUDA float32 (N,) tensor")
    dev = boxes.device
    keep = torch.zeros(N, dtype=torch.uint8, device=dev)
    kept_idx = torch.empty(N, dtype=torch.int64, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    nbytes = lib.tdn_nms_workspace(N)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    _lib.check(lib.tdn_nms(_ptr(boxes), _ptr(scores), N, float(iou_thr), _ptr(keep), _ptr(kept_idx), _ptr(num),
                           ctypes.c_void_p(ws.data_ptr() + off), nbytes, _lib.stream_ptr()), "tdn_nms")
    return keep, kept_idx, num


# ---- image batch staging ---------------------------------------------------------------------------
COLLATE_MAX = 16


def collate_images(images, means, stds, flips=None, batch_hw=None, size_divisor=32, staged=False, dtype=BF16):
    """normalize + flip + zero-pad + HWC->CHW + collate of already-resized images in one launch.

    images: list of CUDA uint8 or float32 tensors (H_i, W_i, 3), contiguous, all of one dtype.
    Returns float32 (N, 3, Hb, Wb) — or, with ``staged=True``, the stem's input (N, Hb+6, Wb+8, 4) in ``dtype``.
    Hb, Wb default to the largest image rounded up to ``size_divisor`` (image.py:340-347 + collate.py:52-56)."""
    N = len(images)
    if N == 0 or N > COLLATE_MAX:
        raise ValueError("collate_images takes 1..%d images per call, got %d" % (COLLATE_MAX, N))
    kind = images[0].dtype
    if kind not in (torch.uint8, torch.float32):
        raise ValueError("images must be uint8 or float32, got %s" % (kind,))
    dev = images[0].device
    for i, im in enumerate(images):
        if im.dtype != kind or not im.is_cuda or im.device != dev or im.dim() != 3 or im.shape[2] != 3 or \
                not im.is_contiguous() or im.numel() == 0:
            raise ValueError("image %d must be a non-empty contiguous CUDA %s (H, W, 3) tensor on %s, got %s %s %s" %
                             (i, kind, dev, im.dtype, im.device, tuple(im.shape)))
    if len(means) != 3 or len(stds) != 3:
        raise ValueError("means / stds must have 3 entries")
    d = int(size_divisor) if size_divisor else 1
    if batch_hw is None:
        Hb = max(-(-im.shape[0] // d) * d for im in images)
        Wb = max(-(-im.shape[1] // d) * d for im in images)
    else:
        Hb, Wb = batch_hw
        if any(im.shape[0] > Hb or im.shape[1] > Wb for im in images):
            raise RuntimeError("collate_images: an image is larger than the batch size %s" % ((Hb, Wb),))
    if flips is not None and len(flips) != N:
        raise ValueError("flips must have one flag per image")
    ptrs = (ctypes.c_void_p * N)(*[im.data_ptr() for im in images])
    hw = (ctypes.c_int32 * (2 * N))(*[v for im in images for v in (im.shape[0], im.shape[1])])
    fl = (ctypes.c_uint8 * N)(*[1 if f else 0 for f in flips]) if flips is not None else None
    m3 = (ctypes.c_float * 3)(*[float(v) for v in means])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in stds])
    if staged:
        out = torch.empty(N, Hb + 6, Wb + 8, 4, dtype=dtype, device=dev)
        code = dtype_code(dtype)
    else:
        out = torch.empty(N, 3, Hb, Wb, dtype=torch.float32, device=dev)
        code = TDN_BF16
    _lib.check(_lib.load().tdn_collate_images(ptrs, hw, fl, N, 0 if kind == torch.uint8 else 1, m3, s3, Hb, Wb,
                                              _ptr(out), 1 if staged else 0, code, _lib.stream_ptr()),
               "tdn_collate_images")
    return out


# ---- GroupNorm ----------------------------------------------------------------------------------------
def _gn_ws(N, H, W, C, G, device):
    nbytes = _lib.load().tdn_gn_workspace(N, H, W, C, G)
    if nbytes < 0:
        _lib.check(-1, "tdn_gn_workspace")
    return _workspace(nbytes, device)


def gn_fwd(z, gamma, beta, groups, eps=1e-5, addend=None, relu=False, addend_mode=ADD_SAME):
    """y = relu?(GroupNorm(z) (+ addend)) on an NHWC raw conv output; returns (y, stats) with stats (N, C, 2) float32
    = per-channel (mean, rstd) for the backward."""
    _chk_act(z, "z")
    N, H, W, C = z.shape
    _chk_vec(gamma.detach(), "gamma", C)
    _chk_vec(beta.detach(), "beta", C)
    if addend is not None:
        _chk_act(addend, "addend", C, z.dtype)
        if addend_mode not in (ADD_SAME, ADD_UP2X):
            raise ValueError("gn_fwd: addend_mode must be ADD_SAME or ADD_UP2X")
        exp = (N, H, W, C) if addend_mode == ADD_SAME else (N, H // 2, W // 2, C)
        if tuple(addend.shape) != exp or (addend_mode == ADD_UP2X and (H % 2 or W % 2)):
            raise RuntimeError("epilogue addend of spatial size %s does not match output %s (mode %d)" %
                               (tuple(addend.shape[1:3]), (H, W), addend_mode))
    y = torch.empty_like(z)
    stats = torch.empty(N, C, 2, dtype=torch.float32, device=z.device)
    ws = _gn_ws(N, H, W, C, groups, z.device)
    _lib.check(_lib.load().tdn_gn_fwd(_ptr(z), _ptr(gamma), _ptr(beta), N, H, W, C, int(groups), float(eps),
                                      _ptr(addend), int(addend_mode), _relu_code(relu), _ptr(y), _ptr(stats), _ptr(ws),
                                      ws.numel(),
                                      dtype_code(z.dtype), _lib.stream_ptr()), "tdn_gn_fwd")
    return y, stats


def gn_bwd(g, z, stats, gamma, groups, dgamma=None, dbeta=None, accumulate=False):
    """(dz, dgamma, dbeta) from g = dL/dy (ReLU mask already applied)."""
    _chk_act(z, "z")
    _chk_act(g, "g", z.shape[3], z.dtype)
    N, H, W, C = z.shape
    if g.shape != z.shape or tuple(stats.shape) != (N, C, 2) or stats.dtype != torch.float32:
        raise RuntimeError("gn_bwd: inconsistent shapes g=%s z=%s stats=%s" %
                           (tuple(g.shape), tuple(z.shape), tuple(stats.shape)))
    _chk_vec(gamma.detach(), "gamma", C)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=z.device)
    if dbeta is None:
        dbeta = torch.empty(C, dtype=torch.float32, device=z.device)
    _chk_vec(dgamma, "dgamma", C)
    _chk_vec(dbeta, "dbeta", C)
    dz = torch.empty_like(z)
    ws = _gn_ws(N, H, W, C, groups, z.device)
    _lib.check(_lib.load().tdn_gn_bwd(_ptr(g), _ptr(z), _ptr(stats), _ptr(gamma), N, H, W, C, int(groups), _ptr(dz),
                                      _ptr(dgamma), _ptr(dbeta), 1.0 if accumulate else 0.0, _ptr(ws), ws.numel(),
                                      dtype_code(z.dtype), _lib.stream_ptr()), "tdn_gn_bwd")
    return dz, dgamma, dbeta


# ---- grouped convolution (ResNeXt) ---------------------------------------------------------------------
def pack_gconv_weight(w, groups, scale=None, want_dgrad=True, dtype=BF16, out=None):
    """Grouped conv weight fp32 [C][C/groups][kh][kw] (any strides) -> block-diagonal operands
    (w_fwd [C][kh][kw][64], w_dgrad [C][kh][kw][64] with the BN scale folded)."""
    w = w.detach()
    if w.dtype != torch.float32 or not w.is_cuda or w.dim() != 4:
        raise ValueError("weight must be a CUDA float32 4-D tensor")
    C, cpg, kh, kw = w.shape
    if cpg * groups != C:
        raise ValueError("grouped weight %s does not match %d groups" % (tuple(w.shape), groups))
    _chk_vec(scale, "scale", C)
    if out is not None:
        w_fwd, w_dg = _pack_out(out, [(C, kh, kw, 64), (C, kh, kw, 64)], dtype, "pack_gconv_weight")
    else:
        w_fwd = torch.empty(C, kh, kw, 64, dtype=dtype, device=w.device)
        w_dg = torch.empty(C, kh, kw, 64, dtype=dtype, device=w.device) if want_dgrad else None
    s = w.stride()
    _lib.check(_lib.load().tdn_pack_gconv_weight(_ptr(w), s[0], s[1], s[2], s[3], C, int(groups), kh, kw, _ptr(scale),
                                                 _ptr(w_fwd), _ptr(w_dg), dtype_code(dtype), _lib.stream_ptr()),
               "tdn_pack_gconv_weight")
    return w_fwd, w_dg


def _chk_gw(wp, name, C, k, dtype):
    if wp.dtype != dtype or tuple(wp.shape) != (C, k, k, 64) or not wp.is_contiguous():
        raise ValueError("%s must be %s [C,k,k,64] contiguous (pack_gconv_weight), got %s %s" %
                         (name, dtype, wp.dtype, tuple(wp.shape)))


def gconv2d_fwd(x, w_fwd, groups, k, stride, pad, scale=None, shift=None, addend=None, addend_mode=ADD_NONE,
                relu=False, out_f32=False):
    _chk_act(x, "x")
    N, H, W, C = x.shape
    _chk_gw(w_fwd, "w_fwd", C, k, x.dtype)
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    y = torch.empty(N, Ho, Wo, C, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    ep = make_epilogue(C, Ho, Wo, scale, shift, addend, addend_mode, relu, None, N, out_f32, x.dtype)
    _lib.check(_lib.load().tdn_gconv2d_fwd(_ptr(x), _ptr(w_fwd), _ptr(y), N, H, W, C, int(groups), k, stride, pad,
                                           ctypes.byref(ep), dtype_code(x.dtype), _lib.stream_ptr()),
               "tdn_gconv2d_fwd")
    return y


def gconv2d_dgrad(g, w_dgrad, groups, in_hw, k, stride, pad, addend=None, addend_mode=ADD_NONE, mask_src=None,
                  out_f32=False):
    _chk_act(g, "g")
    N, Ho, Wo, C = g.shape
    H, W = in_hw
    _chk_gw(w_dgrad, "w_dgrad", C, k, g.dtype)
    if (Ho, Wo) != (conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)):
        raise RuntimeError("dgrad: g spatial size %s inconsistent with input %s" % ((Ho, Wo), (H, W)))
    dx = torch.empty(N, H, W, C, dtype=torch.float32 if out_f32 else g.dtype, device=g.device)
    ep = make_epilogue(C, H, W, None, None, addend, addend_mode, False, mask_src, N, out_f32, g.dtype)
    _lib.check(_lib.load().tdn_gconv2d_dgrad(_ptr(g), _ptr(w_dgrad), _ptr(dx), N, H, W, C, int(groups), k, stride,
                                             pad, ctypes.byref(ep), dtype_code(g.dtype), _lib.stream_ptr()),
               "tdn_gconv2d_dgrad")
    return dx


def gconv2d_wgrad_item(x, g, w_fwd, groups, k, stride, pad, scale=None, mean=None, invstd=None, dw=None, dgamma=None,
                       dbeta=None, beta=0.0, want_dbeta=True):
    _chk_act(x, "x")
    _chk_act(g, "g", None, x.dtype)
    N, H, W, C = x.shape
    _chk_gw(w_fwd, "w_fwd", C, k, x.dtype)
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    if tuple(g.shape) != (N, Ho, Wo, C):
        raise RuntimeError("gconv wgrad: inconsistent shapes x=%s g=%s" % (tuple(x.shape), tuple(g.shape)))
    cpg = C // groups
    dev = x.device
    if dw is None:
        dw = torch.empty(C, k, k, cpg, dtype=torch.float32, device=dev)
    elif dw.dtype != torch.float32 or dw.numel() != C * k * k * cpg:
        raise ValueError("dw has wrong dtype/size")
    if dbeta is None and want_dbeta:
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
    if mean is not None and dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
    for t, n in ((scale, "scale"), (mean, "mean"), (invstd, "invstd"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _chk_vec(t, n, C)
    it = wgrad_item(WGRAD_GCONV, x, g, w_fwd, scale, mean, invstd, dw, dgamma, dbeta, beta, N, H, W, C, C, k, stride,
                    pad, int(groups))
    return it, dw, dgamma, dbeta


def gconv2d_wgrad(x, g, w_fwd, groups, k, stride, pad, scale=None, mean=None, invstd=None, dw=None, dgamma=None,
                  dbeta=None, beta=0.0, want_dbeta=True):
    """dw fp32 [C,k,k,C/groups] (+ BN affine grads like conv2d_wgrad)."""
    it, dw, dgamma, dbeta = gconv2d_wgrad_item(x, g, w_fwd, groups, k, stride, pad, scale, mean, invstd, dw, dgamma,
                                               dbeta, beta, want_dbeta)
    wgrad_group([it], x.dtype, x.device)
    return dw, dgamma, dbeta


# ---- BatchNorm2d with batch statistics (training mode) ----------------------------------------------------
def bn_train_fwd(z, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, addend=None,
                 relu=False, addend_mode=ADD_SAME):
    """y = relu?(BatchNorm_train(z) (+ addend)); running statistics (if given) are updated in place like
    nn.BatchNorm2d does.  Returns (y, stats) with stats (N, C, 2) float32 for the backward."""
    _chk_act(z, "z")
    N, H, W, C = z.shape
    _chk_vec(gamma.detach(), "gamma", C)
    _chk_vec(beta.detach(), "beta", C)
    _chk_vec(running_mean, "running_mean", C)
    _chk_vec(running_var, "running_var", C)
    if (running_mean is None) != (running_var is None):
        raise ValueError("bn_train_fwd: give both running statistics or none")
    if addend is not None:
        _chk_act(addend, "addend", C, z.dtype)
        exp = (N, H, W, C) if addend_mode == ADD_SAME else (N, H // 2, W // 2, C)
        if addend_mode not in (ADD_SAME, ADD_UP2X) or tuple(addend.shape) != exp:
            raise RuntimeError("epilogue addend of spatial size %s does not match output %s (mode %d)" %
                               (tuple(addend.shape[1:3]), (H, W), addend_mode))
    y = torch.empty_like(z)
    stats = torch.empty(N, C, 2, dtype=torch.float32, device=z.device)
    ws = _gn_ws(N, H, W, C, C, z.device)
    _lib.check(_lib.load().tdn_bn_train_fwd(_ptr(z), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                            float(momentum), N, H, W, C, float(eps), _ptr(addend), int(addend_mode),
                                            _relu_code(relu), _ptr(y), _ptr(stats), _ptr(ws), ws.numel(),
                                            dtype_code(z.dtype), _lib.stream_ptr()), "tdn_bn_train_fwd")
    return y, stats


def bn_train_bwd(g, z, stats, gamma, dgamma=None, dbeta=None, accumulate=False):
    """(dz, dgamma, dbeta) of training-mode BatchNorm from g = dL/dy (ReLU mask already applied)."""
    _chk_act(z, "z")
    _chk_act(g, "g", z.shape[3], z.dtype)
    N, H, W, C = z.shape
    if g.shape != z.shape or tuple(stats.shape) != (N, C, 2) or stats.dtype != torch.float32:
        raise RuntimeError("bn_train_bwd: inconsistent shapes")
    _chk_vec(gamma.detach(), "gamma", C)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=z.device)
    if dbeta is None:
        dbeta = torch.empty(C, dtype=torch.float32, device=z.device)
    _chk_vec(dgamma, "dgamma", C)
    _chk_vec(dbeta, "dbeta", C)
    dz = torch.empty_like(z)
    ws = _gn_ws(N, H, W, C, C, z.device)
    _lib.check(_lib.load().tdn_bn_train_bwd(_ptr(g), _ptr(z), _ptr(stats), _ptr(gamma), N, H, W, C, _ptr(dz),
                                            _ptr(dgamma), _ptr(dbeta), 1.0 if accumulate else 0.0, _ptr(ws),
                                            ws.numel(), dtype_code(z.dtype), _lib.stream_ptr()), "tdn_bn_train_bwd")
    return dz, dgamma, dbeta
