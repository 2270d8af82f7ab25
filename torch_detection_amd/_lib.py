"""ctypes binding of libtdn.so (the C ABI declared in include/tdn.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C torch_detection_amd/csrc``.
There is no fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (loads the HIP runtime first so libtdn binds to the same libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TDN_LIB: another build of the library in the package directory — scripts/ use libtdn_trace.so (`make TRACE=1`: ablation
# and cycle-stamp instantiations) for their sweeps; the package, the tests and bench.py use the product, libtdn.so
LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("TDN_LIB", "") or "libtdn.so"))

TDN_BF16 = 0
TDN_F16 = 1
SPLITK_TICKET_BYTES = 65536
ADD_NONE, ADD_SAME, ADD_UP2X, ADD_SUMPOOL2 = 0, 1, 2, 3

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_float = ctypes.c_float


class Epilogue(ctypes.Structure):
    """Mirror of ``tdn_epilogue`` (include/tdn.h)."""
    _fields_ = [
        ("scale", c_void_p),
        ("shift", c_void_p),
        ("addend", c_void_p),
        ("addend_mode", ctypes.c_int32),
        ("addend_h", ctypes.c_int32),
        ("addend_w", ctypes.c_int32),
        ("relu", ctypes.c_int32),
        ("mask_src", c_void_p),
        ("out_f32", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("splitk_ws", c_void_p),
        ("splitk_ws_bytes", c_i64),
    ]


_EP = ctypes.POINTER(Epilogue)


class WgradItem(ctypes.Structure):
    """Mirror of ``tdn_wgrad_item`` (include/tdn.h): one member of a grouped weight-gradient launch."""
    _fields_ = [
        ("x", c_void_p), ("g", c_void_p), ("w_fwd", c_void_p),
        ("scale", c_void_p), ("mean", c_void_p), ("invstd", c_void_p),
        ("dw", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p),
        ("beta", c_float), ("kind", ctypes.c_int32),
        ("N", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("k", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
        ("groups", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


_WI = ctypes.POINTER(WgradItem)


class PrepItem(ctypes.Structure):
    """Mirror of ``tdn_prep_item`` (include/tdn.h): one conv unit of a grouped fold + pack launch."""
    _fields_ = [
        ("w", c_void_p), ("w_fwd", c_void_p), ("w_dgrad", c_void_p),
        ("gamma", c_void_p), ("beta", c_void_p), ("mean", c_void_p), ("var", c_void_p), ("fold", c_void_p),
        ("s_o", c_i64), ("s_i", c_i64), ("s_h", c_i64), ("s_w", c_i64),
        ("Cout", ctypes.c_int32), ("Cin", ctypes.c_int32), ("kh", ctypes.c_int32), ("kw", ctypes.c_int32),
        ("eps", c_float), ("reserved", ctypes.c_int32),
    ]


_PI = ctypes.POINTER(PrepItem)


class AnchorLevel(ctypes.Structure):
    """Mirror of ``tdn_anchor_level`` (include/tdn.h)."""
    _fields_ = [("base_anchors", c_void_p), ("A", ctypes.c_int32), ("featH", ctypes.c_int32),
                ("featW", ctypes.c_int32), ("stride", ctypes.c_int32), ("valid_h", ctypes.c_int32),
                ("valid_w", ctypes.c_int32)]


_AL = ctypes.POINTER(AnchorLevel)


class BottleneckArgs(ctypes.Structure):
    """Mirror of ``tdn_bottleneck_args`` (include/tdn.h)."""
    _fields_ = [("in_", c_void_p), ("w1", c_void_p), ("w2", c_void_p), ("w3", c_void_p),
                ("scale1", c_void_p), ("shift1", c_void_p), ("scale2", c_void_p), ("shift2", c_void_p),
                ("scale3", c_void_p), ("shift3", c_void_p),
                ("mask1", c_void_p), ("mask2", c_void_p), ("mask3", c_void_p),
                ("out1", c_void_p), ("out2", c_void_p), ("out3", c_void_p),
                ("N", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("C", ctypes.c_int32),
                ("bits1", c_void_p), ("bits2", c_void_p), ("bits3", c_void_p)]


_BA = ctypes.POINTER(BottleneckArgs)


class BottleneckHeadArgs(ctypes.Structure):
    """Mirror of ``tdn_bottleneck_head_args`` (include/tdn.h)."""
    _fields_ = [("b", BottleneckArgs), ("addend", c_void_p), ("wd", c_void_p), ("scale_d", c_void_p),
                ("shift_d", c_void_p)]


_BHA = ctypes.POINTER(BottleneckHeadArgs)

# name -> (restype, argtypes); must list every symbol of include/tdn.h (tests/test_abi.py checks this)
SIGNATURES = {
    "tdn_last_error": (ctypes.c_char_p, []),
    "tdn_version": (c_int, []),
    "tdn_bn_fold": (c_int, [c_void_p] * 4 + [c_float, c_int] + [c_void_p] * 3 + [c_void_p]),
    "tdn_pack_conv_weight": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "tdn_pack_stem_weight": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "tdn_prepare_group": (c_int, [_PI, c_int, c_int, c_void_p]),
    "tdn_conv2d_fwd": (c_int, [c_void_p] * 3 + [c_int] * 8 + [_EP, c_int, c_void_p]),
    "tdn_conv2d_dgrad": (c_int, [c_void_p] * 3 + [c_int] * 8 + [_EP, c_int, c_void_p]),
    "tdn_conv2d_wgrad_workspace": (c_i64, [c_int] * 8),
    "tdn_conv2d_wgrad": (c_int, [c_void_p] * 9 + [c_float] + [c_int] * 8 + [c_void_p, c_i64, c_int, c_void_p]),
    "tdn_wgrad_group_workspace": (c_i64, [_WI, c_int, c_int]),
    "tdn_wgrad_group": (c_int, [_WI, c_int, c_void_p, c_i64, c_int, c_void_p]),
    "tdn_wgrad_group_plan": (c_int, [_WI, c_int, c_int, ctypes.POINTER(ctypes.c_int32),
                                     ctypes.POINTER(ctypes.c_int32)]),
    "tdn_stage_image": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_void_p, c_int,
                                c_void_p]),
    "tdn_stem_conv_fwd": (c_int, [c_void_p] * 3 + [c_int] * 4 + [_EP, c_int, c_void_p]),
    "tdn_stem_pool_fwd": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "tdn_stem_conv_wgrad_workspace": (c_i64, [c_int] * 4),
    "tdn_stem_conv_wgrad": (c_int, [c_void_p] * 9 + [c_float] + [c_int] * 4 + [c_void_p, c_i64, c_int, c_void_p]),
    "tdn_maxpool3x3s2_fwd": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "tdn_maxpool3x3s2_bwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "tdn_maxpool3x3s2_relu_bwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "tdn_subsample2_fwd": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    "tdn_subsample2_bwd": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "tdn_add_relu_mask": (c_int, [c_void_p] * 4 + [c_i64, c_int, c_void_p]),
    "tdn_clamp_max": (c_int, [c_void_p, c_float, c_i64, c_int, c_void_p]),
    "tdn_act_mask": (c_int, [c_void_p] * 3 + [c_float, c_i64, c_int, c_void_p]),
    "tdn_channel_affine_fwd": (c_int, [c_void_p] * 4 + [c_i64, c_int, c_int, c_int, c_void_p]),
    "tdn_channel_affine_bwd_workspace": (c_i64, [c_i64, c_int]),
    "tdn_channel_affine_bwd": (c_int, [c_void_p] * 8 + [c_float, c_i64, c_int, c_void_p, c_i64, c_int, c_void_p]),
    "tdn_nchw_f32_to_nhwc": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_void_p,
                                     c_int, c_void_p]),
    "tdn_nchw16_to_nhwc": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_void_p,
                                   c_void_p]),
    "tdn_nhwc_to_nchw_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "tdn_anchor_grid": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tdn_anchor_pyramid": (c_int, [_AL, c_int, c_void_p, c_void_p, c_void_p]),
    "tdn_bbox_iou_pairwise": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "tdn_nms_workspace": (c_i64, [c_int]),
    "tdn_nms": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                        c_void_p]),
    "tdn_bbox_normalize": (c_int, [c_void_p, c_i64, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p]),
    "tdn_bbox_denormalize": (c_int, [c_void_p, c_void_p, c_i64, c_int, ctypes.POINTER(c_float),
                                     ctypes.POINTER(c_float), c_void_p]),
    "tdn_pack_gconv_weight": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_void_p,
                                      c_void_p, c_void_p, c_int, c_void_p]),
    "tdn_bottleneck_supported": (c_int, [c_int] * 5),
    "tdn_bottleneck_fwd": (c_int, [_BA, c_int, c_void_p]),
    "tdn_bottleneck_dgrad": (c_int, [_BA, c_int, c_void_p]),
    "tdn_bottleneck_head_supported": (c_int, [c_int] * 6),
    "tdn_bottleneck_head_fwd": (c_int, [_BHA, c_int, c_void_p]),
    "tdn_bottleneck_head_dgrad": (c_int, [_BHA, c_int, c_void_p]),
    "tdn_gconv2d_fwd": (c_int, [c_void_p] * 3 + [c_int] * 8 + [_EP, c_int, c_void_p]),
    "tdn_gconv2d_dgrad": (c_int, [c_void_p] * 3 + [c_int] * 8 + [_EP, c_int, c_void_p]),
    "tdn_gconv2d_wgrad_workspace": (c_i64, [c_int] * 8),
    "tdn_gconv2d_wgrad": (c_int, [c_void_p] * 9 + [c_float] + [c_int] * 8 + [c_void_p, c_i64, c_int, c_void_p]),
    "tdn_gn_workspace": (c_i64, [c_int] * 5),
    "tdn_gn_fwd": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_float, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                           c_i64, c_int, c_void_p]),
    "tdn_gn_bwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_i64,
                           c_int, c_void_p]),
    "tdn_bn_train_fwd": (c_int, [c_void_p] * 5 + [c_float] + [c_int] * 4 + [c_float, c_void_p, c_int, c_int, c_void_p,
                                 c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "tdn_bn_train_bwd": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_i64,
                                 c_int, c_void_p]),
    "tdn_collate_images": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32),
                                   ctypes.POINTER(ctypes.c_uint8), c_int, c_int, ctypes.POINTER(c_float),
                                   ctypes.POINTER(c_float), c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "tdn_plan_begin": (c_int, []),
    "tdn_plan_end": (c_void_p, []),
    "tdn_plan_event_record": (c_int, [c_void_p]),
    "tdn_plan_stream_wait": (c_int, [c_void_p, c_int]),
    "tdn_plan_run": (c_int, [c_void_p]),
    "tdn_plan_stats": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "tdn_plan_free": (c_int, [c_void_p]),
    "tdn_probe_xcd_mapping": (c_int, []),
    "tdn_conv2d_plan": (c_int, [c_int] * 9 + [ctypes.POINTER(ctypes.c_int32)]),
    "tdn_debug_trace": (c_int, [c_void_p, ctypes.c_longlong]),
}

_lib = None


def load():
    """Load libtdn.so (once) and attach prototypes. Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "torch_detection_amd: native library %s is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C torch_detection_amd/csrc`. "
            "There is no CPU / eager fallback for this path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    if torch.cuda.is_available() and os.environ.get("TDN_SPLITK", "0") == "1":
        lib.tdn_probe_xcd_mapping()    # once per process, before anything could be capturing a stream
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().tdn_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


_tls = threading.local()


def stream_ptr():
    """Raw hipStream_t the kernels are enqueued on: PyTorch's current stream, or the stream set by
    ``use_stream`` for this thread (the weight-gradient side stream of functional.py)."""
    ov = getattr(_tls, "stream", None)
    if ov is not None:
        return ov
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def set_stream_override(raw_stream):
    """Route this thread's kernel launches to ``raw_stream`` (int) until reset with ``None``.  Cheaper than
    ``with torch.cuda.stream(...)`` (no allocator / current-stream switching); allocations stay on the current
    PyTorch stream, so callers must order memory reuse themselves (functional._side_refs)."""
    prev = getattr(_tls, "stream", None)
    _tls.stream = raw_stream
    return prev
