"""torch_detection_amd — MI355X-native ResNet/FPN + box-op hot path behind the Torch_Detection registry.

Drop-in surface (same names as the reference's ``models`` package): ``BACKBONES``, ``NECKS``, ``ResNet``,
``FPN``, ``PAFPN``, ``ConvModule``, the conv/norm builders and init helpers, plus the steps either side of it
(``ImageTransforms`` device-side batch staging, box ops, ``bbox_normalize`` / ``bbox_denormalize``).  Everything computes through libtdn.so
(hand-written gfx950 HIP kernels, C ABI in include/tdn.h); there is no CPU or eager fallback.
"""
__version__ = "0.1.0"

from .registry import BACKBONES, NECKS, Registry  # noqa: F401
from .layers import (ConvModule, conv1x1_group, conv3x3_group, conv7x7_group, get_group_gn,  # noqa: F401
                     norm_layer)
from .inits import (bias_init_with_prob, constant_init, kaiming_init, normal_init, uniform_init,  # noqa: F401
                    xavier_init)
from .checkpoint import load_checkpoint, load_state_dict, save_checkpoint  # noqa: F401
from .backbone import (BasicBlock, Bottleneck, ResNet, ResNeXt, ResNeXtBasicBlock,  # noqa: F401
                       ResNeXtBottleneck)
from .necks import FPN, PAFPN  # noqa: F401
from .staging import ImageTransforms, StagedImages  # noqa: F401
from .graph import GraphedStep, PreparedStep  # noqa: F401
from .functional import invalidate_packed  # noqa: F401
from .box import (AnchorGenerator, anchor_pyramid, bbox_denormalize, bbox_normalize, bbox_overlaps, nms,  # noqa: F401
                  nms_mask)
