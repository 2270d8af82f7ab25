"""ResNeXt backbone on the MI355X HIP path — drop-in for the reference's ``models/backbone/resnext.py``.

Same class names, constructor signatures (``ResNeXt(depth, base_width, cardinality, ...)``), attribute names and
state_dict keys as the reference (resnext.py:12-330).  The only arithmetic that differs from ResNet is the grouped 3x3
convolution (``conv3x3_group(..., groups=cardinality)``, resnext.py:26-28,82-83), which runs in block-diagonal form on
the same GEMM kernels (``tdn_gconv2d_*``): the channel count must be a multiple of 64 and the channels per group must
divide 64 — true for the usual 32x4d / 32x8d / 64x4d settings.

Reference quirk kept for state_dict / behaviour parity: ``_make_resX_layer`` builds the downsample norm WITHOUT
``use_gn`` (resnext.py:147), so a ``use_gn=True`` ResNeXt still has BatchNorm in its downsample branches — and leaves
them in training mode (resnext.py:303-310 only switches BN to eval when ``use_gn`` is False).  Those branches then run
with batch statistics (``tdn_bn_train_fwd/bwd``) like any training-mode BatchNorm2d on this path.
"""
import math

import torch.nn as nn

from ..layers import conv1x1_group, conv3x3_group, conv7x7_group, norm_layer
from ..registry import BACKBONES
from .resnet import ResNet, _ResBlock


class ResNeXtBasicBlock(_ResBlock):
    """3x3(stride) -> norm -> ReLU -> grouped 3x3 -> norm, + residual, ReLU (resnext.py:12-66)."""
    expansion = 1
    _kind = 'basic'

    def __init__(self, inplanes, planes, cardinality, stride=1, dilation=1, use_gn=False, downsample=None):
        super(ResNeXtBasicBlock, self).__init__()
        self.conv1 = conv3x3_group(inplanes, planes * self.expansion, stride, dilation)
        self.conv2 = conv3x3_group(planes * self.expansion, planes * self.expansion, groups=cardinality)
        self.norm_names = ['bn1', 'bn2'] if not use_gn else ['gn1', 'gn2']
        for name in self.norm_names:
            self.add_module(name, norm_layer(planes * self.expansion, use_gn))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.cardinality = cardinality
        self.stride = stride
        self.dilation = dilation
        self.use_gn = use_gn


class ResNeXtBottleneck(_ResBlock):
    """1x1 -> norm -> ReLU -> grouped 3x3(stride) -> norm -> ReLU -> 1x1 -> norm, + residual, ReLU
    (resnext.py:69-137); D = floor(planes * base_width / 64) channels per group, C = cardinality groups."""
    expansion = 4
    _kind = 'bottleneck'

    def __init__(self, inplanes, planes, base_width, cardinality, stride=1, dilation=1, use_gn=False,
                 downsample=None):
        super(ResNeXtBottleneck, self).__init__()
        D = int(math.floor(planes * (base_width / 64.)))
        C = cardinality
        self.conv1 = conv1x1_group(inplanes, D * C, stride=1)
        self.conv2 = conv3x3_group(D * C, D * C, stride=stride, dilation=dilation, groups=C)
        self.conv3 = conv1x1_group(D * C, planes * self.expansion, stride=1)
        self.norm_names = ['bn1', 'bn2', 'bn3'] if not use_gn else ['gn1', 'gn2', 'gn3']
        for name, c in zip(self.norm_names, [D * C, D * C, planes * self.expansion]):
            self.add_module(name, norm_layer(c, use_gn))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.base_width = base_width
        self.cardinality = cardinality
        self.stride = stride
        self.dilation = dilation
        self.use_gn = use_gn


def _make_resX_layer(block, inplanes, planes, blocks, base_width, cardinality, stride=1, dilation=1, use_gn=False):
    """One stage (resnext.py:134-170); the downsample norm is BatchNorm whatever ``use_gn`` says (resnext.py:147)."""
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(conv1x1_group(inplanes, planes * block.expansion, stride=stride),
                                   norm_layer(planes * block.expansion))

    def make(inpl, s, down):
        if block is ResNeXtBasicBlock:   # the reference passes base_width positionally to both block types; the basic
            # block's signature has no base_width, so there it lands in `cardinality` and cardinality in `stride`
            # (resnext.py:151-159) — ResNeXt-18/34 therefore only construct in the reference when that happens to be
            # consistent; mirrored here by passing the same positional arguments
            return block(inpl, planes, base_width, cardinality, stride=s, dilation=dilation, use_gn=use_gn,
                         downsample=down)
        return block(inpl, planes, base_width, cardinality, stride=s, dilation=dilation, use_gn=use_gn,
                     downsample=down)

    layers = [make(inplanes, stride, downsample)]
    inplanes = planes * block.expansion
    for _ in range(1, blocks):
        layers.append(make(inplanes, 1, None))
    return nn.Sequential(*layers)


@BACKBONES.register_module
class ResNeXt(ResNet):
    """ResNeXt backbone (resnext.py:173-330): ``ResNeXt(depth, base_width, cardinality, ...)``; forward, ``train()``
    and ``init_weights`` semantics are those of :class:`ResNet`."""

    arch_settings = {
        18: (ResNeXtBasicBlock, (2, 2, 2, 2)),
        34: (ResNeXtBasicBlock, (3, 4, 6, 3)),
        50: (ResNeXtBottleneck, (3, 4, 6, 3)),
        101: (ResNeXtBottleneck, (3, 4, 23, 3)),
        152: (ResNeXtBottleneck, (3, 8, 36, 3)),
    }

    def __init__(self, depth, base_width, cardinality, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_gn=False, bn_eval=True, bn_frozen=False):
        nn.Module.__init__(self)
        if depth not in self.arch_settings:
            raise KeyError('invalid depth {} for resnet'.format(depth))
        assert 1 <= num_stages <= 4
        block, stage_blocks = self.arch_settings[depth]
        stage_blocks = stage_blocks[:num_stages]
        assert len(strides) == len(dilations) == num_stages
        assert max(out_indices) < num_stages

        self.depth = depth
        self.base_width = base_width
        self.cardinality = cardinality
        self.out_indices = out_indices
        self.frozen_stages = frozen_stages
        if not use_gn:
            self.bn_eval = bn_eval
            self.bn_frozen = bn_frozen
        self.use_gn = use_gn
        self.dilations = tuple(dilations)

        self.inplanes = 64
        self.conv1 = conv7x7_group(3, 64, stride=2)
        self.norm_name = 'bn1' if not use_gn else 'gn1'
        self.add_module(self.norm_name, norm_layer(64, use_gn))
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

        self.resX_layers = []
        for i, num_blocks in enumerate(stage_blocks):
            planes = 64 * 2 ** i
            layer = _make_resX_layer(block, self.inplanes, planes, num_blocks, base_width, cardinality,
                                     stride=strides[i], dilation=dilations[i], use_gn=use_gn)
            self.inplanes = planes * block.expansion
            layer_name = 'layer{}'.format(i + 1)
            self.add_module(layer_name, layer)
            self.resX_layers.append(layer_name)
        self.res_layers = self.resX_layers   # the name ResNet's shared machinery uses
        self.feat_dim = block.expansion * 64 * 2 ** (len(stage_blocks) - 1)
