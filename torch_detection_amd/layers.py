"""Layer builders mirroring the reference's models/utils/layers.py (conv/norm factories, ConvModule).

The builders return ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` objects so parameter names, shapes and
state_dict keys are identical to the reference (layers.py:6-54) and checkpoints interchange.  On the hot
path those modules are parameter holders only: ``ResNet`` / ``FPN`` / ``ConvModule`` run their arithmetic
through the fused HIP kernels (functional.py), never through ``nn.Conv2d.forward``.

3x3 (and larger) conv weights are kept in ``channels_last`` memory format: the logical OIHW shape and the
state_dict are unchanged, but the bytes are [O][kh][kw][I] — the K-major order the MFMA kernels consume and
the order the weight-gradient kernel writes, so gradients obey autograd's layout contract without a copy.
"""
import warnings

import torch
import torch.nn as nn

from . import functional as HF


def _channels_last_(conv):
    if conv.weight.shape[2] * conv.weight.shape[3] > 1 and (conv.weight.shape[1] % 64 == 0 or conv.groups > 1):
        conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    return conv


def conv1x1_group(in_planes, out_planes, stride=1, groups=1):
    """1x1 convolution without bias (layers.py:6-17)."""
    return nn.Conv2d(in_channels=in_planes, out_channels=out_planes, kernel_size=1, stride=stride,
                     groups=groups, bias=False)


def conv3x3_group(in_planes, out_planes, stride=1, dilation=1, groups=1):
    """3x3 convolution, padding = dilation, without bias (layers.py:20-32)."""
    return _channels_last_(nn.Conv2d(in_channels=in_planes, out_channels=out_planes, kernel_size=3,
                                     stride=stride, padding=dilation, dilation=dilation, groups=groups,
                                     bias=False))


def conv7x7_group(in_planes, out_planes, stride=1, groups=1):
    """7x7 convolution, padding 3, without bias (layers.py:35-47)."""
    return nn.Conv2d(in_channels=in_planes, out_channels=out_planes, kernel_size=7, stride=stride, padding=3,
                     dilation=1, groups=groups, bias=False)


def get_group_gn(planes):
    """Number of GroupNorm groups (layers.py:138-154): 32 groups."""
    num_groups = 32
    assert planes % num_groups == 0
    return num_groups


def norm_layer(planes, use_gn=False):
    """BatchNorm2d, or GroupNorm(32, planes) (layers.py:50-54)."""
    if not use_gn:
        return nn.BatchNorm2d(planes)
    return nn.GroupNorm(get_group_gn(planes), planes)


class ConvModule(nn.Module):
    """conv (+bias) [+ BN] [+ ReLU] — the reference's ConvModule (layers.py:57-135).

    Same constructor, attributes (``conv``, ``norm``, ``activate``, ``with_norm`` ...) and state_dict keys.
    All five layer types of the reference's docstring are on the HIP path: conv, conv + BN/GN, conv + BN/GN + ReLU,
    conv + ReLU, and (``activate_last=False``) BN/GN + ReLU + conv; ``activation`` 'relu' or 'relu6'; BatchNorm2d in
    eval mode (folded) or training mode (batch statistics), GroupNorm via ``use_gn=True``.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, normalize=None, use_gn=False, activation=None, activate_last=True):
        super(ConvModule, self).__init__()
        self.with_norm = normalize is not None
        self.with_activation = activation is not None
        self.with_bias = bias
        self.activation = activation
        self.activate_last = activate_last
        self.use_gn = use_gn

        if self.with_norm and self.with_bias:
            warnings.warn('ConvModule has norm and bias at the same time')

        self.conv = _channels_last_(nn.Conv2d(in_channels=in_channels, out_channels=out_channels,
                                              kernel_size=kernel_size, stride=stride, padding=padding,
                                              dilation=dilation, groups=groups, bias=bias))
        self.in_channels = self.conv.in_channels
        self.out_channels = self.conv.out_channels
        self.kernel_size = self.conv.kernel_size
        self.stride = self.conv.stride
        self.padding = self.conv.padding
        self.dilation = self.conv.dilation
        self.groups = self.conv.groups

        if self.with_norm:
            norm_channels = out_channels if self.activate_last else in_channels
            self.norm = norm_layer(norm_channels, use_gn=use_gn)

        if self.with_activation:
            assert activation in ['relu', 'relu6'], 'Only ReLU and ReLU6 are supported'
            if self.activation == 'relu':
                self.activate = nn.ReLU(inplace=True)
            elif self.activation == 'relu6':
                self.activate = nn.ReLU6(inplace=True)

    def hip_unit(self, dtype=torch.bfloat16):
        """Prepared (cached) fused unit of this module: a ``ConvUnit`` (conv -> [norm] -> [ReLU | ReLU6]) or, for
        ``activate_last=False``, a ``PreActUnit`` ([norm] -> [ReLU | ReLU6] -> conv)."""
        if self.with_norm and not isinstance(self.norm, (nn.BatchNorm2d, nn.GroupNorm)):
            raise NotImplementedError('ConvModule norm %s is not on the HIP path' % type(self.norm).__name__)
        if not self.activate_last:
            conv = HF.prepare_unit(self, 'conv', self.conv, None, False, dtype)
            act = 0 if not self.with_activation else (2 if self.activation == 'relu6' else 1)
            return HF.PreActUnit(conv, self.norm if self.with_norm else None, act)
        unit = HF.prepare_unit(self, 'conv', self.conv, self.norm if self.with_norm else None,
                               self.with_activation, dtype)
        unit.act6 = self.with_activation and self.activation == 'relu6'
        return unit

    def forward(self, x):
        unit = self.hip_unit(HF.pick_dtype(self, x))
        if isinstance(unit, HF.PreActUnit):
            return HF.PreActConvFunction.apply(unit, x, *unit.params())
        return HF.ConvUnitFunction.apply(unit, x, *unit.params())
