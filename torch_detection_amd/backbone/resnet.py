"""ResNet-18/34/50/101/152 backbone — drop-in for the reference's models/backbone/resnet.py on the MI355X HIP path.

Constructor signature, attribute names (``conv1, bn1, relu, maxpool, layer1..4, res_layers, feat_dim,
out_indices``), block attributes (``conv1..3, bn1..3, downsample``), state_dict keys/shapes and error
behaviour follow resnet.py:9-294.  The arithmetic does not: ``forward`` runs the whole backbone as ONE
autograd node over fused implicit-GEMM MFMA kernels (functional.SeqNetFunction); the ``nn.Conv2d`` /
``nn.BatchNorm2d`` children only hold parameters.

Documented deviations:
  * ``train()`` returns ``self`` (the reference returns None, resnet.py:270-294);
  * ``frozen_stages >= 0`` implements the evident intent instead of raising AttributeError (resnet.py:288);
  * activations are bfloat16, outputs are bfloat16 NCHW-shaped tensors with channels_last strides;
  * the input image gets no gradient (nothing in the reference consumes one);
  * ``use_gn=True``, ``dilations`` and ``bn_eval=False`` (BatchNorm with batch statistics) all run on the HIP path.
There is no CPU fallback: ``forward`` on a CPU tensor raises RuntimeError.
"""
import logging

import torch
import torch.nn as nn

from .. import functional as HF
from ..checkpoint import load_checkpoint
from ..inits import constant_init, kaiming_init
from ..layers import conv1x1_group, conv3x3_group, conv7x7_group, norm_layer
from ..registry import BACKBONES


class _ResBlock(nn.Module):
    """Shared plumbing of BasicBlock / Bottleneck: parameter holders + the fused HIP schedule."""
    expansion = 1
    _kind = 'basic'

    def _norms(self):
        return [getattr(self, n) for n in self.norm_names]

    def hip_spec(self, dtype=torch.bfloat16):
        norms = self._norms()
        convs = [self.conv1, self.conv2] + ([self.conv3] if self._kind == 'bottleneck' else [])
        us = [HF.prepare_unit(self, 'u%d' % i, c, n, False, dtype) for i, (c, n) in enumerate(zip(convs, norms))]
        ud = None
        if self.downsample is not None:
            ud = HF.prepare_unit(self, 'ud', self.downsample[0], self.downsample[1], False, dtype)
        u3 = us[2] if self._kind == 'bottleneck' else None
        return HF.BlockSpec(self._kind, us[0], us[1], u3, ud, self.stride)

    def forward(self, x):
        with HF.batched_refresh():
            net = HF.SeqNet(None, [self.hip_spec(HF.pick_dtype(self, x))], [0])
        return HF.SeqNetFunction.apply(net, x, *net.params())[0]


class BasicBlock(_ResBlock):
    """3x3(stride) -> BN -> ReLU -> 3x3 -> BN, + residual, ReLU (resnet.py:9-59)."""
    expansion = 1
    _kind = 'basic'

    def __init__(self, inplanes, planes, stride=1, dilation=1, use_gn=False, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = conv3x3_group(inplanes, planes * self.expansion, stride, dilation)
        self.conv2 = conv3x3_group(planes * self.expansion, planes * self.expansion)
        self.norm_names = ['bn1', 'bn2'] if not use_gn else ['gn1', 'gn2']
        for name in self.norm_names:
            self.add_module(name, norm_layer(planes * self.expansion, use_gn))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation
        self.use_gn = use_gn


class Bottleneck(_ResBlock):
    """1x1 -> BN -> ReLU -> 3x3(stride) -> BN -> ReLU -> 1x1(x4) -> BN, + residual, ReLU (resnet.py:62-119)."""
    expansion = 4
    _kind = 'bottleneck'

    def __init__(self, inplanes, planes, stride=1, dilation=1, use_gn=False, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = conv1x1_group(inplanes, planes)
        self.conv2 = conv3x3_group(planes, planes, stride=stride, dilation=dilation)
        self.conv3 = conv1x1_group(planes, planes * self.expansion)
        self.norm_names = ['bn1', 'bn2', 'bn3'] if not use_gn else ['gn1', 'gn2', 'gn3']
        widths = [planes, planes, planes * self.expansion]
        for name, c in zip(self.norm_names, widths):
            self.add_module(name, norm_layer(c, use_gn))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation
        self.use_gn = use_gn


def _make_res_layer(block, inplanes, planes, blocks, stride=1, dilation=1, use_gn=False):
    """One stage; the first block gets a 1x1(stride)+norm downsample when shape changes (resnet.py:122-155)."""
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(
            conv1x1_group(inplanes, planes * block.expansion, stride=stride),
            norm_layer(planes * block.expansion, use_gn=use_gn))
    layers = [block(inplanes, planes, stride=stride, dilation=dilation, use_gn=use_gn, downsample=downsample)]
    inplanes = planes * block.expansion
    for _ in range(1, blocks):
        layers.append(block(inplanes, planes, stride=1, dilation=dilation, use_gn=use_gn))
    return nn.Sequential(*layers)


@BACKBONES.register_module
class ResNet(nn.Module):
    """ResNet backbone (resnet.py:158-294).

    Args:
        depth (int): one of {18, 34, 50, 101, 152}; anything else raises KeyError.
        num_stages (int): 1..4.
        strides / dilations (Sequence[int]): per stage.
        out_indices (Sequence[int]): stages whose output is returned (a bare tensor if only one).
        frozen_stages (int): stem + stages [1..frozen_stages] get requires_grad=False in train mode.
        use_gn (bool): GroupNorm(32 groups) instead of BatchNorm.
        bn_eval (bool): keep BN layers in eval mode (running statistics) while training — reference default.
        bn_frozen (bool): also freeze BN weight / bias.
    """

    arch_settings = {
        18: (BasicBlock, (2, 2, 2, 2)),
        34: (BasicBlock, (3, 4, 6, 3)),
        50: (Bottleneck, (3, 4, 6, 3)),
        101: (Bottleneck, (3, 4, 23, 3)),
        152: (Bottleneck, (3, 8, 36, 3)),
    }

    def __init__(self, depth, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_gn=False, bn_eval=True, bn_frozen=False):
        super(ResNet, self).__init__()
        if depth not in self.arch_settings:
            raise KeyError('invalid depth {} for resnet'.format(depth))
        assert 1 <= num_stages <= 4
        block, stage_blocks = self.arch_settings[depth]
        stage_blocks = stage_blocks[:num_stages]
        assert len(strides) == len(dilations) == num_stages
        assert max(out_indices) < num_stages

        self.depth = depth
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

        self.res_layers = []
        for i, num_blocks in enumerate(stage_blocks):
            planes = 64 * 2 ** i
            res_layer = _make_res_layer(block, self.inplanes, planes, num_blocks, stride=strides[i],
                                        dilation=dilations[i], use_gn=use_gn)
            self.inplanes = planes * block.expansion
            layer_name = 'layer{}'.format(i + 1)
            self.add_module(layer_name, res_layer)
            self.res_layers.append(layer_name)
        self.feat_dim = block.expansion * 64 * 2 ** (len(stage_blocks) - 1)

    def init_weights(self, pretrained=None):
        """``None``: kaiming-normal(fan_out) convs, BN/GN weight 1 (resnet.py:240-251); ``str``: local checkpoint."""
        if isinstance(pretrained, str):
            load_checkpoint(self, pretrained, strict=False, logger=logging.getLogger())
        elif pretrained is None:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    kaiming_init(m)
                elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                    constant_init(m, 1)
        else:
            raise TypeError('pretrained must be a str or None')

    # ---- HIP schedule -------------------------------------------------------------------------------
    def hip_net(self, dtype=None):
        """Prepared whole-backbone program (units are cached on the blocks; only changed weights are re-packed).
        ``dtype``: torch.bfloat16 / torch.float16 operands; default = ``self.compute_dtype`` (bfloat16)."""
        if dtype is None:
            dtype = getattr(self, 'compute_dtype', torch.bfloat16)
        blocks, out_blocks = [], []
        with HF.batched_refresh():     # one grouped fold + pack launch per 30 convs instead of two launches per conv
            stem = HF.prepare_unit(self, 'stem', self.conv1, getattr(self, self.norm_name), True, dtype)
            for i, layer_name in enumerate(self.res_layers):
                for blk in getattr(self, layer_name):
                    blocks.append(blk.hip_spec(dtype))
                if i in self.out_indices:
                    out_blocks.append(len(blocks) - 1)
        return HF.SeqNet(stem, blocks, out_blocks)

    def forward(self, x):
        net = self.hip_net(HF.pick_dtype(self, x))
        outs = HF.SeqNetFunction.apply(net, x, *net.params())
        return outs[0] if len(outs) == 1 else tuple(outs)

    def train(self, mode=True):
        super(ResNet, self).train(mode)
        if not self.use_gn and self.bn_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
                    if self.bn_frozen:
                        for p in m.parameters():
                            p.requires_grad = False
        if mode and self.frozen_stages >= 0:
            norm1 = getattr(self, self.norm_name)
            norm1.eval()
            for p in list(self.conv1.parameters()) + list(norm1.parameters()):
                p.requires_grad = False
            for i in range(1, self.frozen_stages + 1):
                mod = getattr(self, 'layer{}'.format(i))
                mod.eval()
                for p in mod.parameters():
                    p.requires_grad = False
        return self
