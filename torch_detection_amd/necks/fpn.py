"""Feature Pyramid Network neck — drop-in for the reference's models/necks/fpn.py on the MI355X HIP path.

Same constructor (fpn.py:11-19), attributes (``lateral_convs[i].conv``, ``fpn_convs[i].conv`` ...), state_dict
keys and assertions.  ``forward`` runs the whole neck as one autograd node (functional.FPNFunction): the
top-down ``laterals[i-1] += F.interpolate(laterals[i], scale_factor=2)`` (fpn.py:99-101) is fused into the
lateral 1x1 conv's epilogue and its adjoint (2x2 sum-pool) into the 3x3 dgrad epilogue.

As in the reference, every level must be exactly 2x the next one, otherwise RuntimeError (fpn.py:100 raises
the same type): inputs must come from a batch padded to a multiple of 32 (datasets/utils/image.py:326-347).
"""
import torch
import torch.nn as nn

from .. import functional as HF
from ..inits import constant_init, xavier_init
from ..layers import ConvModule
from ..registry import NECKS


@NECKS.register_module
class FPN(nn.Module):

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 normalize=None, use_gn=False):
        super(FPN, self).__init__()
        assert isinstance(in_channels, list)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_ins = len(in_channels)
        self.num_outs = num_outs
        self.with_bias = normalize is None

        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            # if end_level < inputs, no extra level is allowed
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels)
            assert num_outs == end_level - start_level
        self.start_level = start_level
        self.end_level = end_level
        self.add_extra_convs = add_extra_convs

        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        common = dict(normalize=normalize, bias=self.with_bias, use_gn=use_gn)
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, kernel_size=1, **common))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, kernel_size=3, padding=1, **common))

        # extra stride-2 conv levels (RetinaNet style, fpn.py:63-78)
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                cin = self.in_channels[self.backbone_end_level - 1] if i == 0 else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, kernel_size=3, stride=2, padding=1, **common))

    def init_weights(self):
        """xavier-uniform convs (bias 0), norm weight 1 (fpn.py:80-86)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')
            if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                constant_init(m, 1)

    def hip_net(self, dtype=None):
        if dtype is None:
            dtype = getattr(self, 'compute_dtype', torch.bfloat16)
        with HF.batched_refresh():
            lat = [m.hip_unit(dtype) for m in self.lateral_convs]
            fpn = [m.hip_unit(dtype) for m in self.fpn_convs]
        return HF.FPNNet(lat, fpn, self.start_level, self.backbone_end_level, self.num_outs,
                         self.add_extra_convs, self.num_ins)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        net = self.hip_net(HF.pick_dtype(self, inputs))
        return HF.FPNFunction.apply(net, *(tuple(inputs) + tuple(net.params())))
