"""Path-Aggregation FPN neck — drop-in for the reference's models/necks/pafpn.py:8-148.

FPN (laterals, top-down add, 3x3 output convs) followed by the bottom-up path
``N_i = pa_convs2[i-1](P_i + pa_convs1[i-1](N_{i-1}))`` (pafpn.py:127-131).  Same constructor, attribute names and
state_dict keys as the reference; the arithmetic reuses the fused conv kernels (the ``P_i +`` rides in the stride-2
conv's epilogue).  The forward is two autograd nodes: functional.FPNFunction and functional.PAPathFunction.
"""
import torch.nn as nn

from .. import functional as HF
from ..inits import constant_init, xavier_init
from ..layers import ConvModule
from ..registry import NECKS


@NECKS.register_module
class PAFPN(nn.Module):

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 normalize=None, use_gn=False, activation=None):
        super(PAFPN, self).__init__()
        assert isinstance(in_channels, list)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_ins = len(in_channels)
        self.num_outs = num_outs
        self.with_bias = normalize is None
        self.activation = activation

        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.backbone_end_level - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= self.num_ins
            assert num_outs == end_level - start_level
        self.start_level = start_level
        self.end_level = end_level
        self.add_extra_convs = add_extra_convs

        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        self.pa_convs1 = nn.ModuleList()
        self.pa_convs2 = nn.ModuleList()
        common = dict(bias=self.with_bias, normalize=normalize, use_gn=use_gn)
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, kernel_size=1, **common))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, kernel_size=3, padding=1, **common))
            if i < self.backbone_end_level - 1:
                self.pa_convs1.append(ConvModule(out_channels, out_channels, kernel_size=3, stride=2, padding=1,
                                                 activation=activation, **common))
                self.pa_convs2.append(ConvModule(out_channels, out_channels, kernel_size=3, padding=1,
                                                 activation=activation, **common))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                cin = self.in_channels[self.backbone_end_level - 1] if i == 0 else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, kernel_size=3, stride=2, padding=1, **common))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')
            if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                constant_init(m, 1)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        nlat = len(self.lateral_convs)
        dtype = HF.pick_dtype(self, inputs)
        lat = [m.hip_unit(dtype) for m in self.lateral_convs]
        extra_convs = self.add_extra_convs and self.num_outs > nlat
        # the extra stride-2 conv levels (pafpn.py:139-147) hang off the last backbone input exactly as in FPN
        # (fpn.py:118-124): the FPN node computes them, the PA path then rebuilds the first nlat levels
        fpn = [m.hip_unit(dtype) for m in (self.fpn_convs if extra_convs else self.fpn_convs[:nlat])]
        fnet = HF.FPNNet(lat, fpn, self.start_level, self.backbone_end_level, self.num_outs if extra_convs else nlat,
                         extra_convs, self.num_ins)
        P = HF.FPNFunction.apply(fnet, *(tuple(inputs) + tuple(fnet.params())))
        pnet = HF.PAPathNet([m.hip_unit(dtype) for m in self.pa_convs1], [m.hip_unit(dtype) for m in self.pa_convs2],
                            0 if extra_convs else self.num_outs - nlat)
        outs = HF.PAPathFunction.apply(pnet, *(tuple(P[:nlat]) + tuple(pnet.params())))
        return tuple(outs) + tuple(P[nlat:]) if extra_convs else outs
