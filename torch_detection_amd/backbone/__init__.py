from .resnet import BasicBlock, Bottleneck, ResNet
from .resnext import ResNeXt, ResNeXtBasicBlock, ResNeXtBottleneck

__all__ = ['ResNet', 'BasicBlock', 'Bottleneck', 'ResNeXt', 'ResNeXtBasicBlock', 'ResNeXtBottleneck']
