from .fpn import FPN
from .pafpn import PAFPN

__all__ = ['FPN', 'PAFPN']
