"""Host-side weight initialisers with the reference's names and argument meaning
(models/utils/inits.py:5-52).  They only drive ``torch.nn.init`` — no kernel is involved — and are built
from one helper so every initialiser treats an optional ``bias`` the same way."""
import math

from torch.nn import init as _init


def _apply(module, weight_fn, bias):
    weight_fn(module.weight)
    b = getattr(module, 'bias', None)
    if b is not None:
        _init.constant_(b, bias)


def _pick(distribution, uniform_fn, normal_fn):
    if distribution not in ('uniform', 'normal'):
        raise AssertionError("distribution must be 'uniform' or 'normal', got %r" % (distribution,))
    return uniform_fn if distribution == 'uniform' else normal_fn


def constant_init(module, val, bias=0):
    _apply(module, lambda w: _init.constant_(w, val), bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    fn = _pick(distribution, _init.xavier_uniform_, _init.xavier_normal_)
    _apply(module, lambda w: fn(w, gain=gain), bias)


def normal_init(module, mean=0, std=1, bias=0):
    _apply(module, lambda w: _init.normal_(w, mean, std), bias)


def uniform_init(module, a=0, b=1, bias=0):
    _apply(module, lambda w: _init.uniform_(w, a, b), bias)


def kaiming_init(module, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    fn = _pick(distribution, _init.kaiming_uniform_, _init.kaiming_normal_)
    _apply(module, lambda w: fn(w, mode=mode, nonlinearity=nonlinearity), bias)


def bias_init_with_prob(prior_prob):
    """Bias b with sigmoid(b) == prior_prob."""
    return float(-math.log((1.0 - prior_prob) / prior_prob))
