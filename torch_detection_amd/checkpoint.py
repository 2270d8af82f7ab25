"""Checkpoint I/O with the reference's semantics (models/utils/checkpoint.py:11-169), local files only.

Key names / tensor shapes of ``ResNet`` and ``FPN`` are identical to the reference, so files written by either
side load into the other.  ``modelzoo://`` / ``http(s)://`` sources (checkpoint.py:87-99) need network access
and are rejected with a clear error instead of being fetched.
"""
import logging
import os
import time
from collections import OrderedDict

import torch


def load_state_dict(module, state_dict, strict=False, logger=None):
    """Non-strict load by key name (the behaviour of checkpoint.py:11-64), built on PyTorch's own loader: tensors whose
    shapes disagree raise ``RuntimeError`` (``nn.Module.load_state_dict`` reports them even when not strict); keys
    only one side has are reported — raised with ``strict``, else logged / printed."""
    report = module.load_state_dict(state_dict, strict=False)
    lines = []
    if report.unexpected_keys:
        lines.append('unexpected key in source state_dict: ' + ', '.join(report.unexpected_keys))
    if report.missing_keys:
        lines.append('missing keys in source state_dict: ' + ', '.join(sorted(report.missing_keys)))
    if not lines:
        return
    text = '\n'.join(lines)
    if strict:
        raise RuntimeError(text)
    (logger.warning if logger is not None else print)(text)


def load_checkpoint(model, filename, map_location=None, strict=False, logger=None):
    """Load a local checkpoint file (checkpoint.py:67-120): accepts a raw OrderedDict or {'state_dict': ...},
    strips a leading ``module.`` prefix."""
    if filename.startswith(('modelzoo://', 'http://', 'https://')):
        raise IOError('{}: remote checkpoints are not supported (no network); download the file and pass its '
                      'local path'.format(filename))
    if not os.path.isfile(filename):
        raise IOError('{} is not a checkpoint file'.format(filename))
    checkpoint = torch.load(filename, map_location=map_location, weights_only=True)
    if isinstance(checkpoint, OrderedDict):
        state_dict = checkpoint
    elif isinstance(checkpoint, dict) and 'state_dict' in checkpoint:
        state_dict = checkpoint['state_dict']
    else:
        raise RuntimeError('No state_dict found in checkpoint file {}'.format(filename))
    if list(state_dict.keys())[0].startswith('module.'):
        state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
    target = model.module if hasattr(model, 'module') else model
    load_state_dict(target, state_dict, strict, logger)
    return checkpoint


def weights_to_cpu(state_dict):
    """Contiguous CPU copy of a state_dict (checkpoint.py:123-135); drops the channels_last strides so the
    file is byte-identical in layout to one written by the reference."""
    out = OrderedDict()
    for key, val in state_dict.items():
        out[key] = val.detach().cpu().contiguous()
    return out


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """Write {'meta', 'state_dict' (cpu), 'optimizer'?} (checkpoint.py:138-169)."""
    if meta is None:
        meta = {}
    elif not isinstance(meta, dict):
        raise TypeError('meta must be a dict or None, but got {}'.format(type(meta)))
    meta.update(time=time.asctime())
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    target = model.module if hasattr(model, 'module') else model
    checkpoint = {'meta': meta, 'state_dict': weights_to_cpu(target.state_dict())}
    if optimizer is not None:
        checkpoint['optimizer'] = optimizer.state_dict()
    torch.save(checkpoint, filename)


def get_logger():
    return logging.getLogger()
