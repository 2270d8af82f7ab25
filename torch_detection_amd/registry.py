"""Module registry — same behaviour as the reference's models/registry.py:4-41.

``Registry(name)`` keeps a ``module_dict`` name -> class; ``register_module`` is a class decorator that
returns the class, raises ``TypeError`` for non-``nn.Module`` classes (registry.py:25-28) and ``KeyError``
for a duplicate name (registry.py:30-32).  ``BACKBONES`` / ``NECKS`` are the two registries of the hot path
(registry.py:40-41).
"""
import torch.nn as nn


class Registry(object):

    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def _register_module(self, module_class):
        if not (isinstance(module_class, type) and issubclass(module_class, nn.Module)):
            raise TypeError('module must be a child of nn.Module, but got {}'.format(type(module_class)))
        module_name = module_class.__name__
        if module_name in self._module_dict:
            raise KeyError('{} is already registered in {}'.format(module_name, self.name))
        self._module_dict[module_name] = module_class

    def register_module(self, cls):
        self._register_module(cls)
        return cls

    def build(self, cfg):
        """Convenience not present in the reference: ``{'type': name, **kwargs}`` -> instance."""
        cfg = dict(cfg)
        return self._module_dict[cfg.pop('type')](**cfg)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
