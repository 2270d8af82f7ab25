"""Deterministic synthetic parameters / inputs shared by oracle/gen_golden.py and the tests.

Values come from an integer hash evaluated with numpy uint64 arithmetic (not from any RNG stream), so the
fixtures under tests/golden/ stay valid across torch / numpy versions.  Everything is rounded to
bf16-representable float32 so the HIP path and the fp32 oracle see identical operand values.
"""
import math

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _hash_u01(n, seed):
    """n floats in [0, 1) from a splitmix64-style hash of (index, seed)."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def det_tensor(shape, seed, lo=-1.0, hi=1.0, bf16=True):
    n = int(np.prod(shape))
    u = _hash_u01(n, seed)
    t = torch.from_numpy(u * np.float32(hi - lo) + np.float32(lo)).reshape(shape)
    if bf16:
        t = t.bfloat16().float()
    return t.contiguous()


def _name_seed(name, base):
    h = base
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000003
    return h


def fill_state_dict(sd, base_seed):
    """Overwrite every tensor of a ResNet / FPN state_dict with deterministic, non-trivial values (in place).

    conv weights ~ U(-a, a), a = sqrt(6 / fan_out)  (kaiming-uniform-like, keeps activations O(1));
    BN: gamma in [0.5, 1.5], beta in [-0.1, 0.1], running_mean in [-0.1, 0.1], running_var in [0.5, 1.5];
    biases in [-0.1, 0.1].  Non-trivial BN stats matter: fresh 0/1 stats hide bugs.
    """
    out = {}
    for k, v in sd.items():
        s = _name_seed(k, base_seed)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif v.dim() == 4:
            fan_out = v.shape[0] * v.shape[2] * v.shape[3]
            a = math.sqrt(6.0 / fan_out)
            out[k] = det_tensor(tuple(v.shape), s, -a, a)
        elif k.endswith("running_var"):
            out[k] = det_tensor(tuple(v.shape), s, 0.5, 1.5)
        elif k.endswith("running_mean"):
            out[k] = det_tensor(tuple(v.shape), s, -0.1, 0.1)
        elif k.endswith(".weight"):  # BN gamma
            out[k] = det_tensor(tuple(v.shape), s, 0.5, 1.5)
        else:  # BN beta / conv bias
            out[k] = det_tensor(tuple(v.shape), s, -0.1, 0.1)
    return out


def rel_l2(a, b):
    a = a.detach().double().flatten()
    b = b.detach().double().flatten()
    d = (a - b).norm()
    n = b.norm()
    return float(d / n) if n > 0 else float(d)


def max_rel(a, b):
    """max |a-b| / max |b| — the '1e-3 relative' figure used for conv activations."""
    a = a.detach().double()
    b = b.detach().double()
    den = float(b.abs().max())
    return float((a - b).abs().max()) / (den if den > 0 else 1.0)
