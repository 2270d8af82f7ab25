"""oracle/sched_ref.py — CPU restatement of the *fused schedule* the HIP path runs (TEST INFRASTRUCTURE ONLY).

torch_ref.py restates the reference's forward with torch autograd doing the backward in fp32.  The HIP path
(torch_detection_amd/functional.py) instead runs an explicit backward schedule and stores activations and
activation-gradients in bf16.  Gradients through ~50 ReLU masks are discontinuous in the activations, so the
fp32 autograd result and a bf16-activation result legitimately differ by far more than rounding (mask flips).
This module therefore restates the same schedule with plain torch CPU ops:

  * ``quant=False``: no rounding anywhere -> must equal torch_ref's autograd gradients to fp32 accuracy.  That
    pins the schedule algebra (BN-gamma gradient via the weight-space identity, scale folded into the dgrad
    weights, residual / stage / FPN gradient routing, max-pool tie rule) to the reference semantics on CPU
    (tests/test_oracle_sched.py).
  * ``quant=True``: bf16 rounding at exactly the points where the HIP kernels store bf16 -> the GPU result must
    agree to ~1e-3 (tests/test_gpu_models.py), which checks every kernel launch of the real schedule in situ.

Reference call sites restated: resnet.py:42-59,97-119,253-268; fpn.py:88-125; layers.py:50-54,122-135.
"""
import torch
import torch.nn.functional as F

from .torch_ref import ARCH, BN_EPS


def rnd(t, quant):
    """Round to the storage type of the HIP path: ``quant`` is False (no rounding), True (bfloat16) or a torch
    16-bit float dtype (torch.bfloat16 / torch.float16)."""
    if not quant:
        return t
    return t.to(torch.bfloat16 if quant is True else quant).float()


class Unit(object):
    def __init__(self, w, bn=None, bias=None, stride=1, quant=True):
        """w: OIHW fp32; bn: (gamma, beta, running_mean, running_var) or None; bias: tensor or None."""
        self.w = w
        self.k = w.shape[2]
        self.stride = stride
        self.pad = self.k // 2
        self.wb = rnd(w, quant)
        if bn is not None:
            gamma, beta, mean, var = bn
            self.invstd = 1.0 / torch.sqrt(var + BN_EPS)
            self.scale = gamma * self.invstd
            self.shift = beta - mean * self.scale
            self.mean = mean
        else:
            self.invstd = self.mean = None
            self.scale = None
            self.shift = bias
        sc = self.scale.view(-1, 1, 1, 1) if self.scale is not None else 1.0
        self.wd = rnd(self.wb * sc, quant)       # dgrad operand: bf16(scale * bf16(w))
        self.quant = quant

    def fwd(self, x, addend=None, mode=None, relu=False):
        v = F.conv2d(x, self.wb, None, self.stride, self.pad)
        if self.scale is not None:
            v = v * self.scale.view(1, -1, 1, 1)
        if self.shift is not None:
            v = v + self.shift.view(1, -1, 1, 1)
        if addend is not None:
            v = v + (addend if mode == 'same' else F.interpolate(addend, scale_factor=2, mode='nearest'))
        if relu:
            v = F.relu(v)
        return rnd(v, self.quant)

    def dgrad(self, g, in_hw, addend=None, mode=None, mask_src=None):
        H, W = in_hw
        oph = H - ((g.shape[2] - 1) * self.stride - 2 * self.pad + self.k)
        opw = W - ((g.shape[3] - 1) * self.stride - 2 * self.pad + self.k)
        dx = F.conv_transpose2d(g, self.wd, None, self.stride, self.pad, (oph, opw))
        if addend is not None:
            dx = dx + (addend if mode == 'same' else F.avg_pool2d(addend, 2) * 4.0)
        if mask_src is not None:
            dx = dx * (mask_src > 0).to(dx.dtype)
        return rnd(dx, self.quant)

    def wgrad(self, x, g):
        """Returns grads aligned with the parameters: BN -> (dw, dgamma, dbeta); bias -> (dw, dbias); else (dw,)."""
        G = torch.nn.grad.conv2d_weight(x, self.w.shape, g, self.stride, self.pad)
        colsum = g.sum((0, 2, 3))
        if self.invstd is not None:
            dw = G * self.scale.view(-1, 1, 1, 1)
            dgamma = ((self.wb * G).sum((1, 2, 3)) - self.mean * colsum) * self.invstd
            return [dw, dgamma, colsum]
        if self.shift is not None:
            return [G, colsum]
        return [G]


def _bn_of(sd, p):
    return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"])


class Block(object):
    def __init__(self, sd, p, kind, stride, has_down, quant):
        self.kind, self.p = kind, p
        if kind == "bottleneck":
            self.u1 = Unit(sd[p + ".conv1.weight"], _bn_of(sd, p + ".bn1"), None, 1, quant)
            self.u2 = Unit(sd[p + ".conv2.weight"], _bn_of(sd, p + ".bn2"), None, stride, quant)
            self.u3 = Unit(sd[p + ".conv3.weight"], _bn_of(sd, p + ".bn3"), None, 1, quant)
        else:
            self.u1 = Unit(sd[p + ".conv1.weight"], _bn_of(sd, p + ".bn1"), None, stride, quant)
            self.u2 = Unit(sd[p + ".conv2.weight"], _bn_of(sd, p + ".bn2"), None, 1, quant)
            self.u3 = None
        self.ud = Unit(sd[p + ".downsample.0.weight"], _bn_of(sd, p + ".downsample.1"), None, stride, quant) \
            if has_down else None

    def fwd(self, x):
        h1 = self.u1.fwd(x, relu=True)
        res = x if self.ud is None else self.ud.fwd(x)
        if self.kind == "bottleneck":
            h2 = self.u2.fwd(h1, relu=True)
            out = self.u3.fwd(h2, res, 'same', True)
        else:
            h2 = None
            out = self.u2.fwd(h1, res, 'same', True)
        self.saved = (x, h1, h2, out)
        return out

    def bwd(self, g, extra, mask_src, grads):
        x, h1, h2, out = self.saved
        hw = lambda t: (t.shape[2], t.shape[3])  # noqa: E731
        p = self.p
        if self.kind == "bottleneck":
            _put(grads, p, "conv3", "bn3", self.u3.wgrad(h2, g))
            g2 = self.u3.dgrad(g, hw(h2), mask_src=h2)
            _put(grads, p, "conv2", "bn2", self.u2.wgrad(h1, g2))
            g1 = self.u2.dgrad(g2, hw(h1), mask_src=h1)
        else:
            _put(grads, p, "conv2", "bn2", self.u2.wgrad(h1, g))
            g1 = self.u2.dgrad(g, hw(h1), mask_src=h1)
        _put(grads, p, "conv1", "bn1", self.u1.wgrad(x, g1))
        if self.ud is not None:
            _put(grads, p, "downsample.0", "downsample.1", self.ud.wgrad(x, g))
            t = self.ud.dgrad(g, hw(x), extra, 'same')
        elif extra is not None:
            t = rnd(g + extra, self.u1.quant)
        else:
            t = g
        return self.u1.dgrad(g1, hw(x), t, 'same', mask_src)


def _put(grads, p, conv, bn, vals):
    pre = (p + ".") if p else ""
    grads[pre + conv + ".weight"] = vals[0]
    if len(vals) == 3:
        grads[pre + bn + ".weight"] = vals[1]
        grads[pre + bn + ".bias"] = vals[2]
    elif len(vals) == 2:
        grads[pre + conv + ".bias"] = vals[1]


class Sched(object):
    """ResNet(depth) + FPN(…, 256, num_outs) as the explicit fused schedule.  ``forward`` keeps the tensors the
    HIP path saves; ``load_saved`` replaces them (teacher forcing with the GPU's own activations, so that the
    backward comparison is not blurred by ReLU-mask flips); ``backward`` returns the parameter gradients."""

    def __init__(self, sd_b, sd_f, depth, num_outs=5, quant=True):
        q = self.q = quant
        self.num_outs = num_outs
        kind, nblocks = ARCH[depth]
        expansion = 1 if kind == "basic" else 4
        self.stem = Unit(sd_b["conv1.weight"], _bn_of(sd_b, "bn1"), None, 2, q)
        self.blocks, self.stage_last = [], []
        inplanes = 64
        strides = (1, 2, 2, 2)
        for i, nb in enumerate(nblocks):
            planes = 64 * 2 ** i
            for b in range(nb):
                st = strides[i] if b == 0 else 1
                has_down = b == 0 and (st != 1 or inplanes != planes * expansion)
                self.blocks.append(Block(sd_b, "layer%d.%d" % (i + 1, b), kind, st, has_down, q))
            inplanes = planes * expansion
            self.stage_last.append(len(self.blocks) - 1)
        self.nlat = 4
        self.lat_u = [Unit(sd_f["lateral_convs.%d.conv.weight" % i], None, sd_f["lateral_convs.%d.conv.bias" % i],
                           1, q) for i in range(self.nlat)]
        self.fpn_u = [Unit(sd_f["fpn_convs.%d.conv.weight" % i], None, sd_f["fpn_convs.%d.conv.bias" % i], 1, q)
                      for i in range(self.nlat)]

    def forward(self, x):
        q, nlat = self.q, self.nlat
        self.x = rnd(x, q)
        self.s = self.stem.fwd(self.x, relu=True)
        cur, self.idx = F.max_pool2d(self.s, 3, 2, 1, return_indices=True)
        for blk in self.blocks:
            cur = blk.fwd(cur)
        self.feats = [self.blocks[i].saved[3] for i in self.stage_last]
        self.lat = [None] * nlat
        for i in reversed(range(nlat)):
            self.lat[i] = self.lat_u[i].fwd(self.feats[i]) if i == nlat - 1 else \
                self.lat_u[i].fwd(self.feats[i], self.lat[i + 1], 'up2x')
        outs = [self.fpn_u[i].fwd(self.lat[i]) for i in range(nlat)]
        for _ in range(self.num_outs - nlat):
            outs.append(outs[-1][:, :, ::2, ::2].contiguous())
        self.out_shapes = [tuple(o.shape) for o in outs]
        return tuple(outs)

    def load_saved(self, s, block_saved, lat):
        """Teacher forcing: NCHW fp32 copies of the tensors the HIP path saved in its forward."""
        self.s = s
        self.idx = F.max_pool2d(s, 3, 2, 1, return_indices=True)[1]
        for blk, sv in zip(self.blocks, block_saved):
            blk.saved = sv
        self.feats = [self.blocks[i].saved[3] for i in self.stage_last]
        self.lat = list(lat)

    def backward(self, cotangents):
        q, nlat = self.q, self.nlat
        hw = lambda t: (t.shape[2], t.shape[3])  # noqa: E731
        grads = {}
        d = [rnd(c, q) for c in cotangents]
        for j in range(self.num_outs - 1, nlat - 1, -1):
            up = torch.zeros(self.out_shapes[j - 1])
            up[:, :, ::2, ::2] = d[j]
            d[j - 1] = rnd(d[j - 1] + up, q)
        dL = [None] * nlat
        fg = {}
        for i in range(nlat):
            _put(fg, "fpn_convs.%d" % i, "conv", None, self.fpn_u[i].wgrad(self.lat[i], d[i]))
            dL[i] = self.fpn_u[i].dgrad(d[i], hw(self.lat[i]), dL[i - 1] if i > 0 else None, 'sumpool')
        dC = [None] * nlat
        for i in range(nlat):
            _put(fg, "lateral_convs.%d" % i, "conv", None, self.lat_u[i].wgrad(self.feats[i], dL[i]))
            dC[i] = self.lat_u[i].dgrad(dL[i], hw(self.feats[i]))
        for k, v in fg.items():
            grads["neck." + k] = v
        ext = {self.stage_last[i]: dC[i] for i in range(nlat)}
        bg = {}
        g = None
        for bi in reversed(range(len(self.blocks))):
            blk = self.blocks[bi]
            if g is None:
                g = ext[bi] * (blk.saved[3] > 0).float()
            extra = ext.get(bi - 1) if bi > 0 else None
            mask_src = self.blocks[bi - 1].saved[3] if bi > 0 else None
            g = blk.bwd(g, extra, mask_src, bg)
        # max-pool adjoint (first maximum wins) + stem ReLU mask, then stem wgrad
        s = self.s
        ds = torch.zeros_like(s).view(s.shape[0], s.shape[1], -1)
        ds.scatter_add_(2, self.idx.view(s.shape[0], s.shape[1], -1), g.reshape(g.shape[0], g.shape[1], -1))
        ds = rnd(ds.view_as(s) * (s > 0).float(), q)
        _put(bg, "", "conv1", "bn1", self.stem.wgrad(self.x, ds))
        for k, v in bg.items():
            grads["backbone." + k] = v
        return grads


def resnet_fpn_fwd_bwd(sd_b, sd_f, x, depth, cotangents, num_outs=5, quant=True):
    """Same contract as torch_ref.resnet_fpn_fwd_bwd, computed with the explicit fused schedule."""
    sch = Sched(sd_b, sd_f, depth, num_outs, quant)
    outs = sch.forward(x)
    return outs, sch.backward(cotangents)
