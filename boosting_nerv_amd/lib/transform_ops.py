"""Learned quantisers of the CEM compression path (reference lib/transform_ops.py).  Built: the two the compression recipes
use (scripts/compression/*.sh: --quantizer_w scale --quantizer_b scale --quantizer_e scalebeta) with the reference's
constructor / init_form / init_data / forward surface and state_dict keys (`scale`, `beta`).  The other entries of the
reference's quant_map (lsq, lsqv2, dq, edgescale, multiscale, log, exp) raise NotImplementedError when constructed."""
import torch
import torch.nn as nn


def ste(x):                                                   # lib/transform_ops.py:8-9
    return (x.round() - x).detach() + x


def _qrange(bits, signed):
    return (-2 ** (bits - 1), 2 ** (bits - 1) - 1) if signed else (0, 2 ** bits - 1)


class Scale_T(nn.Module):                                     # lib/transform_ops.py:200-251
    def __init__(self, bits, signed=False, per_channel=False):
        super().__init__()
        self.scale = nn.Parameter(torch.Tensor([1.0]), requires_grad=True)
        self.init = False
        self.signed = signed
        self.per_channel = per_channel
        self.qmin, self.qmax = _qrange(bits, signed)

    def init_form(self, tensor):
        if self.per_channel:
            self.scale = nn.Parameter(torch.ones(tensor.size(0)), requires_grad=True)

    def init_data(self, tensor):
        if not self.init:
            device = tensor.device
            if self.per_channel:
                if len(tensor.shape) > 1:
                    t_min = tensor.flatten(1).min(dim=1)[0]
                    t_max = tensor.flatten(1).max(dim=1)[0]
                    self.scale.data = ((t_max - t_min) / (self.qmax - self.qmin)).to(device)
                else:   # the reference assigns a tuple here (lib/transform_ops.py:232): not a behaviour to reproduce
                    raise NotImplementedError("Scale_T(per_channel=True) on a 1-D tensor is broken in the reference")
            else:
                t_min, t_max = tensor.min(), tensor.max()
                self.scale.data = ((t_max - t_min) / (self.qmax - self.qmin)).reshape(1).to(device=device, dtype=torch.float32)
            self.init = True

    def encode(self, x):
        return x / self._bshape(x)

    def decode(self, x):
        return x * self._bshape(x)

    def _bshape(self, x):
        return self.scale.reshape(-1, *([1] * (x.dim() - 1))) if self.per_channel and x.dim() > 1 else self.scale

    def forward(self, x):
        code = self.encode(x)
        quant = ste(code)
        return code, quant, self.decode(quant)


class ScaleBeta_T(nn.Module):                                 # lib/transform_ops.py:253-286
    def __init__(self, bits, signed=False, per_channel=False):
        super().__init__()
        self.qmin, self.qmax = _qrange(bits, signed)
        self.scale = nn.Parameter(torch.Tensor([1.0]), requires_grad=True)
        self.beta = nn.Parameter(torch.Tensor([0.0]), requires_grad=True)
        self.init = False
        self.per_channel = per_channel

    def init_form(self, tensor):
        if self.per_channel:
            self.scale = nn.Parameter(torch.ones(tensor.size(1)), requires_grad=True)
            self.beta = nn.Parameter(torch.ones(tensor.size(1)), requires_grad=True)

    def init_data(self, tensor):
        if not self.init:
            device = tensor.device
            t_min, t_max = tensor.min(), tensor.max()
            self.beta.data = t_min.detach().reshape(1).to(device=device, dtype=torch.float32)
            self.scale.data = ((t_max - t_min) / (self.qmax - self.qmin)).detach().reshape(1).to(device=device, dtype=torch.float32)
        self.init = True

    def forward(self, x):
        code = (x - self.beta) / self.scale
        quant = ste(code)
        return code, quant, quant * self.scale + self.beta


def _unbuilt(name):
    class _Unbuilt(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"quantizer '{name}' (lib/transform_ops.py) is not part of this build; the compression recipes "
                                      f"use 'scale' and 'scalebeta'")
    _Unbuilt.__name__ = f"Unbuilt_{name}"
    return _Unbuilt


quant_map = {"scale": Scale_T, "scalebeta": ScaleBeta_T}
for _n in ("edgescale", "multiscale", "log", "exp", "lsq", "lsqv2", "dq"):
    quant_map[_n] = _unbuilt(_n)
