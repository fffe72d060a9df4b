"""TEST INFRASTRUCTURE -- CPU restatement of the CEM (compression) pieces that sit on the train step of
train_nerv_compression.py: the learned scale quantisers and the Gaussian rate estimate.  Pinned against the reference's own
modules by tests/golden/cem.npz (oracle/make_goldens.py gen_cem imports lib/transform_ops.py and lib/entropy_model.py).

Only tests, __graft_entry__.smoke() and bench.py's baseline legs may import this file.

  Scale_T.forward        lib/transform_ops.py:239-251   code = x / scale; quant = ste(code); dequant = quant * scale
  Scale_T.init_data      lib/transform_ops.py:221-237   scale = (max - min) / (qmax - qmin)            (per-tensor form)
  ScaleBeta_T            lib/transform_ops.py:253-286   code = (x - beta) / scale; dequant = ste(code) * scale + beta
  DiffEntropyModel       lib/entropy_model.py:14-43     bits = max(-log2(cdf(x+.5) - cdf(x-.5) + 1e-5), 0), Normal(mean(code), std(code)),
                                                        x = code + U(-.5,.5) when training, x = quant otherwise
  LowerBound             lib/entropy_model.py:100-114   max(x, b) whose gradient also passes where grad < 0
The ANS coder behind real_bitrate (constriction, lib/entropy_model.py:46-62) is a third-party dependency that is absent here:
PARITY UNPINNED for real bits; `ideal_bits` below is the Shannon length under the same leaky-free quantised Gaussian.
"""
import math

import torch


def ste(x):
    return (x.round() - x).detach() + x


def qrange(bits, signed):
    return (-2 ** (bits - 1), 2 ** (bits - 1) - 1) if signed else (0, 2 ** bits - 1)


def scale_init(t, bits, signed):
    qmin, qmax = qrange(bits, signed)
    return (t.max() - t.min()) / (qmax - qmin)


def scale_t(x, scale):
    code = x / scale
    quant = ste(code)
    return code, quant, quant * scale


def scalebeta_init(t, bits, signed):
    qmin, qmax = qrange(bits, signed)
    return (t.max() - t.min()) / (qmax - qmin), t.min()


def scalebeta_t(x, scale, beta):
    code = (x - beta) / scale
    quant = ste(code)
    return code, quant, quant * scale + beta


class _LowerBound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        b = torch.ones_like(x) * bound
        ctx.save_for_backward(x, b)
        return torch.max(x, b)

    @staticmethod
    def backward(ctx, g):
        x, b = ctx.saved_tensors
        return ((x >= b) | (g < 0)).type(g.dtype) * g, None


def gaussian_bits(x, mu, sigma):
    sigma = sigma.clamp(1e-5, 1e10)
    n = torch.distributions.normal.Normal(mu, sigma)
    probs = n.cdf(x + 0.5) - n.cdf(x - 0.5)
    bits = -1.0 * torch.log(probs + 1e-5) / math.log(2.0)
    return _LowerBound.apply(bits, 0)


def cal_bitrate(code, quant, training, noise=None):
    """noise: the U(-.5,.5) draw the reference takes with torch.empty_like(code).uniform_(-0.5, 0.5) (passed in so that both
    sides of a comparison use the same numbers)."""
    mean, std = torch.mean(code), torch.std(code)
    x = code + noise if training else quant
    return {"bitrate": torch.sum(gaussian_bits(x, mean, std)), "mean": mean, "std": std}


def ideal_bits(quant, mean, std):
    """Shannon code length of the integer symbols under the quantised Gaussian restricted to [min, max] (what an ideal entropy
    coder with constriction's QuantizedGaussian(min, max, mean, std) model would spend, without its leakiness / word padding)."""
    q = quant.detach().double().flatten().round()
    lo, hi = q.min(), q.max()
    if lo == hi:
        hi = lo + 1
    n = torch.distributions.normal.Normal(mean.detach().double(), std.detach().double().clamp(1e-5, 1e10))
    z = n.cdf(hi + 0.5) - n.cdf(lo - 0.5)
    p = (n.cdf(q + 0.5) - n.cdf(q - 0.5)) / z
    return float((-torch.log2(p.clamp_min(1e-300))).sum())
