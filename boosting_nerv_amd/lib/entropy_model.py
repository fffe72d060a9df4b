"""Rate model of the CEM compression path (reference lib/entropy_model.py:14-43, :100-114): bits of a code tensor under a
Gaussian fitted to it, with additive uniform noise while training and on the rounded symbols otherwise.

`real_bitrate` (reference: constriction's QuantizedGaussian + AnsCoder, lib/entropy_model.py:46-62) -- the coder is a third-party
dependency that is not available here: this build reports the ideal code length of the symbols under the same quantised
Gaussian, rounded up to whole 32-bit words like the coder's output buffer.  PARITY UNPINNED for that one number."""
import math

import torch
from torch.autograd import Function


class LowerBound(Function):                                   # lib/entropy_model.py:100-114
    @staticmethod
    def forward(ctx, inputs, bound):
        b = torch.ones_like(inputs) * bound
        ctx.save_for_backward(inputs, b)
        return torch.max(inputs, b)

    @staticmethod
    def backward(ctx, grad_output):
        inputs, b = ctx.saved_tensors
        pass_through = (inputs >= b) | (grad_output < 0)
        return pass_through.type(grad_output.dtype) * grad_output, None


def ideal_code_bits(quant, mean, std):
    q = quant.detach().double().flatten().round()
    lo, hi = q.min(), q.max()
    if lo == hi:
        hi = lo + 1
    n = torch.distributions.normal.Normal(mean.detach().double(), std.detach().double().clamp(1e-5, 1e10))
    z = n.cdf(hi + 0.5) - n.cdf(lo - 0.5)
    p = (n.cdf(q + 0.5) - n.cdf(q - 0.5)) / z
    bits = float((-torch.log2(p.clamp_min(1e-300))).sum())
    return int(math.ceil(bits / 32.0)) * 32


class DiffEntropyModel:
    def __init__(self, distribution="gaussian"):
        if distribution != "gaussian":
            raise NotImplementedError("only the Gaussian rate model of the recipes is built (lib/entropy_model.py:36-39 also has Laplace)")
        self.distribution = distribution
        self.noise_source = None          # tests: callable(code) -> U(-.5,.5) draw, so both sides of a comparison share the numbers

    def cal_bitrate(self, code, quant, training):
        return self.cal_global_bitrate(code, quant, training)

    def cal_global_bitrate(self, code, quant, training):
        mean = torch.mean(code)
        std = torch.std(code)
        if training:
            noise = self.noise_source(code) if self.noise_source is not None else torch.empty_like(code).uniform_(-0.5, 0.5)
            x = code + noise
            real_bits = 0
        else:
            x = quant
            real_bits = ideal_code_bits(quant, mean, std)
        bits = torch.sum(self.get_bits(x, mean, std))
        return {"bitrate": bits, "mean": mean, "std": std, "real_bitrate": real_bits}

    def get_bits(self, x, mu, sigma):
        sigma = sigma.clamp(1e-5, 1e10)
        # validate_args=False: the default argument check reads a device flag on the host every call (a sync per step, and not
        # capturable in a graph); sigma is clamped positive just above, so the check cannot fail
        gaussian = torch.distributions.normal.Normal(mu, sigma, validate_args=False)
        probs = gaussian.cdf(x + 0.5) - gaussian.cdf(x - 0.5)
        bits = -1.0 * torch.log(probs + 1e-5) / math.log(2.0)
        return LowerBound.apply(bits, 0)
