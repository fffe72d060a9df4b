"""Rate model of the CEM compression path (reference lib/entropy_model.py:14-43, :100-114): bits of a code tensor under a
Gaussian fitted to it, with additive uniform noise while training and on the rounded symbols otherwise.

`real_bitrate` (reference: constriction's QuantizedGaussian + AnsCoder, lib/entropy_model.py:46-62) is the size of the message
the range-ANS coder of this build (csrc/ans.cpp behind bnerv_ans_encode_gaussian: same stream code and model construction as
constriction's defaults) produces for the rounded symbols -- a bitstream, not an estimate.  constriction itself is a third-party
dependency absent from the reference tree, so byte-for-byte equality with ITS stream is PARITY UNPINNED; the tests pin
decode(encode(x)) == x and the distance to the ideal code length."""
import ctypes as C
import math

import numpy as np
import torch
from torch.autograd import Function

from .. import _lib as L


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
    """-sum log2 p(q) under the quantised Gaussian restricted to the support of the symbols (the lower bound of any coder)."""
    q = quant.detach().double().flatten().round()
    lo, hi = q.min(), q.max()
    if lo == hi:
        hi = lo + 1
    n = torch.distributions.normal.Normal(mean.detach().double(), std.detach().double().clamp(1e-5, 1e10))
    z = n.cdf(hi + 0.5) - n.cdf(lo - 0.5)
    p = (n.cdf(q + 0.5) - n.cdf(q - 0.5)) / z
    return float((-torch.log2(p.clamp_min(1e-300))).sum())


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def ans_encode_gaussian(symbols, lo, hi, mean, std):
    """int32 symbols in [lo, hi] -> uint32 words of the rANS message (numpy)."""
    lib = L.load()
    sym = _i32(symbols)
    need = lib.bnerv_ans_encode_gaussian(sym.ctypes.data, sym.size, int(lo), int(hi), float(mean), float(std), None, 0)
    if need < 0:
        L.check(-1, "bnerv_ans_encode_gaussian")
    out = np.empty(max(need, 1), dtype=np.uint32)
    got = lib.bnerv_ans_encode_gaussian(sym.ctypes.data, sym.size, int(lo), int(hi), float(mean), float(std), out.ctypes.data, out.size)
    assert got == need
    return out[:need]


def ans_decode_gaussian(words, n, lo, hi, mean, std):
    lib = L.load()
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty(max(int(n), 1), dtype=np.int32)
    L.check(lib.bnerv_ans_decode_gaussian(w.ctypes.data, w.size, int(n), int(lo), int(hi), float(mean), float(std), out.ctypes.data), "bnerv_ans_decode_gaussian")
    return out[:n]


def ans_encode_categorical(symbols, probs):
    """symbols 0 .. K-1 under probs[K] -> uint32 words."""
    lib = L.load()
    sym = _i32(symbols)
    p = np.ascontiguousarray(probs, dtype=np.float64)
    need = lib.bnerv_ans_encode_categorical(sym.ctypes.data, sym.size, p.ctypes.data, p.size, None, 0)
    if need < 0:
        L.check(-1, "bnerv_ans_encode_categorical")
    out = np.empty(max(need, 1), dtype=np.uint32)
    lib.bnerv_ans_encode_categorical(sym.ctypes.data, sym.size, p.ctypes.data, p.size, out.ctypes.data, out.size)
    return out[:need]


def ans_decode_categorical(words, n, probs):
    lib = L.load()
    w = np.ascontiguousarray(words, dtype=np.uint32)
    p = np.ascontiguousarray(probs, dtype=np.float64)
    out = np.empty(max(int(n), 1), dtype=np.int32)
    L.check(lib.bnerv_ans_decode_categorical(w.ctypes.data, w.size, int(n), p.ctypes.data, p.size, out.ctypes.data), "bnerv_ans_decode_categorical")
    return out[:n]


def compress_matrix_flatten_gaussian_global(matrix, mean, std):
    """Bits of the rANS message of the integer tensor `matrix` under QuantizedGaussian(min, max, mean, std) -- the reference's
    function of the same name (lib/entropy_model.py:46-62), which returns 8 * bytes of constriction's compressed buffer."""
    mean = float(mean)
    std = float(torch.as_tensor(std).clamp(1e-5, 1e10))
    sym = matrix.detach().int().flatten().cpu().numpy()
    lo, hi = int(sym.min()), int(sym.max())
    if lo == hi:
        hi = lo + 1
    return int(ans_encode_gaussian(sym, lo, hi, mean, std).size) * 32


def compress_matrix_flatten_categorical(matrix):
    """(uint32 words, counts, unique values) of the flattened integer array under its own histogram (reference
    lib/entropy_model.py:65-81: np.unique -> Categorical -> AnsCoder)."""
    flat = np.asarray(matrix).reshape(-1)
    unique, inverse, counts = np.unique(flat, return_inverse=True, return_counts=True)
    words = ans_encode_categorical(inverse.astype(np.int32), counts.astype(np.float64) / counts.sum())
    return words, counts, unique


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
            real_bits = compress_matrix_flatten_gaussian_global(quant, mean, std)
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
