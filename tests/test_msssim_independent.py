"""The MS-SSIM term of the headline loss (Fusion10_freq, hnerv_utils.py:369-370: 0.3 * (1 - ms_ssim)) cannot be pinned to
pytorch_msssim itself (not vendored, not installed).  These tests hold the restatement `oracle/msssim_ref.py` against a second,
independently written float64 form (tests/msssim_independent.py) and against identities of the definition; the GPU half
(tests/test_gpu_ops.py::test_msssim_kernel_against_independent_form) holds the HIP kernels against the same second form."""
import numpy as np
import pytest
import torch

import msssim_independent as ind
from oracle import msssim_ref


def _pair(shape, seed, noise=0.08):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(*shape, generator=g)
    # smooth the images a little so every scale carries structure, then perturb
    x = torch.nn.functional.avg_pool2d(x, 3, stride=1, padding=1)
    y = (x + noise * torch.randn(*shape, generator=g)).clamp(0, 1)
    return x, y


@pytest.mark.parametrize("shape,seed", [((1, 3, 176, 176), 1), ((2, 3, 177, 203), 2), ((1, 1, 180, 320), 3), ((1, 3, 191, 165), 4)])
def test_restatement_equals_independent_form(shape, seed):
    x, y = _pair(shape, seed)
    ref = msssim_ref.ms_ssim(x, y, data_range=1, size_average=False).double().numpy()
    got = ind.ms_ssim(x.numpy(), y.numpy())
    assert np.abs(ref - got).max() < 5e-6, (ref, got)          # the restatement runs in float32


def test_restatement_equals_independent_form_in_float64():
    x, y = _pair((1, 3, 179, 211), 5)
    ref = msssim_ref.ms_ssim(x.double(), y.double(), data_range=1, size_average=False).numpy()
    got = ind.ms_ssim(x.numpy(), y.numpy())
    assert np.abs(ref - got).max() < 1e-6, (ref, got)           # the package's window is a float32-rounded Gaussian (3e-7 on the result)
    w1 = msssim_ref.gauss_window_1d().double().numpy()           # ... with the SAME 121 weights the two forms agree to rounding
    got = ind.ms_ssim(x.numpy(), y.numpy(), window=np.outer(w1, w1))
    assert np.abs(ref - got).max() < 1e-12, (ref, got)


def test_identities_of_the_definition():
    x, y = _pair((1, 3, 176, 208), 6)
    for f in (lambda a, b: ind.ms_ssim(a.numpy(), b.numpy()), lambda a, b: msssim_ref.ms_ssim(a, b, data_range=1, size_average=False).double().numpy()):
        assert np.abs(f(x, x) - 1.0).max() < 1e-6                      # identical images
        assert np.abs(f(x, y) - f(y, x)).max() < 1e-6                  # symmetry
        assert (f(x, y) < 1.0).all() and (f(x, y) > 0.0).all()
    # constant planes (even sides down to the last scale, so no padded border): zero variance -> every contrast-structure term is 1,
    # the last scale's luminance term is (2ab + C1) / (a^2 + b^2 + C1)
    a, b = float(np.float32(0.3)), float(np.float32(0.7))       # the planes are float32
    X, Y = torch.full((1, 2, 176, 352), a), torch.full((1, 2, 176, 352), b)
    c1 = 0.01 ** 2
    want = ((2 * a * b + c1) / (a * a + b * b + c1)) ** 0.1333
    assert abs(ind.ms_ssim(X.numpy(), Y.numpy())[0] - want) < 1e-12
    # (float32: the variances E[x^2] - mu^2 of a constant plane cancel to ~1e-8, against C2 = 9e-4 -> 1e-5 on the result)
    assert abs(float(msssim_ref.ms_ssim(X, Y, data_range=1, size_average=False)[0]) - want) < 2e-5


def test_minimum_size_and_odd_side_pooling():
    # 161 is the smallest legal side (> (11 - 1) * 16): 161 -> 81 -> 41 -> 21 -> 11, every halving pads
    x, y = _pair((1, 1, 161, 161), 7)
    ref = msssim_ref.ms_ssim(x, y, data_range=1, size_average=False).double().numpy()
    got = ind.ms_ssim(x.numpy(), y.numpy())
    assert np.abs(ref - got).max() < 5e-6
    with pytest.raises(AssertionError):
        ind.ms_ssim(np.zeros((1, 1, 160, 200)), np.zeros((1, 1, 160, 200)))
    # the padded halving: an odd side gains one zero row / column in front AND behind before the 2x2 mean
    img = np.arange(15, dtype=np.float64).reshape(3, 5)
    h = ind.halve(img)
    t = torch.nn.functional.avg_pool2d(torch.from_numpy(img)[None, None], kernel_size=2, padding=[1, 1])[0, 0].numpy()
    assert h.shape == t.shape and np.abs(h - t).max() < 1e-15
