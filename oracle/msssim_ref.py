"""TEST INFRASTRUCTURE -- restatement of ``pytorch_msssim==0.2.1`` (ms_ssim / ssim).

PARITY UNPINNED: the package is a third-party dependency of the reference
(/root/reference/requirements.txt:11; call sites hnerv_utils.py:8, :363, :370, :411) that is
not vendored and not installed in this image, and the reference has no tests or golden vectors
for it.  This file restates the published algorithm of that version:

  * 1-D Gaussian window, size 11, sigma 1.5, normalised to sum 1
  * separable "valid" (no padding) depthwise filtering, first along H then along W
  * SSIM constants K=(0.01, 0.03), C1=(K1*L)^2, C2=(K2*L)^2
  * cs_map   = (2*s12 + C2) / (s1 + s2 + C2)
    ssim_map = (2*mu1*mu2 + C1) / (mu1^2 + mu2^2 + C1) * cs_map
  * 5 levels, weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333); between levels a 2x2 average
    pool with padding = size % 2 (count_include_pad=True); relu on every level's statistic;
    ms_ssim = prod_l stat_l ** w_l  per (batch, channel); size_average=False -> mean over channels.

Only plain torch CPU ops are used.  It is used (a) as the ``pytorch_msssim`` stand-in module when
the reference is imported by ``ref_harness.py`` (a stand-in for an un-vendored *dependency*, not
for reference source) and (b) by ``cpu_ref.loss_fn``.
"""
import torch
import torch.nn.functional as F

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
WIN_SIZE = 11
WIN_SIGMA = 1.5
K1, K2 = 0.01, 0.03


def gauss_window_1d(size=WIN_SIZE, sigma=WIN_SIGMA, dtype=torch.float32):
    coords = torch.arange(size, dtype=torch.float32)
    coords -= size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g /= g.sum()
    return g.to(dtype)


def gaussian_filter(x, win1d):
    """Depthwise separable valid filtering of [B,C,H,W]: along H, then along W."""
    C = x.shape[1]
    k = win1d.numel()
    out = x
    if x.shape[2] >= k:
        out = F.conv2d(out, win1d.view(1, 1, k, 1).repeat(C, 1, 1, 1), stride=1, padding=0, groups=C)
    if x.shape[3] >= k:
        out = F.conv2d(out, win1d.view(1, 1, 1, k).repeat(C, 1, 1, 1), stride=1, padding=0, groups=C)
    return out


def _ssim(X, Y, data_range, win1d):
    C1 = (K1 * data_range) ** 2
    C2 = (K2 * data_range) ** 2
    mu1 = gaussian_filter(X, win1d)
    mu2 = gaussian_filter(Y, win1d)
    mu1_sq = mu1.pow(2)
    mu2_sq = mu2.pow(2)
    mu1_mu2 = mu1 * mu2
    sigma1_sq = gaussian_filter(X * X, win1d) - mu1_sq
    sigma2_sq = gaussian_filter(Y * Y, win1d) - mu2_sq
    sigma12 = gaussian_filter(X * Y, win1d) - mu1_mu2
    cs_map = (2 * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    ssim_per_channel = torch.flatten(ssim_map, 2).mean(-1)
    cs = torch.flatten(cs_map, 2).mean(-1)
    return ssim_per_channel, cs


def ssim(X, Y, data_range=255, size_average=True, win_size=WIN_SIZE, win_sigma=WIN_SIGMA,
         win=None, K=(K1, K2), nonnegative_ssim=False):
    win1d = gauss_window_1d(win_size, win_sigma, X.dtype).to(X.device)
    ssim_per_channel, _ = _ssim(X, Y, data_range, win1d)
    if nonnegative_ssim:
        ssim_per_channel = torch.relu(ssim_per_channel)
    return ssim_per_channel.mean() if size_average else ssim_per_channel.mean(1)


def ms_ssim(X, Y, data_range=255, size_average=True, win_size=WIN_SIZE, win_sigma=WIN_SIGMA,
            win=None, weights=None, K=(K1, K2)):
    if X.shape != Y.shape:
        raise ValueError("Input images should have the same dimensions.")
    smaller_side = min(X.shape[-2:])
    assert smaller_side > (win_size - 1) * (2 ** 4), \
        "Image size should be larger than %d due to the 4 downsamplings in ms-ssim" % ((win_size - 1) * (2 ** 4))
    if weights is None:
        weights = MS_WEIGHTS
    wts = torch.tensor(weights, dtype=X.dtype, device=X.device)
    win1d = gauss_window_1d(win_size, win_sigma, X.dtype).to(X.device)
    levels = wts.shape[0]
    mcs = []
    for i in range(levels):
        ssim_per_channel, cs = _ssim(X, Y, data_range, win1d)
        if i < levels - 1:
            mcs.append(torch.relu(cs))
            padding = [s % 2 for s in X.shape[2:]]
            X = F.avg_pool2d(X, kernel_size=2, padding=padding)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=padding)
    ssim_per_channel = torch.relu(ssim_per_channel)
    mcs_and_ssim = torch.stack(mcs + [ssim_per_channel], dim=0)  # (level, batch, channel)
    ms_ssim_val = torch.prod(mcs_and_ssim ** wts.view(-1, 1, 1), dim=0)
    return ms_ssim_val.mean() if size_average else ms_ssim_val.mean(1)
