"""TEST INFRASTRUCTURE -- a second, independent MS-SSIM, written from the definition (Wang, Simoncelli, Bovik 2003: "Multi-scale
structural similarity for image quality assessment"), in float64, to cross-check `oracle/msssim_ref.py` (the restatement of
pytorch_msssim 0.2.1 that the reference calls at hnerv_utils.py:363, :370, :411) and the HIP kernels behind `bnerv_msssim`.

Deliberately different in form from the restatement: numpy float64 instead of torch float32; the Gaussian window is a genuine
2-D 11x11 kernel exp(-(dx^2 + dy^2) / (2 sigma^2)) / sum applied by DIRECT 2-D correlation (no separable passes); the local
moments are central moments computed from the definition E[(x - mu_x)(y - mu_y)] re-expanded per pixel; the scales are an explicit
loop with an explicit zero-padded 2x2 mean; the final product is a loop over scales.  Conventions it shares with the package,
because they DEFINE the quantity the reference optimises: window 11 / sigma 1.5, valid windows only, K = (0.01, 0.03), five scales with
exponents (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), contrast-structure term on scales 1..4 and the full SSIM on scale 5, negative
scale means clipped to 0, 2x2 mean pooling with zero padding on odd sides (divisor 4), mean over channels."""
import numpy as np
from scipy.ndimage import correlate

EXPONENTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def window2d(size=11, sigma=1.5):
    r = np.arange(size, dtype=np.float64) - size // 2
    w = np.exp(-(r[:, None] ** 2 + r[None, :] ** 2) / (2.0 * sigma * sigma))
    return w / w.sum()


def local_mean(img, w):
    """Valid-window weighted mean of a 2-D float64 image (direct 2-D correlation, borders cropped)."""
    h = w.shape[0] // 2
    full = correlate(img, w, mode="constant", cval=0.0)
    return full[h:img.shape[0] - h, h:img.shape[1] - h]


def scale_terms(x, y, data_range=1.0, K=(0.01, 0.03), w=None):
    """Mean over valid windows of (luminance * contrast-structure) and of contrast-structure alone, one 2-D plane."""
    c1, c2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    w = window2d() if w is None else w
    mx, my = local_mean(x, w), local_mean(y, w)
    vx = local_mean(x * x, w) - mx * mx
    vy = local_mean(y * y, w) - my * my
    cxy = local_mean(x * y, w) - mx * my
    cs = (2.0 * cxy + c2) / (vx + vy + c2)
    lum = (2.0 * mx * my + c1) / (mx * mx + my * my + c1)
    return float((lum * cs).mean()), float(cs.mean())


def halve(img):
    """2x2 mean with zero padding on odd sides (the sum of each 2x2 cell of the padded plane divided by 4)."""
    H, W = img.shape
    ph, pw = H % 2, W % 2
    p = np.zeros((H + 2 * ph, W + 2 * pw), dtype=np.float64)
    p[ph:ph + H, pw:pw + W] = img
    Ho, Wo = p.shape[0] // 2, p.shape[1] // 2
    p = p[:2 * Ho, :2 * Wo]
    return 0.25 * (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2])


def ms_ssim(X, Y, data_range=1.0, window=None):
    """X, Y: [B, C, H, W] array-likes.  Returns [B] float64: mean over channels of prod_s term_s ** exponent_s.
    `window`: another 11x11 weight array (the package rounds its 1-D Gaussian to float32; window2d() is the unrounded one)."""
    X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    assert X.shape == Y.shape and X.ndim == 4
    assert min(X.shape[2:]) > (11 - 1) * 2 ** 4, "four halvings must leave room for an 11x11 window"
    out = np.zeros(X.shape[0])
    for b in range(X.shape[0]):
        acc = 0.0
        for c in range(X.shape[1]):
            x, y = X[b, c], Y[b, c]
            val = 1.0
            for s, e in enumerate(EXPONENTS):
                full, cs = scale_terms(x, y, data_range, w=window)
                term = full if s == len(EXPONENTS) - 1 else cs
                val *= max(term, 0.0) ** e
                if s < len(EXPONENTS) - 1:
                    x, y = halve(x), halve(y)
            acc += val
        out[b] = acc / X.shape[1]
    return out
