"""CPU restatements of the evaluation metrics the reference applies around the hot path.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's checker legs).

* `psnr_cv2`   - cv2.PSNR(a, b) for uint8 inputs: 10 log10(255^2 / MSE) (embed_process.py:205, app.py:342).
* `psnr_reference_quirk` - evaluation.psnr (evaluation.py:9-19): the squared difference is taken ON uint8
  arrays, i.e. modulo 256 - equal to the true PSNR only while every |difference| < 16.
* `ssim_skimage` - evaluation.calc_ssim (evaluation.py:21-26) = skimage.metrics.structural_similarity(a, b,
  data_range=b.max()-b.min()) with skimage's defaults.  Parity status: UNPINNED - scikit-image is not
  installed in the build image, so this restates skimage's published algorithm (Wang et al. 2004 as implemented in
  skimage/metrics/_structural_similarity.py: 7x7 `scipy.ndimage.uniform_filter`, K1 = 0.01, K2 = 0.03, sample
  covariance NP/(NP-1), float64, mean over the map cropped by (win-1)//2) using the same SciPy filter.
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import uniform_filter


def psnr_cv2(a: np.ndarray, b: np.ndarray) -> float:
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d * d).mean())
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def psnr_reference_quirk(a: np.ndarray, b: np.ndarray) -> float:
    mse = np.mean((a - b) ** 2)          # uint8 arithmetic wraps, as in the reference
    return float("inf") if mse == 0 else float(20 * np.log10(255.0 / np.sqrt(mse)))


def ssim_skimage(a: np.ndarray, b: np.ndarray, data_range=None, win_size: int = 7) -> float:
    if data_range is None:
        data_range = float(b.max()) - float(b.min())          # the reference's choice (evaluation.py:26)
    x, y = a.astype(np.float64), b.astype(np.float64)
    n_pix = win_size ** 2
    cov_norm = n_pix / (n_pix - 1)
    ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
    uxx, uyy, uxy = (uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size),
                     uniform_filter(x * y, size=win_size))
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win_size - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean(dtype=np.float64))
