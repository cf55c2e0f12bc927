"""Image-quality metrics of the evaluation harness (src/util/error.py:23-84).

The reference calls scikit-image (``peak_signal_noise_ratio``, ``structural_similarity``,
``normalized_root_mse``) with ``data_range = max - min over both images`` (error.py:23-38).
scikit-image is not installed in this image, so these are restatements of its published
definitions (PSNR, NRMSE: closed forms; SSIM: Wang et al. 2004 with skimage's defaults -- 7x7
uniform window, K1=0.01, K2=0.03, sample covariance, mean over the window-valid interior).
PARITY UNPINNED for SSIM: no skimage here to generate a fixture (tests/test_host_logic.py checks it against a second,
independent derivation by explicit windows, oracle/ssim_windows.py); PSNR/NRMSE are exact formulas.
Host-side, off the hot path.
"""

from __future__ import annotations

import numpy as np


def calculate_data_range(original, predicted) -> float:
    return float(max(np.max(original), np.max(predicted)) - min(np.min(original), np.min(predicted)))


def calculate_psnr(original, predicted) -> float:
    o = np.asarray(original, dtype=np.float64)
    p = np.asarray(predicted, dtype=np.float64)
    mse = np.mean((o - p) ** 2)
    dr = calculate_data_range(original, predicted)
    return float(10.0 * np.log10(dr * dr / mse))


def calculate_nrmse(original, predicted) -> float:
    """skimage default normalization='euclidean': sqrt(mean((o-p)^2)) / sqrt(mean(o^2))."""
    o = np.asarray(original, dtype=np.float64)
    p = np.asarray(predicted, dtype=np.float64)
    return float(np.sqrt(np.mean((o - p) ** 2)) / np.sqrt(np.mean(o * o)))


def calculate_ssim(original, predicted, win_size: int = 7) -> float:
    from scipy.ndimage import uniform_filter

    o = np.asarray(original, dtype=np.float64)
    p = np.asarray(predicted, dtype=np.float64)
    dr = calculate_data_range(original, predicted)
    c1, c2 = (0.01 * dr) ** 2, (0.03 * dr) ** 2
    n = win_size * win_size
    cov_norm = n / (n - 1.0)  # sample covariance
    ux, uy = uniform_filter(o, win_size), uniform_filter(p, win_size)
    uxx, uyy, uxy = uniform_filter(o * o, win_size), uniform_filter(p * p, win_size), uniform_filter(o * p, win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win_size - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())
