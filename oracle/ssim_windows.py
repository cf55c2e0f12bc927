"""TEST INFRASTRUCTURE (never imported by the product): SSIM by explicit windows.

The reference's `calculate_ssim` (src/util/error.py:52-65) is scikit-image's `structural_similarity(original, predicted,
data_range=...)` with every other argument at its default.  scikit-image is not installed in this image and the reference's tests
hold no SSIM vector, so `mri_inr_amd/metrics.py` stays PARITY UNPINNED against the library itself.  What this file adds is a second,
independent derivation of the published definition (Wang, Bovik, Sheikh, Simoncelli 2004, eq. 13, as skimage 0.19-0.22 documents
its defaults) that shares no code with metrics.py:

  * one 7 x 7 window per pixel whose window lies FULLY inside the image -- skimage filters the whole image (`uniform_filter`,
    reflecting at the border) and then crops (win_size - 1) // 2 pixels from every side before averaging, so the border mode never
    enters the result: the mean runs over exactly these windows;
  * per window: means, SAMPLE variances / covariance (divisor N - 1 = 48: `use_sample_covariance=True`), no Gaussian weights;
  * C1 = (0.01 R)^2, C2 = (0.03 R)^2 with R = data_range;
  * the mean of S over the windows.
Plain loops over windows through `sliding_window_view`: O(H W 49), fine at test sizes.
"""
import numpy as np


def ssim_by_windows(x, y, data_range, win=7):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    assert x.shape == y.shape and x.ndim == 2 and min(x.shape) >= win
    wx = np.lib.stride_tricks.sliding_window_view(x, (win, win)).reshape(x.shape[0] - win + 1, x.shape[1] - win + 1, -1)
    wy = np.lib.stride_tricks.sliding_window_view(y, (win, win)).reshape(wx.shape)
    n = win * win
    mx, my = wx.mean(-1), wy.mean(-1)
    dx, dy = wx - mx[..., None], wy - my[..., None]
    vx, vy, vxy = (dx * dx).sum(-1) / (n - 1), (dy * dy).sum(-1) / (n - 1), (dx * dy).sum(-1) / (n - 1)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * mx * my + c1) * (2 * vxy + c2)) / ((mx * mx + my * my + c1) * (vx + vy + c2))
    return float(s.mean())
