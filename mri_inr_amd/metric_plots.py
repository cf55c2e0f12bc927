"""The two summary plots of the reference's evaluation driver, under the reference's names and file names.

``test_mod_siren.py:248-256`` calls ``metrics_boxplot`` and ``metrics_density_plot`` (``src/util/visualization.py:129-165``) on the
PSNR / SSIM / NRMSE lists of the metric samples: one ``{key}_metrics_boxplot.png`` and one ``{key}_density_plot.png`` per metric in
the output directory.  The reference draws the density with seaborn's ``kdeplot`` (not in this image: the curve is restated here from
its published defaults -- a Gaussian kernel density estimate, Scott's bandwidth, 200 grid points from ``min - 3 bw`` to ``max + 3 bw``
-- with scipy); the box plot is plain matplotlib there as here.  ``save_image`` / ``save_image_comparison`` are the two image writers
``visual_error`` uses (``src/util/visualization.py:44-110``): min-max normalised, grey (the difference in viridis), a colour bar beside
the single images, four titled panels in the comparison.  Host-side presentation only: nothing on the hot path.
"""

from __future__ import annotations

import os

import numpy as np

KDE_GRIDSIZE, KDE_CUT = 200, 3.0  # seaborn.kdeplot defaults (gridsize, cut); bandwidth: scipy's "scott", bw_adjust = 1


def kde_curve(values):
    """(x, density) of seaborn's univariate ``kdeplot(values)`` at its defaults, or ``None`` where seaborn draws nothing (fewer than two
    points, or zero variance: it warns and leaves the axes empty)."""
    from scipy.stats import gaussian_kde

    v = np.asarray(values, dtype=np.float64).ravel()
    v = v[np.isfinite(v)]
    if v.size < 2 or np.var(v) == 0.0:
        return None
    kde = gaussian_kde(v, bw_method="scott")
    bw = float(np.sqrt(kde.covariance.squeeze()))
    x = np.linspace(v.min() - KDE_CUT * bw, v.max() + KDE_CUT * bw, KDE_GRIDSIZE)
    return x, kde(x)


def _pyplot():
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    return plt


def _one_figure_per_metric(metrics, output_dir, file_suffix, draw):
    """``draw(ax, key, values)`` on a fresh figure per metric, saved as ``{output_dir}/{key}{file_suffix}``."""
    plt = _pyplot()
    os.makedirs(output_dir, exist_ok=True)
    for key, values in metrics.items():
        fig, ax = plt.subplots()
        draw(ax, key, np.asarray(values, dtype=np.float64))
        ax.set_title(key)
        fig.savefig(os.path.join(output_dir, key + file_suffix))
        plt.close(fig)


def metrics_boxplot(metrics, output_dir, suffix=None):
    """One box plot per metric: ``{output_dir}/{key}_metrics_boxplot.png`` (visualization.py:129-146; ``suffix`` is accepted and unused
    there as well)."""
    def draw(ax, key, values):
        ax.boxplot(values)
        ax.set_xticklabels([key])

    _one_figure_per_metric(metrics, output_dir, "_metrics_boxplot.png", draw)


def metrics_density_plot(metrics, output_dir, suffix=None):
    """One kernel-density plot per metric: ``{output_dir}/{key}_density_plot.png`` (visualization.py:149-165)."""
    def draw(ax, key, values):
        curve = kde_curve(values)
        if curve is not None:
            ax.plot(curve[0], curve[1])
            ax.set_ylim(bottom=0)
        ax.set_ylabel("Density")

    _one_figure_per_metric(metrics, output_dir, "_density_plot.png", draw)


def normalize_scan(scan):
    """``(scan - min) / (max - min)`` (visualization.py:113-127); a constant image maps to zeros instead of the reference's NaNs."""
    a = np.asarray(scan, dtype=np.float64)
    lo, hi = float(a.min()), float(a.max())
    return (a - lo) / (hi - lo) if hi > lo else np.zeros_like(a)


def save_image(image, filename, output_dir, cmap="gray", dpi=300):
    """One image as ``{output_dir}/{filename}.png``: min-max normalised, shown on [0, 1] without axes, a colour bar beside it
    (visualization.py:85-110; the reference writes at 1200 dpi -- 25 MB per slice -- pass ``dpi=1200`` for that)."""
    plt = _pyplot()
    os.makedirs(output_dir, exist_ok=True)
    fig, ax = plt.subplots()
    im = ax.imshow(np.squeeze(normalize_scan(image)), cmap=cmap, vmin=0, vmax=1)
    ax.axis("off")
    cbar = fig.colorbar(im, ax=ax, fraction=0.046, pad=0.04)
    cbar.ax.tick_params(labelsize=8)
    fig.savefig(os.path.join(output_dir, f"{filename}.png"), bbox_inches="tight", pad_inches=0, dpi=dpi)
    plt.close(fig)


def save_image_comparison(fully_sampled, undersampled, reconstructed, path, cmap="gray"):
    """Undersampled | Fully Sampled | Reconstruction | Difference side by side, each min-max normalised, the last panel
    ``1 - |reconstruction - fully sampled|`` in viridis (visualization.py:44-82)."""
    plt = _pyplot()
    full, under, rec = (np.squeeze(normalize_scan(a)) for a in (fully_sampled, undersampled, reconstructed))
    fig, ax = plt.subplots(1, 4, figsize=(10, 5))
    panels = (("Undersampled", under, dict(cmap=cmap, vmin=0, vmax=1)), ("Fully Sampled", full, dict(cmap=cmap, vmin=0, vmax=1)),
              ("Reconstruction", rec, dict(cmap=cmap)), ("Difference", 1 - np.abs(rec - full), dict(cmap="viridis")))
    for a, (title, img, kw) in zip(ax, panels):
        a.imshow(img, **kw)
        a.set_title(title)
        a.axis("off")
    path = os.fspath(path)
    fig.savefig(path if path.lower().endswith(".png") else path + ".png")
    plt.close(fig)
