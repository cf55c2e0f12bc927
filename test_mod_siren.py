#!/usr/bin/env python3
"""Evaluation driver with the reference's command line and artefacts, on the MI355X path.

    python test_mod_siren.py --config configuration/eval_sine.yaml

Mirrors test_mod_siren.py of the reference (:78-262): build ``ModulatedSiren`` from ``config.model``,
load ``config.testing.model_path``, reconstruct ``config.data.metric_samples`` slices and write
``metrics_error.csv`` + ``metrics_summary.txt`` (same formats, :38-75, :236-247) under
``{output_dir}/{output_name}/test``.  Differences, all forced by the container:
  * ``image_to_patches`` / ``metrics_error`` are the mirrors in ``mri_inr_amd/harness.py`` (same names, same positional
    arguments, :205-232; ``harness.bind(model)`` names the handle the free tiling functions run on): tiling, black filter, model, weighted fold and the fold of the fully-sampled tiles
    the reconstruction is scored against (error.py:229-254) run in libmsiren's kernels;
  * ``data.dataset: synthetic`` / ``testing.model_path: synthetic`` select seeded synthetic slices and
    weights (no fastMRI data or checkpoints here); a directory of ``*.npy`` slice pairs
    (``<name>_fully.npy`` / ``<name>_under.npy``) is read otherwise;
  * ``data.visual_samples`` slices get the reference's per-slice image folder (:122-173; ``harness.visual_error``: arrays as
    ``.npy``, images as ``.png`` through mirrors of the reference's ``save_image`` / ``save_image_comparison``, the slice's ``_error.txt``);
  * the box and density plots of the metric samples (:248-256) come from ``mri_inr_amd/metric_plots.py``: same function names and file
    names; the density curve restates seaborn's ``kdeplot`` defaults with scipy (seaborn is not in the image).
"""

from __future__ import annotations

import glob
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from mri_inr_amd import ModulatedSiren, load_configuration, model_kwargs, synthetic  # noqa: E402
from mri_inr_amd.configuration import parse_args  # noqa: E402
from mri_inr_amd import harness  # noqa: E402
from mri_inr_amd.harness import image_to_patches, metrics_error, visual_error  # noqa: E402
from mri_inr_amd.metric_plots import metrics_boxplot, metrics_density_plot  # noqa: E402
from mri_inr_amd.weights import load_checkpoint  # noqa: E402


def save_args_to_file(args, output_dir):
    """``config.txt``: one ``name: value`` line per top-level entry of the configuration (reference :18-33)."""
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, "config.txt"), "w") as f:
        for arg, value in vars(args).items():
            f.write(f"{arg}: {value}\n")


def save_metrics_summary(psnr_values, ssim_values, nrmse_values, output_dir):
    with open(os.path.join(output_dir, "metrics_summary.txt"), "w") as f:
        for metric, v in (("PSNR", psnr_values), ("SSIM", ssim_values), ("NRMSE", nrmse_values)):
            f.write(f"{metric}:\n")
            for stat, fn in (("mean", np.mean), ("std", np.std), ("min", np.min), ("max", np.max)):
                f.write(f"  {stat}: {fn(v)}\n")
            f.write("\n")


def samples(config, n=None):
    """Yield (fully_sampled, undersampled, filename) float32 (H, W) pairs."""
    n = config.data.metric_samples if n is None else n
    if config.data.dataset == "synthetic":
        for k in range(n or 8):
            full = synthetic.make_slice(k, brain_mask=True)
            # a crude stand-in for undersampling artefacts: horizontal blur (no fastmri mask maths here)
            under = (full + np.roll(full, 1, 1) + np.roll(full, -1, 1)) / np.float32(3)
            yield full, under.astype(np.float32), f"synthetic_{k:04d}"
        return
    files = sorted(glob.glob(os.path.join(config.data.dataset, "*_fully.npy")))
    for path in files[: (n or len(files))]:
        name = os.path.basename(path)[: -len("_fully.npy")]
        yield (np.load(path).astype(np.float32), np.load(path.replace("_fully", "_under")).astype(np.float32), name)


def test_mod_siren(config):
    print("Testing the modulated SIREN...")
    output_dir = f"{config.testing.output_dir}/{config.testing.output_name}/test"
    os.makedirs(output_dir, exist_ok=True)
    save_args_to_file(config, output_dir)
    model = ModulatedSiren(**model_kwargs(config, device="cuda", modulate=True))
    if config.testing.model_path == "synthetic":
        sd = synthetic.make_state_dict(seed=7, dim_hidden=config.model.dim_hidden, num_layers=config.model.num_layers,
                                       latent_dim=config.model.latent_dim, w0=config.model.w0,
                                       siren_patch_size=config.model.siren_patch_size,
                                       use_bias=config.model.use_bias, trained_like=True)
    else:
        sd = load_checkpoint(config.testing.model_path)
    model.load_state_dict(sd)
    model.to("cuda")
    model.eval()
    harness.bind(model)  # the tiling functions below run on this model's device / stream

    names, psnrs, ssims, nrmses = [], [], [], []
    t_gpu = 0.0
    O, I, S = config.model.outer_patch_size, config.model.inner_patch_size, config.model.siren_patch_size
    if config.data.visual_samples > 0:
        print("Evaluating visual samples ...")
        for i, (full, under, name) in enumerate(samples(config, config.data.visual_samples)):
            print(f"Processing visual sample {i + 1}/{config.data.visual_samples}...")
            fully_sampled_patch, _ = image_to_patches(model.device_array((1,) + full.shape).copy_from(full[None]), O, I)
            undersampled_patch, undersampled_information = image_to_patches(
                model.device_array((1,) + under.shape).copy_from(under[None]), O, I)
            visual_error(model, os.path.join(output_dir, name), name, fully_sampled_patch, undersampled_patch,
                         undersampled_information, "cuda", O, I, S)
    print("Evaluating metric samples ...")
    for i, (full, under, name) in enumerate(samples(config)):
        print(f"Processing metric sample {i + 1}...")
        t0 = time.perf_counter()
        # unsqueeze image to add batch dimension (reference :206-207), tile both images, score
        # (device-resident: the slices go up once, the tiles never come back to the host)
        fully_sampled_patch, _ = image_to_patches(model.device_array((1,) + full.shape).copy_from(full[None]), O, I)
        undersampled_patch, undersampled_information = image_to_patches(
            model.device_array((1,) + under.shape).copy_from(under[None]), O, I)
        psnr, ssim, nrmse = metrics_error(model, fully_sampled_patch, undersampled_patch, undersampled_information,
                                          "cuda", O, I, S)
        t_gpu += time.perf_counter() - t0
        names.append(name)
        psnrs.append(psnr)
        ssims.append(ssim)
        nrmses.append(nrmse)
    with open(os.path.join(output_dir, "metrics_error.csv"), "w") as f:
        f.write("FILENAME,PSNR,SSIM,NRMSE\n")
        for row in zip(names, psnrs, ssims, nrmses):
            f.write(",".join(str(v) for v in row) + "\n")
    save_metrics_summary(psnrs, ssims, nrmses, output_dir)
    # Visualize the metrics (reference :248-256)
    metrics_boxplot({"PSNR": psnrs, "SSIM": ssims, "NRMSE": nrmses}, output_dir)
    metrics_density_plot({"PSNR": psnrs, "SSIM": ssims, "NRMSE": nrmses}, output_dir)
    print(f"{len(names)} slices scored in {t_gpu:.3f} s host wall (incl. H2D/D2H and the metrics) -> {output_dir}")
    return output_dir


if __name__ == "__main__":
    args = parse_args()
    test_mod_siren(load_configuration(args.config, testing=True))
