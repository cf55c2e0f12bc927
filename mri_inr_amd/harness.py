"""Host mirror of the evaluation harness around the model call, on the device through the C ABI.

Same names and argument meaning as the reference's functions, so that a driver written against the reference reads
the same here:

    tiles, info = image_to_patches(model, image, outer, inner)            # src/util/tiling.py:10-64
    image       = patches_to_image(model, tiles, info, outer, inner)      # src/util/tiling.py:143-181
    psnr, ssim, nrmse = metrics_error(model, fully_tiles, under_tiles, info, device, outer, inner, siren)
                                                                          # src/util/error.py:200-271

``metrics_error`` does what the reference does, in its order: black-tile filter -> model on the kept tiles -> zeros
re-inserted -> weighted overlap-add of the 24x24 outputs; the fully-sampled tiles folded with the plain overlap
average give the image the reconstruction is scored against (error.py:251-254) -- not the raw slice: on sizes that
are not a multiple of ``inner`` the scored image includes the reflect-padded rim.  Arrays are numpy float32; every
step runs in libmsiren's kernels (msiren_reconstruct_tiles_dev is msiren_reconstruct_slices' chain without the cut).  Metrics: mri_inr_amd/metrics.py (host, off the hot path).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .metrics import calculate_nrmse, calculate_psnr, calculate_ssim


def _shape(model, height, width):
    nv, nh = C.c_int32(), C.c_int32()
    _lib.check(model._lib.msiren_recon_shape(model._h, int(height), int(width), C.byref(nv), C.byref(nh)))
    return nv.value, nh.value


def _check_sizes(model, outer, inner):
    if (outer, inner) != (model.outer_patch_size, model.inner_patch_size):
        raise ValueError(f"patch sizes ({outer}, {inner}) differ from the model's ({model.outer_patch_size}, {model.inner_patch_size})")


def image_to_patches(model, tensor, outer_patch_size, inner_patch_size):
    """(batch, H, W) images of one size -> ((batch*nV*nH, O, O) tiles, [(nV, nH)] * batch).  tiling.py:10-64."""
    _check_sizes(model, outer_patch_size, inner_patch_size)
    model._ensure_committed()
    a = np.ascontiguousarray(tensor, dtype=np.float32)
    if a.ndim != 3:
        raise ValueError(f"expected (batch, H, W), got {a.shape}")
    n, hh, ww = a.shape
    nv, nh = _shape(model, hh, ww)
    d_img = model.device_array(a.shape).copy_from(a)
    d_p = model.device_array((n * nv * nh, outer_patch_size, outer_patch_size))
    _lib.check(model._lib.msiren_image_to_patches_dev(model._h, d_img.ptr, n, hh, ww, d_p.ptr))
    model.sync()
    return d_p.numpy(), [(nv, nh)] * n


def patches_to_image(model, tiles, image_information, outer_patch_size, inner_patch_size):
    """Plain overlap average of O x O tiles -> (n, nV*I, nH*I); output size from image_information[0].  tiling.py:143-181."""
    _check_sizes(model, outer_patch_size, inner_patch_size)
    model._ensure_committed()
    t = np.ascontiguousarray(tiles, dtype=np.float32)
    nv, nh = image_information[0]
    n = t.shape[0] // (nv * nh)
    if t.shape != (n * nv * nh, outer_patch_size, outer_patch_size):
        raise ValueError(f"tiles {t.shape} do not match image_information {image_information[0]}")
    d_t = model.device_array(t.shape).copy_from(t)
    d_o = model.device_array((n, nv * inner_patch_size, nh * inner_patch_size))
    _lib.check(model._lib.msiren_patches_to_image_dev(model._h, d_t.ptr, n, nv, nh, d_o.ptr))
    model.sync()
    return d_o.numpy()


def reconstruct_from_patches(model, undersampled, img_information):
    """filter_and_remember_black_patches -> model -> reintegrate_black_patches -> patches_to_image_weighted_average
    (error.py:229-249): (n*nV*nH, O, O) tiles -> (n, nV*I, nH*I), one device call (msiren_reconstruct_tiles_dev)."""
    model._ensure_committed()
    t = np.ascontiguousarray(undersampled, dtype=np.float32)
    nv, nh = img_information[0]
    n = t.shape[0] // (nv * nh)
    if t.ndim != 3 or t.shape[0] != n * nv * nh:
        raise ValueError(f"tiles {t.shape} do not match image_information {img_information[0]}")
    d_t = model.device_array(t.shape).copy_from(t)
    d_o = model.device_array((n, nv * model.inner_patch_size, nh * model.inner_patch_size))
    _lib.check(model._lib.msiren_reconstruct_tiles_dev(model._h, d_t.ptr, n, nv, nh, d_o.ptr))
    model.sync()
    return d_o.numpy()


def metrics_error(model, fully_sampled, undersampled, img_information, device, outer_patch_size, inner_patch_size,
                  siren_patch_size):
    """PSNR / SSIM / NRMSE of the reconstruction against the folded fully-sampled tiles.  error.py:200-271."""
    if siren_patch_size != model.siren_patch_size:
        raise ValueError(f"siren_patch_size {siren_patch_size} differs from the model's {model.siren_patch_size}")
    rec = reconstruct_from_patches(model, undersampled, img_information)[0]
    full = patches_to_image(model, fully_sampled, img_information, outer_patch_size, inner_patch_size)[0]
    return calculate_psnr(full, rec), calculate_ssim(full, rec), calculate_nrmse(full, rec)
