"""Host mirror of the evaluation harness around the model call, on the device through the C ABI.

Same names, same positional arguments as the reference's functions, so that a driver written against the reference reads
the same here:

    tiles, info = image_to_patches(image, outer, inner)                   # src/util/tiling.py:10
    image       = patches_to_image(tiles, info, outer, inner)             # src/util/tiling.py:143
    psnr, ssim, nrmse = metrics_error(model, fully_tiles, under_tiles, info, device, outer, inner, siren)
                                                                          # src/util/error.py:200-271

The reference's tiling functions are free functions of torch; here they run in libmsiren's kernels, which live behind a
handle -- the model's.  ``bind(model)`` names the model whose handle (device, stream) the free functions use (done by
``metrics_error`` for its own model, and by ``test_mod_siren.py`` right after ``model.to(device)``); a ``model=`` keyword
overrides it per call.

Arrays: numpy float32 in -> numpy out; a ``DeviceArray`` (``model.device_array``) in -> a ``DeviceArray`` out, nothing
crosses PCIe.  ``metrics_error`` keeps everything between its steps on the device: tiles are uploaded once, the
reconstruction and the folded fully-sampled image come back once, for the host-side metrics (mri_inr_amd/metrics.py).

``visual_error`` (src/util/error.py:104-183) is the same chain for ONE slice with images written instead of scores.

The steps of that chain also exist one by one, as the reference's callers spell them (src/train/training.py:438-445,
src/util/error.py:132-152): ``filter_and_remember_black_patches`` -> ``model(...)`` -> ``reintegrate_black_patches`` ->
``patches_to_image_weighted_average`` (tiling.py:244-303, 91-140).  ``reconstruct_from_patches`` does the four in one device call.

``metrics_error`` does what the reference does, in its order: black-tile filter -> model on the kept tiles -> zeros
re-inserted -> weighted overlap-add of the 24x24 outputs; the fully-sampled tiles folded with the plain overlap
average give the image the reconstruction is scored against (error.py:251-254) -- not the raw slice: on sizes that
are not a multiple of ``inner`` the scored image includes the reflect-padded rim.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .metrics import calculate_nrmse, calculate_psnr, calculate_ssim
from .model import DeviceArray

_BOUND = None


def bind(model):
    """The model whose handle image_to_patches / patches_to_image use from now on (returns it)."""
    global _BOUND
    _BOUND = model
    return model


def _model(model):
    m = model if model is not None else _BOUND
    if m is None:
        raise RuntimeError("no model bound: call mri_inr_amd.harness.bind(model) (or pass model=...) before the tiling functions")
    return m


def _shape(model, height, width):
    nv, nh = C.c_int32(), C.c_int32()
    _lib.check(model._lib.msiren_recon_shape(model._h, int(height), int(width), C.byref(nv), C.byref(nh)))
    return nv.value, nh.value


def _check_sizes(model, outer, inner):
    if (outer, inner) != (model.outer_patch_size, model.inner_patch_size):
        raise ValueError(f"patch sizes ({outer}, {inner}) differ from the model's ({model.outer_patch_size}, {model.inner_patch_size})")


def _to_device(model, x, ndim):
    """numpy / DeviceArray -> (DeviceArray, came_from_host)."""
    if isinstance(x, DeviceArray):
        if len(x.shape) != ndim:
            raise ValueError(f"expected a {ndim}-d array, got {x.shape}")
        return x, False
    a = np.ascontiguousarray(x, dtype=np.float32)
    if a.ndim != ndim:
        raise ValueError(f"expected a {ndim}-d array, got {a.shape}")
    return model.device_array(a.shape).copy_from(a), True


def image_to_patches(tensor, outer_patch_size, inner_patch_size, *, model=None):
    """(batch, H, W) images of one size -> ((batch*nV*nH, O, O) tiles, [(nV, nH)] * batch).  tiling.py:10-64."""
    model = _model(model)
    _check_sizes(model, outer_patch_size, inner_patch_size)
    model._ensure_committed()
    d_img, host = _to_device(model, tensor, 3)
    n, hh, ww = d_img.shape
    nv, nh = _shape(model, hh, ww)
    d_p = model.device_array((n * nv * nh, outer_patch_size, outer_patch_size))
    _lib.check(model._lib.msiren_image_to_patches_dev(model._h, d_img.ptr, n, hh, ww, d_p.ptr))
    model.sync()
    return (d_p.numpy() if host else d_p), [(nv, nh)] * n


def patches_to_image(tiles, image_information, outer_patch_size, inner_patch_size, *, model=None):
    """Plain overlap average of O x O tiles -> (n, nV*I, nH*I); output size from image_information[0].  tiling.py:143-181."""
    model = _model(model)
    _check_sizes(model, outer_patch_size, inner_patch_size)
    model._ensure_committed()
    d_t, host = _to_device(model, tiles, 3)
    nv, nh = image_information[0]
    n = d_t.shape[0] // (nv * nh)
    if d_t.shape != (n * nv * nh, outer_patch_size, outer_patch_size):
        raise ValueError(f"tiles {d_t.shape} do not match image_information {image_information[0]}")
    d_o = model.device_array((n, nv * inner_patch_size, nh * inner_patch_size))
    _lib.check(model._lib.msiren_patches_to_image_dev(model._h, d_t.ptr, n, nv, nh, d_o.ptr))
    model.sync()
    return d_o.numpy() if host else d_o


def _int_device(model, a):
    """int32 numpy -> device (a DeviceArray holds 4-byte words: the bit patterns travel as they are)."""
    a = np.ascontiguousarray(a, dtype=np.int32)
    return model.device_array(a.shape).copy_from(a.view(np.float32))


def filter_and_remember_black_patches(patches, *, model=None):
    """(N, h, w) tiles -> (the non-black tiles in order, indices of the black ones, the original shape).  tiling.py:244-271;
    black = mean < 1e-10 (classify_patches, :184-198), decided on the device; the index list is the only thing that
    visits the host (the reference's is a Python list too)."""
    model = _model(model)
    model._ensure_committed()
    d, host = _to_device(model, patches, 3)
    n, hh, ww = d.shape
    d_flags = model.device_array((max(n, 1),))
    _lib.check(model._lib.msiren_black_patch_flags_dev(model._h, d.ptr, n, hh * ww, d_flags.ptr))
    model.sync()
    flags = d_flags.numpy().view(np.int32)[:n]
    black_indices = [int(i) for i in np.nonzero(flags)[0]]
    keep = np.nonzero(flags == 0)[0].astype(np.int32)
    d_keep = model.device_array((len(keep), hh, ww))
    if len(keep):
        d_idx = _int_device(model, keep)
        _lib.check(model._lib.msiren_gather_rows_dev(model._h, d.ptr, d_idx.ptr, len(keep), hh * ww, d_keep.ptr))
        model.sync()
    return (d_keep.numpy() if host else d_keep), black_indices, (n, hh, ww)


def reintegrate_black_patches(processed_patches, black_indices, original_shape, *, model=None):
    """Processed non-black tiles back at their positions, zeros where the black ones were.  tiling.py:274-303."""
    model = _model(model)
    model._ensure_committed()
    d, host = _to_device(model, processed_patches, 3)
    n_keep, hh, ww = d.shape
    n = int(original_shape[0])
    black = np.zeros(n, dtype=bool)
    black[np.asarray(black_indices, dtype=np.int64)] = True
    keep = np.nonzero(~black)[0].astype(np.int32)
    if len(keep) != n_keep:
        raise ValueError(f"{n_keep} processed tiles for {len(keep)} non-black positions of {n}")
    d_full = model.device_array((n, hh, ww))
    d_idx = _int_device(model, keep if len(keep) else np.zeros(1, np.int32))
    _lib.check(model._lib.msiren_scatter_rows_dev(model._h, d.ptr, d_idx.ptr, n_keep, n, hh * ww, d_full.ptr))
    model.sync()
    return d_full.numpy() if host else d_full


def patches_to_image_weighted_average(patches, image_information, patch_size, inner_patch_size, device=None, *, model=None):
    """Weighted overlap-add of the (N, S, S) model outputs -> (n, nV*I, nH*I).  tiling.py:91-140 (weights :67-88)."""
    model = _model(model)
    if (patch_size, inner_patch_size) != (model.siren_patch_size, model.inner_patch_size):
        raise ValueError(f"patch sizes ({patch_size}, {inner_patch_size}) differ from the model's ({model.siren_patch_size}, {model.inner_patch_size})")
    model._ensure_committed()
    d, host = _to_device(model, patches, 3)
    nv, nh = image_information[0]
    n = d.shape[0] // (nv * nh)
    if d.shape != (n * nv * nh, patch_size, patch_size):
        raise ValueError(f"tiles {d.shape} do not match image_information {image_information[0]}")
    d_o = model.device_array((n, nv * inner_patch_size, nh * inner_patch_size))
    _lib.check(model._lib.msiren_weighted_fold_dev(model._h, d.ptr, n, nv, nh, d_o.ptr))
    model.sync()
    return d_o.numpy() if host else d_o


def reconstruct_from_patches(model, undersampled, img_information):
    """filter_and_remember_black_patches -> model -> reintegrate_black_patches -> patches_to_image_weighted_average
    (error.py:229-249): (n*nV*nH, O, O) tiles -> (n, nV*I, nH*I), one device call (msiren_reconstruct_tiles_dev)."""
    model._ensure_committed()
    d_t, host = _to_device(model, undersampled, 3)
    nv, nh = img_information[0]
    n = d_t.shape[0] // (nv * nh)
    O = model.outer_patch_size
    if d_t.shape != (n * nv * nh, O, O):  # (the device call reads O x O floats per tile)
        raise ValueError(f"tiles {d_t.shape} do not match image_information {img_information[0]} / {O}x{O} tiles")
    d_o = model.device_array((n, nv * model.inner_patch_size, nh * model.inner_patch_size))
    _lib.check(model._lib.msiren_reconstruct_tiles_dev(model._h, d_t.ptr, n, nv, nh, d_o.ptr))
    model.sync()
    return d_o.numpy() if host else d_o


def metrics_error(model, fully_sampled, undersampled, img_information, device, outer_patch_size, inner_patch_size,
                  siren_patch_size):
    """PSNR / SSIM / NRMSE of the reconstruction against the folded fully-sampled tiles.  error.py:200-271."""
    if siren_patch_size != model.siren_patch_size:
        raise ValueError(f"siren_patch_size {siren_patch_size} differs from the model's {model.siren_patch_size}")
    bind(model)
    d_under, _ = _to_device(model, undersampled, 3)   # numpy tiles are uploaded here, once; DeviceArrays stay where they are
    d_full, _ = _to_device(model, fully_sampled, 3)
    rec = reconstruct_from_patches(model, d_under, img_information).numpy()[0]
    full = patches_to_image(d_full, img_information, outer_patch_size, inner_patch_size, model=model).numpy()[0]
    return calculate_psnr(full, rec), calculate_ssim(full, rec), calculate_nrmse(full, rec)


def visual_error(model, output_dir, filename, fully_sampled, undersampled, img_information, device, outer_patch_size,
                 inner_patch_size, siren_patch_size):
    """The reconstruction of one slice next to what it came from, as images.  error.py:104-183: black-tile filter ->
    model -> zeros re-inserted -> weighted fold; the undersampled and fully-sampled tiles folded with the plain overlap
    average; then ``{filename}_reconstructed / _undersampled / _fully_sampled / _difference / _comparison`` under
    ``output_dir`` and ``{filename}_error.txt`` with the slice's PSNR / SSIM / NRMSE (error.py:160-197).  Every array is written as ``.npy``
    (what a test can check) and, where matplotlib is importable, as ``.png`` through the mirrors of the reference's ``save_image`` /
    ``save_image_comparison`` (metric_plots.py: min-max scaled, colour bar, the difference |fully - reconstructed| in viridis)."""
    import os

    if siren_patch_size != model.siren_patch_size:
        raise ValueError(f"siren_patch_size {siren_patch_size} differs from the model's {model.siren_patch_size}")
    bind(model)
    d_under, _ = _to_device(model, undersampled, 3)
    d_full, _ = _to_device(model, fully_sampled, 3)
    rec = reconstruct_from_patches(model, d_under, img_information).numpy()[0]
    under = patches_to_image(d_under, img_information, outer_patch_size, inner_patch_size, model=model).numpy()[0]
    full = patches_to_image(d_full, img_information, outer_patch_size, inner_patch_size, model=model).numpy()[0]
    images = {"reconstructed": rec, "undersampled": under, "fully_sampled": full, "difference": np.abs(full - rec)}
    os.makedirs(output_dir, exist_ok=True)
    for kind, a in images.items():
        np.save(os.path.join(output_dir, f"{filename}_{kind}.npy"), a)
    # the three scores of the slice, as the reference writes them (error.py:185-197)
    with open(os.path.join(output_dir, f"{filename}_error.txt"), "w") as f:
        f.write(f"PSNR: {calculate_psnr(full, rec)}\n")
        f.write(f"SSIM: {calculate_ssim(full, rec)}\n")
        f.write(f"NRMSE: {calculate_nrmse(full, rec)}\n")
    try:
        from . import metric_plots as mp

        mp._pyplot()
    except ImportError:
        return images
    for kind, a in images.items():   # save_image x 4 (the difference in viridis), then save_image_comparison: error.py:160-183
        mp.save_image(a, f"{filename}_{kind}", output_dir, cmap="viridis" if kind == "difference" else "gray")
    mp.save_image_comparison(full, under, rec, os.path.join(output_dir, f"{filename}_comparison"))
    return images
