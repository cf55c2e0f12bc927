"""CPU ORACLE -- test infrastructure only, never the product path.

A numpy restatement of the reference's modulated-SIREN hot path (MatteoWohlrapp/mri-inr,
``/root/reference``), pinned against outputs of the reference itself: ``oracle/gen_fixtures.py``
imports the reference in the build container, pushes identical weights and inputs through both
and commits the results under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks this
file against those fixtures on every run.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``mri_inr_amd``) never does: it fails loudly without its HIP library.

Every function cites the reference lines it restates.  ``dtype=np.float64`` gives the
high-precision twin used to size tolerances.

Config-5 note ("deep residual variant"): the residual model lives on a branch that is not in the
container (README.md:27-29), so ``residual=True`` below is this build's own definition and is
**parity-unpinned** against the reference.
"""

from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------------------------
# coordinates
# --------------------------------------------------------------------------------------------


def linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """``torch.linspace(start, end, steps)`` in fp32 as ATen evaluates it (RangeFactories: ``start + i*step`` below the midpoint,
    ``end - (steps-1-i)*step`` from it on, everything in fp32).  The oracle's OWN statement -- it does not borrow the product's helper
    (mri_inr_amd/synthetic.py); both are pinned by the reference's own buffer in tests/golden/reference_grid.npz."""
    f = np.float32
    if steps == 1:
        return np.array([start], dtype=np.float32)
    step = f((f(end) - f(start)) / f(steps - 1))
    i = np.arange(steps)
    lo = (f(start) + step * i.astype(np.float32)).astype(np.float32)
    hi = (f(end) - step * (steps - 1 - i).astype(np.float32)).astype(np.float32)
    return np.where(i < steps // 2, lo, hi).astype(np.float32)


def make_grid(S: int, dtype=np.float32) -> np.ndarray:
    """(S*S, 2) grid, row ``h*S+w`` = (lin[h], lin[w]).  Ref: modulated_siren.py:427-433."""
    lin = linspace_f32(-1.0, 1.0, S).astype(dtype)
    g = np.empty((S, S, 2), dtype=dtype)
    g[:, :, 0] = lin[:, None]
    g[:, :, 1] = lin[None, :]
    return g.reshape(S * S, 2)


# --------------------------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------------------------


def act_sine(p, w0):
    """``sin(w0 * x)``.  Ref: modulated_siren.py:44-54."""
    return np.sin(p.dtype.type(w0) * p)


def act_morlet(p, w0):
    """``sin(w0 * x) * exp(-0.5 * x**2)`` -- the Gaussian takes x, not w0*x.
    Ref: modulated_siren.py:70-80."""
    return np.sin(p.dtype.type(w0) * p) * np.exp(p.dtype.type(-0.5) * p * p)


# --------------------------------------------------------------------------------------------
# SIREN trunk (the hot loop)
# --------------------------------------------------------------------------------------------


def siren_forward(sd: dict, mods, *, num_layers: int, w0: float = 1.0, w0_initial: float = 30.0,
                  activation: str = "sine", siren_patch_size: int = 24, dtype=np.float32,
                  residual: bool = False, return_hidden: bool = False):
    """``SirenNet.forward`` over the fixed coordinate grid for every patch.

    mods: array (L, B, H) or sequence of L arrays (B, H); ``None`` = unmodulated (B=1).
    Returns (B, S*S) of ``dtype``.

    Ref: modulated_siren.py:215-233 (layer loop, in-place ``x *= mod`` after the activation),
    :144-157 (``F.linear`` then activation; dropout is identity in eval), :211-213 + :120-123
    (the last layer has ``activation=None`` and therefore *always* applies ``Sine(w0)``).
    """
    L = int(num_layers)
    grid = np.asarray(sd["grid"], dtype=dtype) if "grid" in sd else make_grid(siren_patch_size, dtype)
    if mods is None:
        B = 1
        mods_l = [None] * L
    else:
        mods_l = [np.asarray(m, dtype=dtype) for m in mods]
        B = mods_l[0].shape[0]
    act = act_morlet if activation == "morlet" else act_sine
    P = grid.shape[0]
    x = np.broadcast_to(grid[None], (B, P, grid.shape[1]))
    hidden = []
    for l in range(L):
        W = np.asarray(sd[f"net.layers.{l}.weight"], dtype=dtype)
        b = sd.get(f"net.layers.{l}.bias")
        p = x @ W.T
        if b is not None:
            p = p + np.asarray(b, dtype=dtype)
        a = act(p, w0_initial if l == 0 else w0)
        if mods_l[l] is not None:
            a = a * mods_l[l][:, None, :]
        if residual and l > 0:
            # build-defined (parity-unpinned): skip connection around every hidden layer but the first
            a = x + a
        x = a
        if return_hidden:
            hidden.append(x.copy())
    W = np.asarray(sd["net.last_layer.weight"], dtype=dtype)
    b = sd.get("net.last_layer.bias")
    p = x @ W.T
    if b is not None:
        p = p + np.asarray(b, dtype=dtype)
    out = act_sine(p, w0)  # always sine, also for Morlet models
    out = out[..., 0]  # dim_out == 1: squeeze(2) + rearrange, modulated_siren.py:451-455
    if return_hidden:
        return out, hidden
    return out


# --------------------------------------------------------------------------------------------
# modulator + encoder (producers of the modulation vectors)
# --------------------------------------------------------------------------------------------


def modulator_forward(sd: dict, z, *, num_layers: int, dtype=np.float32) -> np.ndarray:
    """``h0 = relu(M0 z + c0)``; ``hl = relu(Ml [h(l-1) ; z] + cl)`` (hidden first, latent second).
    Returns (L, B, H).  Ref: modulated_siren.py:325-343."""
    z = np.asarray(z, dtype=dtype)
    x = z
    outs = []
    for l in range(num_layers):
        W = np.asarray(sd[f"modulator.layers.{l}.0.weight"], dtype=dtype)
        b = np.asarray(sd[f"modulator.layers.{l}.0.bias"], dtype=dtype)
        h = np.maximum(x @ W.T + b, dtype(0))
        outs.append(h)
        x = np.concatenate([h, z], axis=1)
    return np.stack(outs, axis=0)


def _leaky(x, slope):
    return np.where(x >= 0, x, x * x.dtype.type(slope))


def _conv2d(x, W, b, stride, pad):
    """Direct NCHW convolution via a strided window view (cross-correlation, as torch)."""
    B, C, Hh, Ww = x.shape
    O, _, kh, kw = W.shape
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    Ho = (x.shape[2] - kh) // stride + 1
    Wo = (x.shape[3] - kw) // stride + 1
    s = x.strides
    win = np.lib.stride_tricks.as_strided(
        x, (B, C, Ho, Wo, kh, kw), (s[0], s[1], s[2] * stride, s[3] * stride, s[2], s[3]), writeable=False)
    cols = win.transpose(0, 2, 3, 1, 4, 5).reshape(B * Ho * Wo, C * kh * kw)
    y = cols @ W.reshape(O, -1).T + b
    return y.reshape(B, Ho, Wo, O).transpose(0, 3, 1, 2)


def encoder_forward(sd: dict, tiles, *, dtype=np.float32) -> np.ndarray:
    """Custom ("FixedEncoder") branch: (B,32,32) -> (B,latent).

    conv3x3 s2 p1 (1->16), LeakyReLU(0.2), conv3x3 s2 p1 (16->32), LeakyReLU, conv8x8 (32->64),
    LeakyReLU, flatten, Linear(64->latent); ``fc`` is Identity.
    Ref: siren_encoder.py:503-512, :565-577; modulated_siren.py:252-255, :294-300.
    """
    p = "encoder.encoder.encoder."
    g = lambda k: np.asarray(sd[p + k], dtype=dtype)
    x = np.asarray(tiles, dtype=dtype)[:, None, :, :]
    x = _leaky(_conv2d(x, g("0.weight"), g("0.bias"), 2, 1), 0.2)
    x = _leaky(_conv2d(x, g("2.weight"), g("2.bias"), 2, 1), 0.2)
    x = _leaky(_conv2d(x, g("4.weight"), g("4.bias"), 1, 0), 0.2)
    x = x.reshape(x.shape[0], -1)
    return x @ g("7.weight").T + g("7.bias")


def modulated_siren_forward(sd: dict, tiles, *, num_layers: int, w0: float = 1.0,
                            w0_initial: float = 30.0, activation: str = "sine",
                            siren_patch_size: int = 24, dtype=np.float32) -> np.ndarray:
    """``ModulatedSiren.forward``: tiles (B,O,O) -> (B,S,S).  Ref: modulated_siren.py:435-457."""
    z = encoder_forward(sd, tiles, dtype=dtype)
    mods = modulator_forward(sd, z, num_layers=num_layers, dtype=dtype)
    out = siren_forward(sd, mods, num_layers=num_layers, w0=w0, w0_initial=w0_initial,
                        activation=activation, siren_patch_size=siren_patch_size, dtype=dtype)
    S = siren_patch_size
    return out.reshape(out.shape[0], S, S)


# --------------------------------------------------------------------------------------------
# tiling (the steps either side of the hot path)
# --------------------------------------------------------------------------------------------


def image_to_patches(img, outer: int, inner: int):
    """(H,W) -> ((nV*nH, outer, outer), (nV, nH)).

    Reflect-pad by ``(outer-inner)//2`` on every side plus bottom/right up to a multiple of
    ``inner``, then take ``outer`` windows at stride ``inner``, row-major over (nV, nH).
    Ref: src/util/tiling.py:10-64.
    """
    img = np.asarray(img)
    Hh, Ww = img.shape
    pad = (outer - inner) // 2
    vpad = (inner - Hh % inner) % inner
    hpad = (inner - Ww % inner) % inner
    # torch's reflect padding (F.pad in tiling.py:40) raises unless every pad is smaller than the dimension;
    # numpy would silently reflect more than once
    if pad + vpad >= Hh or pad + hpad >= Ww:
        raise ValueError(f"image {Hh}x{Ww} is too small for reflect padding of {pad + vpad}/{pad + hpad}")
    padded = np.pad(img, ((pad, pad + vpad), (pad, pad + hpad)), mode="reflect")
    nV = (Hh + vpad) // inner
    nH = (Ww + hpad) // inner
    out = np.empty((nV * nH, outer, outer), dtype=img.dtype)
    for v in range(nV):
        for h in range(nH):
            out[v * nH + h] = padded[v * inner:v * inner + outer, h * inner:h * inner + outer]
    return out, (nV, nH)


def generate_weight_matrix(tile: int) -> np.ndarray:
    """``w[i,j] = exp(-0.1 * dist((i,j), centre))`` normalised to max 1 (fp64 maths, fp32 storage).
    Ref: src/util/tiling.py:67-88."""
    c = (tile - 1) / 2
    i, j = np.mgrid[0:tile, 0:tile]
    w = np.exp(-0.1 * np.sqrt((i - c) ** 2 + (j - c) ** 2)).astype(np.float32)
    return w / w.max()


def _fold(tiles, nV, nH, k, stride, pad, dtype):
    """Overlap-add of (nV*nH, k, k) tiles into (nV*stride, nH*stride) (``F.fold`` semantics)."""
    Hh, Ww = nV * stride, nH * stride
    acc = np.zeros((Hh + 2 * pad, Ww + 2 * pad), dtype=dtype)
    for v in range(nV):
        for h in range(nH):
            acc[v * stride:v * stride + k, h * stride:h * stride + k] += tiles[v * nH + h]
    return acc[pad:pad + Hh, pad:pad + Ww]


def patches_to_image_weighted_average(tiles, info, tile: int, inner: int) -> np.ndarray:
    """``fold(tiles*w) / fold(w)``; output (nV*inner, nH*inner).  Ref: src/util/tiling.py:91-140.
    Note the reference calls this with ``tile = siren_patch_size`` (error.py:243-249)."""
    nV, nH = info
    tiles = np.asarray(tiles, dtype=np.float32)
    w = generate_weight_matrix(tile)
    pad = (tile - inner) // 2
    num = _fold(tiles * w, nV, nH, tile, inner, pad, np.float32)
    den = _fold(np.broadcast_to(w, tiles.shape), nV, nH, tile, inner, pad, np.float32)
    return num / den


def patches_to_image(tiles, info, outer: int, inner: int) -> np.ndarray:
    """Plain overlap average.  Ref: src/util/tiling.py:143-181."""
    nV, nH = info
    tiles = np.asarray(tiles, dtype=np.float32)
    pad = (outer - inner) // 2
    num = _fold(tiles, nV, nH, outer, inner, pad, np.float32)
    den = _fold(np.ones_like(tiles), nV, nH, outer, inner, pad, np.float32)
    return num / den


def filter_and_remember_black_patches(patches):
    """Patches whose mean is < 1e-10 are "black" and skipped.  Ref: tiling.py:184-198, :244-271."""
    patches = np.asarray(patches)
    means = patches.reshape(patches.shape[0], -1).mean(axis=1, dtype=np.float32)
    black = [int(i) for i in np.nonzero(means < 1e-10)[0]]
    keep = [i for i in range(patches.shape[0]) if i not in set(black)]
    return patches[keep], black, patches.shape


def reintegrate_black_patches(processed, black, original_shape) -> np.ndarray:
    """Zeros at the black indices, processed tiles elsewhere, in order.  Ref: tiling.py:274-303."""
    processed = np.asarray(processed)
    full = np.zeros((original_shape[0],) + processed.shape[1:], dtype=processed.dtype)
    keep = [i for i in range(original_shape[0]) if i not in set(black)]
    full[keep] = processed
    return full


def extract_center_batch(batch, outer: int, inner: int) -> np.ndarray:
    """Centre ``inner`` x ``inner`` window of each tile.  Ref: src/util/tiling.py:306-322."""
    p = (outer - inner) // 2
    return np.asarray(batch)[:, p:p + inner, p:p + inner]


# --------------------------------------------------------------------------------------------
# whole-slice reconstruction as metrics_error does it (error.py:200-258, without the metrics)
# --------------------------------------------------------------------------------------------


def reconstruct_slice(sd: dict, img, *, num_layers: int, w0=1.0, w0_initial=30.0, activation="sine",
                      outer=32, inner=16, siren_patch_size=24, dtype=np.float32) -> np.ndarray:
    """image -> patches -> drop black -> model -> re-insert zeros -> weighted fold."""
    patches, info = image_to_patches(np.asarray(img, dtype=np.float32), outer, inner)
    kept, black, shape = filter_and_remember_black_patches(patches)
    if kept.shape[0]:
        rec = modulated_siren_forward(sd, kept, num_layers=num_layers, w0=w0, w0_initial=w0_initial,
                                      activation=activation, siren_patch_size=siren_patch_size,
                                      dtype=dtype).astype(np.float32)
    else:
        rec = np.zeros((0, siren_patch_size, siren_patch_size), np.float32)
    rec = reintegrate_black_patches(rec, black, shape)
    return patches_to_image_weighted_average(rec, info, siren_patch_size, inner)


def flops_per_coord(H: int, L: int, dim_in: int = 2) -> int:
    """Algorithmic FLOPs per coordinate: ``2*dim_in*H + (L-1)*2*H*H + 2*H``."""
    return 2 * dim_in * H + (L - 1) * 2 * H * H + 2 * H
