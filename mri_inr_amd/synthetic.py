"""Platform-stable synthetic weights and slices (numpy RNG only).

There is no network for checkpoints or fastMRI data, so benchmarks, fixtures and
tests all draw their ``state_dict`` from here.  Key names and shapes are the
reference's (``ModulatedSiren.state_dict()``):

* ``net.layers.{l}.weight/bias``, ``net.last_layer.weight/bias``
  -- init ranges follow ``Siren.init_`` (src/networks/modulated_siren.py:126-142):
  ``U(+-1/dim_in)`` for the first layer, ``U(+-sqrt(6/dim_in)/w0)`` otherwise.
* ``modulator.layers.{l}.0.weight/bias`` -- ``nn.Linear`` default ``U(+-1/sqrt(fan_in))``
  (src/networks/modulated_siren.py:319-323).
* ``encoder.encoder.encoder.{0,2,4,7}.weight/bias`` -- ``FixedAutoencoder.encoder``
  (src/networks/encoding/siren_encoder.py:503-512), torch default init range.
* ``grid`` -- the registered coordinate buffer (src/networks/modulated_siren.py:427-433).
"""

from __future__ import annotations

import math

import numpy as np


def make_grid(siren_patch_size: int) -> np.ndarray:
    """``grid[h*S+w] = (lin[h], lin[w])`` with ``lin = linspace(-1, 1, S)``.

    Reference: src/networks/modulated_siren.py:427-433 (meshgrid indexing="ij").
    Computed the way torch.linspace does in fp32 (start + i*step for the first half,
    end - (S-1-i)*step for the second); agrees with torch's CPU result to 1.2e-7 (one ulp at 1; torch's own
    last bit depends on the host's SIMD width).  Real checkpoints carry the buffer in their state_dict.
    """
    S = int(siren_patch_size)
    lin = torch_like_linspace(-1.0, 1.0, S)
    hh, ww = np.meshgrid(lin, lin, indexing="ij")
    return np.stack([hh, ww], axis=-1).reshape(S * S, 2).astype(np.float32)


def torch_like_linspace(start: float, end: float, steps: int) -> np.ndarray:
    """fp32 linspace with torch's symmetric evaluation order (ATen RangeFactories)."""
    if steps == 1:
        return np.array([start], dtype=np.float32)
    step = np.float32((np.float32(end) - np.float32(start)) / np.float32(steps - 1))
    out = np.empty(steps, dtype=np.float32)
    half = steps // 2
    for i in range(steps):
        if i < half:
            out[i] = np.float32(start) + step * np.float32(i)
        else:
            out[i] = np.float32(end) - step * np.float32(steps - i - 1)
    return out


def _uniform(rng: np.random.Generator, bound: float, shape) -> np.ndarray:
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def make_state_dict(
    seed: int = 7,
    dim_in: int = 2,
    dim_hidden: int = 256,
    dim_out: int = 1,
    num_layers: int = 5,
    latent_dim: int = 256,
    w0: float = 1.0,
    siren_patch_size: int = 24,
    outer_patch_size: int = 32,
    use_bias: bool = True,
    modulator_gain: float = 1.0,
    modulator_bias_center: float = 0.0,
    encoder_gain: float = 1.0,
    with_encoder: bool = True,
    trained_like: bool = False,
) -> dict:
    """Random-init ``state_dict`` with the reference's key names, shapes and init ranges.

    With the torch default init the modulator's ReLU outputs sit in [0, 0.17] and the network
    output is nearly constant (SURVEY.md §7 "degenerate test").  ``trained_like=True`` is the
    preset used for non-degenerate parity cases: modulator biases centred on 1
    (``modulator_bias_center=1``) and the encoder's last linear scaled x10
    (``encoder_gain=10``), which gives modulations ~ relu(1 + N(0, 0.5)) that vary per patch
    and a network output spanning [-1, 1].  The RNG stream is identical in both modes.
    """
    if trained_like:
        modulator_bias_center, encoder_gain = 1.0, 10.0
    rng = np.random.default_rng(seed)
    sd: dict[str, np.ndarray] = {}
    sd["grid"] = make_grid(siren_patch_size)
    for l in range(num_layers):
        k = dim_in if l == 0 else dim_hidden
        bound = (1.0 / k) if l == 0 else math.sqrt(6.0 / k) / w0
        sd[f"net.layers.{l}.weight"] = _uniform(rng, bound, (dim_hidden, k))
        if use_bias:
            sd[f"net.layers.{l}.bias"] = _uniform(rng, bound, (dim_hidden,))
    bound = math.sqrt(6.0 / dim_hidden) / w0
    sd["net.last_layer.weight"] = _uniform(rng, bound, (dim_out, dim_hidden))
    if use_bias:
        sd["net.last_layer.bias"] = _uniform(rng, bound, (dim_out,))
    for l in range(num_layers):
        k = latent_dim if l == 0 else dim_hidden + latent_dim
        bound = modulator_gain / math.sqrt(k)
        sd[f"modulator.layers.{l}.0.weight"] = _uniform(rng, bound, (dim_hidden, k))
        sd[f"modulator.layers.{l}.0.bias"] = (
            _uniform(rng, bound, (dim_hidden,)) + np.float32(modulator_bias_center))
    if with_encoder:
        enc = make_encoder_state_dict(rng, latent_dim, outer_patch_size)
        if encoder_gain != 1.0:
            for k in ("encoder.encoder.encoder.7.weight", "encoder.encoder.encoder.7.bias"):
                enc[k] = (enc[k] * np.float32(encoder_gain)).astype(np.float32)
        sd.update(enc)
    return sd


def make_encoder_state_dict(rng, latent_dim: int = 256, outer_patch_size: int = 32) -> dict:
    """``FixedAutoencoder.encoder`` parameters (siren_encoder.py:503-512), key prefix as seen
    from ``ModulatedSiren`` (``encoder.`` -> Encoder, ``encoder.`` -> FixedEncoder,
    ``encoder.`` -> nn.Sequential)."""
    if isinstance(rng, (int, np.integer)):
        rng = np.random.default_rng(rng)
    if outer_patch_size != 32:
        raise ValueError("FixedAutoencoder is hard-wired to 32x32 tiles (siren_encoder.py:499)")
    p = "encoder.encoder.encoder."
    sd = {}
    shapes = {
        "0": (16, 1, 3, 3),
        "2": (32, 16, 3, 3),
        "4": (64, 32, 8, 8),
    }
    for idx, shp in shapes.items():
        fan_in = shp[1] * shp[2] * shp[3]
        b = 1.0 / math.sqrt(fan_in)
        sd[p + idx + ".weight"] = _uniform(rng, b, shp)
        sd[p + idx + ".bias"] = _uniform(rng, b, (shp[0],))
    b = 1.0 / math.sqrt(64)
    sd[p + "7.weight"] = _uniform(rng, b, (latent_dim, 64))
    sd[p + "7.bias"] = _uniform(rng, b, (latent_dim,))
    return sd


def make_slice(k: int, height: int = 320, width: int = 320, brain_mask: bool = False) -> np.ndarray:
    """Synthetic slice ``k``: ``default_rng(1000+k).random((H, W), float32)``.

    With ``brain_mask`` the image is multiplied by an elliptical support so that corner patches
    are exactly zero and the black-patch filter (src/util/tiling.py:184-198) is exercised.
    """
    img = np.random.default_rng(1000 + k).random((height, width), dtype=np.float32)
    if brain_mask:
        yy, xx = np.mgrid[0:height, 0:width]
        cy, cx = (height - 1) / 2.0, (width - 1) / 2.0
        ell = ((yy - cy) / (0.40 * height)) ** 2 + ((xx - cx) / (0.33 * width)) ** 2
        img = img * (ell <= 1.0).astype(np.float32)
    return img


def make_mods(seed: int, num_layers: int, batch: int, dim_hidden: int,
              lo: float = 0.5, hi: float = 1.5, zero_fraction: float = 0.0) -> np.ndarray:
    """Direct modulations ``(L, B, H)`` ~ U(lo, hi) (config 1's "random latent" regime), with an
    optional fraction of exact zeros (the modulator ends in ReLU, so zeros are common)."""
    rng = np.random.default_rng(seed)
    m = rng.uniform(lo, hi, size=(num_layers, batch, dim_hidden)).astype(np.float32)
    if zero_fraction > 0:
        m[rng.random(m.shape) < zero_fraction] = 0.0
    return m
