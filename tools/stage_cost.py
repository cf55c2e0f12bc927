#!/usr/bin/env python3
"""What the encoder + modulator cost next to the persistent trunk: times, per 400-tile slice, the
two-stream pipeline through (a) msiren_forward_tiles_dev, (b) msiren_forward_latent_dev (no encoder),
(c) msiren_forward_mods_dev (trunk only).  Usage: python tools/stage_cost.py [slices_per_call]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn  # noqa: E402

n_sl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd = syn.make_state_dict(seed=7, trained_like=True)
m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0,
                   use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                   outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda:0", activation="sine")
m.load_state_dict(sd)
m.to("cuda:0").eval()
lib, h = m._lib, m._h
B = 400 * n_sl
tiles = np.random.default_rng(0).random((B, 32, 32), dtype=np.float32)
d_tiles = m.device_array(tiles.shape).copy_from(tiles)
d_z = m.device_array((B, 256))
d_mods = m.device_array((5, B, 256))
d_out = [m.device_array((B, 24, 24)) for _ in range(2)]
# latents and mods of these tiles, produced by the library itself
_lib.check(lib.msiren_forward_tiles_dev(h, d_tiles.ptr, B, d_out[0].ptr))
m.sync()
# inputs of the partial entry points: timing only, so seeded synthetic latents / modulations (same shapes and
# ranges as the model's own) -- the oracle is not imported outside tests/ and bench.py
zz = (np.random.default_rng(1).standard_normal((B, 256)) * 0.3).astype(np.float32)
mods = syn.make_mods(5, 5, B, 256)
d_z.copy_from(zz)
d_mods.copy_from(mods)


def run(name, fn, streams, steps=300):
    _lib.check(lib.msiren_set_streams(h, streams))
    for i in range(20):
        fn(i)
    m.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    m.sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:34s} streams={streams}: {dt * 1e3 / n_sl:.4f} ms per slice  ({n_sl * 0.1024 / dt:.1f} Mpixel/s)", flush=True)


for streams in (1, 2):
    run("tiles  (encoder+modulator+trunk)", lambda i: _lib.check(lib.msiren_forward_tiles_dev(h, d_tiles.ptr, B, d_out[i & 1].ptr)), streams)
    run("latent (modulator+trunk)", lambda i: _lib.check(lib.msiren_forward_latent_dev(h, d_z.ptr, B, d_out[i & 1].ptr, None)), streams)
    run("mods   (trunk only)", lambda i: _lib.check(lib.msiren_forward_mods_dev(h, d_mods.ptr, B, d_out[i & 1].ptr)), streams)
