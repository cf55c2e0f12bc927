#!/usr/bin/env python3
"""Synchronous host-pointer call (numpy in, numpy out: what the reference's metrics_error does per slice)
by call size: 1 / 4 / 8 slices per call (from 2 400 tiles the call pipelines itself over the handle's two streams), pageable and
page-locked tiles; model.reconstruct(slice) beside it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, synthetic as syn  # noqa: E402

sd = syn.make_state_dict(seed=7, trained_like=True)
t = np.random.default_rng(0).random((400, 32, 32), dtype=np.float32)
img = syn.make_slice(3)
m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
m.load_state_dict(sd)
m.to("cuda")
for slices in [int(x) for x in (sys.argv[1:] or ["1", "4", "8"])]:
    x = np.concatenate([t] * slices)
    pin = m.pinned_empty(x.shape)
    pin[...] = x
    for name, arr in (("pageable", x), ("page-locked", pin)):
        for _ in range(5):
            m(arr)
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            out = m(arr)
        dt = (time.perf_counter() - t0) / n
        print(f"{slices} slice(s) per call, {name} tiles: {dt * 1e6:.0f} us per call = {slices * 320 * 320 / dt / 1e6:.1f} Mpixel/s", flush=True)
    del pin
n = 100
m.reconstruct(img)
t0 = time.perf_counter()
for _ in range(n):
    rec = m.reconstruct(img)
dr = (time.perf_counter() - t0) / n
print(f"model.reconstruct(slice): {dr * 1e6:.0f} us per call", flush=True)
