#!/usr/bin/env python3
"""Synchronous host-pointer call (numpy in, numpy out: what the reference's metrics_error does per slice)
for 400 tiles, as a function of MSIREN_HOST_CHUNKS (read once at msiren_create: one model per setting)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, synthetic as syn  # noqa: E402

sd = syn.make_state_dict(seed=7, trained_like=True)
t = np.random.default_rng(0).random((400, 32, 32), dtype=np.float32)
img = syn.make_slice(3)
for chunks in (sys.argv[1:] or ["1", "2", "3", "4", "6"]):
    os.environ["MSIREN_HOST_CHUNKS"] = chunks
    m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
    m.load_state_dict(sd)
    m.to("cuda")
    ref = m(t)
    for _ in range(5):
        m(t)
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        out = m(t)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        rec = m.reconstruct(img)
    dr = (time.perf_counter() - t0) / n
    print(f"chunks={chunks}: model(tiles) {dt * 1e6:.0f} us per 400-tile call; model.reconstruct(slice) {dr * 1e6:.0f} us", flush=True)
