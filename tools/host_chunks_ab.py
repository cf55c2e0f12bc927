#!/usr/bin/env python3
"""numpy -> numpy call of 400 tiles: one chunk on one stream (weight-stationary trunk) against the two-chunk cut over both streams
(register-resident trunk), three handles each (the rate differs from handle to handle: stream -> hardware-queue placement)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, synthetic as syn  # noqa: E402

sd = syn.make_state_dict(seed=7, trained_like=True)
t = np.random.default_rng(0).random((400, 32, 32), dtype=np.float32)
keep = []
for rep in range(3):
    for chunks in ("1", "2"):
        os.environ["MSIREN_HOST_CHUNKS"] = chunks
        m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
        m.load_state_dict(sd)
        m.to("cuda")
        for _ in range(20):
            m(t)
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            out = m(t)
        dt = (time.perf_counter() - t0) / n
        print(f"rep {rep} chunks={chunks}: {dt * 1e6:.0f} us per call = {102400 / dt / 1e6:.1f} Mpixel/s", flush=True)
        if "--keep" in sys.argv:
            keep.append(m)
        else:
            del m
