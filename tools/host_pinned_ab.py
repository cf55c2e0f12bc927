#!/usr/bin/env python3
"""numpy -> numpy call of B tiles: pageable arrays against page-locked ones (model.pinned_empty / pin_outputs), and the chunk plan with them."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, synthetic as syn  # noqa: E402

sd = syn.make_state_dict(seed=7, trained_like=True)
for B in (128, 400, 800):
    t = np.random.default_rng(0).random((B, 32, 32), dtype=np.float32)
    for name, env, pinned in (("pageable", {}, False), ("pinned, one chunk", {"MSIREN_HOST_CHUNKS": "1"}, True), ("pinned, first 112", {}, True),
                              ("pinned, first 56", {"MSIREN_HOST_FIRST": "56"}, True), ("pinned, first 168", {"MSIREN_HOST_FIRST": "168"}, True)):
        for k in ("MSIREN_HOST_CHUNKS", "MSIREN_HOST_FIRST"):
            os.environ.pop(k, None)
        os.environ.update(env)
        m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
        m.load_state_dict(sd); m.to("cuda")
        x = t
        if pinned:
            x = m.pinned_empty(t.shape); x[...] = t
            m.pin_outputs(True)
        for _ in range(20):
            m(x)
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            out = m(x)
        dt = (time.perf_counter() - t0) / n
        print(f"B={B:4d} {name:20s}: {dt * 1e6:7.0f} us per call = {B * 256 / dt / 1e6:.1f} Mpixel/s", flush=True)
        del out, m
