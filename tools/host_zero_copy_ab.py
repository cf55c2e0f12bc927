#!/usr/bin/env python3
"""numpy -> numpy call on page-locked arrays: the trunk storing straight into the caller's output (MSIREN_ZC_OUT) and the conv kernel reading the
caller's tiles in place (MSIREN_ZC_IN) against the copies."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, synthetic as syn  # noqa: E402

sd = syn.make_state_dict(seed=7, trained_like=True)
ref = None
for B in (400, 3200):
    t = np.random.default_rng(0).random((B, 32, 32), dtype=np.float32)
    for zo, zi in ((0, 0), (1, 0), (0, 1), (1, 1)):
        os.environ["MSIREN_ZC_OUT"], os.environ["MSIREN_ZC_IN"] = str(zo), str(zi)
        m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
        m.load_state_dict(sd); m.to("cuda")
        x = m.pinned_empty(t.shape); x[...] = t
        m.pin_outputs(True)
        for _ in range(20):
            out = m(x)
        if zo == 0 and zi == 0:
            ref = out.copy()
        same = bool(np.array_equal(out, ref))
        n = 300 if B <= 800 else 60
        t0 = time.perf_counter()
        for _ in range(n):
            out = m(x)
        dt = (time.perf_counter() - t0) / n
        print(f"B={B:5d} out in place {zo} tiles in place {zi}: {dt * 1e6:7.0f} us per call = {B * 256 / dt / 1e6:.1f} Mpixel/s  same bits {same}", flush=True)
        del out, m
