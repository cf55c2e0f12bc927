"""numpy -> numpy calls of 2 ... 16 slices: ONE chunk with the caller's buffers page-locked and read / written in place against the cut plan with copies."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from mri_inr_amd import ModulatedSiren, synthetic as syn
sd = syn.make_state_dict(seed=7, trained_like=True)
for B in (800, 1200, 1600, 3200, 6400):
    t = np.random.default_rng(0).random((B, 32, 32), dtype=np.float32)
    for name, env in (("one chunk, in place", {"MSIREN_HOST_CHUNKS": "1"}), ("cut, copies", {})):
        os.environ.pop("MSIREN_HOST_CHUNKS", None); os.environ.update(env)
        m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
        m.load_state_dict(sd); m.to("cuda")
        for _ in range(10): m(t)
        n = 100 if B <= 1600 else 40
        t0 = time.perf_counter()
        for _ in range(n): m(t)
        dt = (time.perf_counter() - t0) / n
        print(f"B={B:5d} {name:22s}: {dt*1e6:8.0f} us = {B*256/dt/1e6:.1f} Mpixel/s", flush=True)
        del m
