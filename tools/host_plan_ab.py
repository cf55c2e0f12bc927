#!/usr/bin/env python3
"""numpy -> numpy call of 400 tiles (and of 3200): the pipelined plan's first-chunk / piece sizes against one chunk (three handles each)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, synthetic as syn  # noqa: E402

sd = syn.make_state_dict(seed=7, trained_like=True)
B = int(os.environ.get("B", "400"))
t = np.random.default_rng(0).random((B, 32, 32), dtype=np.float32)
plans = [("one chunk", {"MSIREN_HOST_CHUNKS": "1"})] + [(f"first {f} piece {p}", {"MSIREN_HOST_FIRST": str(f), "MSIREN_HOST_PIECE": str(p)})
                                                         for f, p in ((56, 400), (84, 400), (112, 400), (140, 400), (56, 172), (112, 144))]
for rep in range(2):
    for name, env in plans:
        for k in ("MSIREN_HOST_CHUNKS", "MSIREN_HOST_FIRST", "MSIREN_HOST_PIECE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
        m.load_state_dict(sd); m.to("cuda")
        for _ in range(20):
            m(t)
        n = 300 if B <= 800 else 60
        t0 = time.perf_counter()
        for _ in range(n):
            m(t)
        dt = (time.perf_counter() - t0) / n
        print(f"rep {rep} {name:24s}: {dt * 1e6:7.0f} us per call = {B * 256 / dt / 1e6:.1f} Mpixel/s", flush=True)
        del m
