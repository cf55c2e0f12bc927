#!/usr/bin/env python3
"""Host-side timeline of the synchronous host-pointer call (numpy in -> numpy out, 400 tiles) and its rate for several cuts of
the batch over the two streams (MSIREN_HOST_SPLIT = percent of the tiles in the first chunk; MSIREN_TRACE_HOST=1 prints the
timeline of each call on stderr)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, synthetic as syn  # noqa: E402

sd = syn.make_state_dict(seed=7, trained_like=True)
t = np.random.default_rng(0).random((400, 32, 32), dtype=np.float32)
for split in (sys.argv[1:] or ["50", "40", "30", "25", "20", "15"]):
    os.environ["MSIREN_HOST_SPLIT"] = split
    os.environ["MSIREN_TRACE_HOST"] = "0"
    m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
    m.load_state_dict(sd)
    m.to("cuda")
    for _ in range(10):
        m(t)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        m(t)
    dt = (time.perf_counter() - t0) / n
    print(f"first chunk {split} %: {dt * 1e6:.0f} us per call = {102400 / dt / 1e6:.1f} Mpixel/s", flush=True)
    os.environ["MSIREN_TRACE_HOST"] = "1"
    m2 = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
    m2.load_state_dict(sd)
    m2.to("cuda")
    for _ in range(4):
        m2(t)
    sys.stderr.flush()
