#!/usr/bin/env python3
"""Trunk time vs number of layers (f16x3 / fp32): separates per-layer cost from per-pass overhead."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
res = []
for L in (2, 3, 5, 7, 9, 13):
    sd = syn.make_state_dict(seed=7, num_layers=L, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    m = ModulatedSiren(2, 256, 1, L, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine", precision=prec)
    m.load_state_dict(sd, strict=False)
    m.to("cuda")
    mods = syn.make_mods(1, L, B, 256)
    d_m = m.device_array(mods.shape).copy_from(mods)
    d_o = m.device_array((B, 24, 24))
    for _ in range(3):
        _lib.check(m._lib.msiren_forward_mods_dev(m._h, d_m.ptr, B, d_o.ptr))
    m.sync()
    _lib.check(m._lib.msiren_timer_start(m._h))
    n = 10
    for _ in range(n):
        _lib.check(m._lib.msiren_forward_mods_dev(m._h, d_m.ptr, B, d_o.ptr))
    ms = C.c_float()
    _lib.check(m._lib.msiren_timer_stop(m._h, C.byref(ms)))
    res.append((L, ms.value / n))
    print(f"L={L:2d}  {ms.value / n:8.4f} ms")
for (l0, t0), (l1, t1) in zip(res, res[1:]):
    print(f"  per hidden layer between L={l0} and L={l1}: {(t1 - t0) / (l1 - l0) * 1e3:7.1f} us")
