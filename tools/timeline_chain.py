#!/usr/bin/env python3
"""Stage timeline of the modulator chain launch (msiren_chain_timeline): per stage, when the first / last workgroup finished it
(us after the first workgroup started; s_memrealtime, 100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn  # noqa: E402

os.environ["MSIREN_CHAIN"] = "1"  # opt-in, read when the handle is created
sd = syn.make_state_dict(seed=7, trained_like=True)
m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
m.load_state_dict(sd)
m.to("cuda")
for B in [int(a) for a in sys.argv[1:]] or [1, 16, 400, 3200]:
    t = np.random.default_rng(0).random((B, 32, 32), dtype=np.float32)
    d_t = m.device_array(t.shape).copy_from(t)
    d_o = m.device_array((B, 24, 24))
    for _ in range(3):
        _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_t.ptr, B, d_o.ptr))
    m.sync()
    st = np.zeros((256, 17), dtype=np.uint64)
    _lib.check(m._lib.msiren_chain_timeline(m._h, d_t.ptr, B, d_o.ptr, st.ctypes.data_as(C.c_void_p)))
    ran = st[:, 0] > 0
    t0 = st[ran, 0].min()
    us = lambda x: (float(x) - float(t0)) / 100.0
    print(f"B={B}: {ran.sum()} workgroups with rows; start spread {us(st[ran, 0].max()):.2f} us")
    for s in range(16):
        col = st[ran, 1 + s]
        col = col[col > 0]
        if len(col) == 0:
            continue
        print(f"  stage {s}: {len(col):3d} workgroups, done first {us(col.min()):6.2f}  last {us(col.max()):6.2f} us")
