#!/usr/bin/env python3
"""Slot breakdown of the weight-stationary f16x3 trunk from its stamped diagnostic build (msiren_f16x3w_timeline)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sd = syn.make_state_dict(seed=7, trained_like=True)
m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine", precision="f16x3")
m.load_state_dict(sd)
m.to("cuda")
mods = syn.make_mods(1, 5, B, 256)
d_m = m.device_array(mods.shape).copy_from(mods)
d_o = m.device_array((B, 24, 24))
grid = min(256, (B * 18 + 1) // 2)
st = np.zeros((grid, 96, 8), dtype=np.uint64)
for _ in range(3):
    _lib.check(m._lib.msiren_f16x3w_timeline(m._h, d_m.ptr, B, d_o.ptr, st.ctypes.data))
t = st.astype(np.int64)
valid = t[:, :, 2] != 0
print("slots recorded per WG:", np.bincount(valid.sum(1))[-5:], "grid", grid)
book = (t[:, :, 1] - t[:, :, 0])
body = (t[:, :, 2] - t[:, :, 1])
gap = np.zeros_like(book)
gap[:, 1:] = t[:, 1:, 0] - t[:, :-1, 2]
n = min(96, int(valid.sum(1).min()))
print("slot  bookkeeping  body  boundary-before | this slot's boundary: mods staged, pass id, rest   (median cycles over workgroups; MFMA floor of a body: 3072)")
for i in range(min(n, 40)):
    b1, b2, b3 = (int(np.median(t[:, i, 4] - t[:, i, 2])), int(np.median(t[:, i, 5] - t[:, i, 4])), int(np.median(t[:, i, 6] - t[:, i, 5])))
    print(f"{i:3d} {int(np.median(book[:, i])):8d} {int(np.median(body[:, i])):8d} {int(np.median(gap[:, i])):8d} | {b1:6d} {b2:6d} {b3:6d}")
per = np.median(t[:, 17:33, 2] - t[:, 16:32, 2], axis=0)
print("slot period (slots 17..32):", [int(x) for x in per])
dm = t[:, n - 1, 2] - t[:, 16, 2]
dr = t[:, n - 1, 3] - t[:, 16, 3]
print("in-kernel clock %.0f MHz; mean slot period %.2f us = %.0f cycles" % (np.median(dm / dr) * 100, np.median(dr) / 100 / (n - 17), np.median(dm) / (n - 17)))
