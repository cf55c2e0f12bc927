#!/usr/bin/env python3
"""Where a single launch of the weight-stationary f16x3 trunk spends its time at the LAUNCH level (a
single-slice launch reads 0.54 of the roofline, an 8-slice launch 0.58): dispatch skew, per-workgroup prologue ("ramp"),
steady slots, tail.  Uses the stamped diagnostic instance (msiren_f16x3w_timeline): s_memrealtime marks (100 MHz) at
workgroup entry, at the end of the prologue and at workgroup exit, plus the per-slot stamps.

    python tools/timeline_ws_launch.py [slices=1]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

slices = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = 400 * slices
sd = syn.make_state_dict(seed=7, trained_like=True)
m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine", precision="f16x3")
m.load_state_dict(sd)
m.to("cuda")
mods = syn.make_mods(1, 5, B, 256)
d_m = m.device_array(mods.shape).copy_from(mods)
d_o = m.device_array((B, 24, 24))
grid = min(256, (B * 18 + 1) // 2)
st = np.zeros((grid, 96, 8), dtype=np.uint64)
# warm the card with the product kernel, then stamp a few launches and keep the last
for _ in range(300):
    _lib.check(m._lib.msiren_forward_mods_dev(m._h, d_m.ptr, B, d_o.ptr))
m.sync()
for _ in range(5):
    _lib.check(m._lib.msiren_f16x3w_timeline(m._h, d_m.ptr, B, d_o.ptr, st.ctypes.data))
t = st.astype(np.int64)
entry, pro_end, done = t[:, 0, 7], t[:, 1, 7], t[:, 2, 7]       # 10 ns ticks
t0 = entry.min()
us = lambda x: x / 100.0
units = B * 18
print(f"{slices} slice(s): {units} units on {grid} workgroups = {units / grid:.2f} units per workgroup")
print(f"launch (first workgroup entry -> last workgroup exit)       {us(done.max() - t0):8.2f} us")
print(f"dispatch skew (last entry - first entry)                    {us(entry.max() - t0):8.2f} us   (median entry at {us(np.median(entry) - t0):.2f})")
print(f"prologue per workgroup (entry -> first slot)   median       {us(np.median(pro_end - entry)):8.2f} us   max {us((pro_end - entry).max()):.2f}")
work = done - pro_end
print(f"slots + drain per workgroup                    median       {us(np.median(work)):8.2f} us   min {us(work.min()):.2f} max {us(work.max()):.2f}")
print(f"exit spread (last exit - median exit)                       {us(done.max() - np.median(done)):8.2f} us   (first exit at {us(done.min() - t0):.2f})")
# steady slot period from the per-slot realtime stamps (index 3 = s_memrealtime at slot end)
valid = t[:, :, 3] != 0
n = int(valid.sum(1).min())
if n > 20:
    per = (t[:, n - 1, 3] - t[:, 8, 3]) / (n - 9)
    print(f"steady slot period (slots 8..{n - 1})               median       {us(np.median(per)):8.3f} us")
    ideal = units * 4 / grid * us(np.median(per))
    print(f"units x 4 layers / workgroups x slot period   (no ramp, no tail) {ideal:8.2f} us  -> launch / ideal = {us(done.max() - t0) / ideal:.3f}")
