#!/usr/bin/env python3
"""Run the stamped diagnostic trunk kernel (msiren_trunk_timeline) and print a phase breakdown."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sd = syn.make_state_dict(seed=7, trained_like=True)
if os.environ.get("ZERO_WEIGHTS"):
    sd = {k: (v if k == "grid" else np.zeros_like(v)) for k, v in sd.items()}
m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
m.load_state_dict(sd)
m.to("cuda")
mods = syn.make_mods(1, 5, B, 256)
d_m = m.device_array(mods.shape).copy_from(mods)
d_o = m.device_array((B, 24, 24))
grid = B * 9
st = np.zeros((grid, 32), dtype=np.uint64)
for _ in range(3):
    _lib.check(m._lib.msiren_trunk_timeline(m._h, d_m.ptr, B, d_o.ptr, st.ctypes.data))
hw, lds, xcc, rt0 = st[:, 0], st[:, 1], st[:, 2], st[:, 3].astype(np.int64)
t = st[:, 4:4 + 15].astype(np.int64)
names = ["start", "L0"] + sum([[f"K{l}", f"bar{l}", f"epi{l}"] for l in (1, 2, 3)], []) + ["K4", "bar4", "end"]
print("grid", grid, "distinct LDS_ALLOC values:", sorted(set(int(x) & 0xffffffff for x in lds))[:8])
print("HW_ID sample:", [hex(int(x)) for x in hw[:8]])
d = np.diff(t, axis=1)
first = np.argsort(rt0)[:512]
rest = np.argsort(rt0)[512:3000]
for sel, nm in ((first, "first-round WGs"), (rest, "steady-state WGs")):
    print(nm, "n=", len(sel), "total cycles median", int(np.median(t[sel, -1] - t[sel, 0])))
    for i, n in enumerate(names[1:]):
        print(f"   {n:6s} median {int(np.median(d[sel, i])):8d}   p10 {int(np.percentile(d[sel, i],10)):8d}  p90 {int(np.percentile(d[sel, i],90)):8d}")
span = (rt0.max() - rt0.min()) / 100.0  # memrealtime ticks at 100 MHz -> us
print("start-time span of all WGs (us):", span)
# occupancy over time: count of running WGs sampled on realtime axis is not available (end realtime not stamped)
rt1 = st[:, 31].astype(np.int64)
clk = (t[:, 13] - t[:, 0]) / np.maximum(rt1 - rt0, 1) * 100.0
print("in-kernel clock MHz (s_memtime / s_memrealtime): median %.0f p10 %.0f p90 %.0f" % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
t0 = rt0.min()
dur = (rt1 - rt0) / 100.0
print("WG duration us: median %.1f p10 %.1f p90 %.1f; kernel span %.1f us" % (np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), (rt1.max() - t0) / 100.0))
key = (xcc.astype(np.int64) << 40) | ((hw.astype(np.int64) >> 8) & 0xff) << 20 | (lds.astype(np.int64) & 0x1ff)
gaps, per_slot = [], []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(rt0[idx])]
    per_slot.append(len(idx))
    for a, b in zip(idx[:-1], idx[1:]):
        gaps.append((rt0[b] - rt1[a]) / 100.0)
gaps = np.array(gaps)
print("slots seen:", len(per_slot), "WGs/slot min/med/max", min(per_slot), int(np.median(per_slot)), max(per_slot))
print("gap between consecutive WGs on a slot (us): median %.2f p10 %.2f p90 %.2f max %.2f" % (np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90), gaps.max()))
busy = np.zeros(int((rt1.max() - t0) / 100) + 2)
for a, b in zip(rt0, rt1):
    busy[int((a - t0) / 100): int((b - t0) / 100) + 1] += 1
print("resident WGs over time (per 100us):", [int(busy[i:i + 100].mean()) for i in range(0, len(busy), 100)])
np.save("gpurun_out/timeline.npy", st)
