#!/usr/bin/env python3
"""Phase breakdown of the f16x3 trunk from its stamped diagnostic build (msiren_f16x3_timeline)."""
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
grid = min(256, (B * 18 + 3) // 4)
st = np.zeros((grid, 4, 48), dtype=np.uint64)
for _ in range(3):
    _lib.check(m._lib.msiren_f16x3_timeline(m._h, d_m.ptr, B, d_o.ptr, st.ctypes.data))
t = st.astype(np.int64)
valid = t[:, :, 6] != 0
# stamps: 0 pass start, 1 layer 0 done, 2 hidden layers done, 6 pass end (s_memtime), 7 pass end (s_memrealtime),
#         8..39 end of each of the 32 hidden-layer tiles
print("passes recorded:", int(valid.sum()), "per WG:", np.bincount(valid.sum(1)))
for n, a, b in (("loads+layer0", 0, 1), ("hidden layers", 1, 2), ("final+store", 2, 6)):
    x = (t[:, :, b] - t[:, :, a])[valid]
    print(f"  {n:14s} median {int(np.median(x)):7d}  p10 {int(np.percentile(x,10)):7d}  p90 {int(np.percentile(x,90)):7d} cycles")
tiles = np.diff(t[:, :, 8:40], axis=2)
print("  per-tile cycles (median over WGs/passes), tiles 1..31 of 4 layers x 8 (stamping itself costs ~20 %):")
med = np.median(tiles[valid], axis=0).astype(int)
print("   ", [int(v) for v in med])
tot = (t[:, :, 6] - t[:, :, 0])[valid]
print("  pass total     median", int(np.median(tot)), " (ideal MFMA: 49152)")
# clock: pass-end realtime differences between consecutive passes of the same WG
rt = t[:, :, 7]
ok = valid[:, 1:] & valid[:, :-1]
drt = (rt[:, 1:] - rt[:, :-1])[ok]
dmt = (t[:, 1:, 6] - t[:, :-1, 6])[ok]
print("  in-kernel clock (s_memtime/s_memrealtime): %.0f MHz; pass period %.1f us" % (np.median(dmt / drt) * 100, np.median(drt) / 100))
