#!/usr/bin/env python3
"""Where a steady-state SLOT of the weight-stationary f16x3 trunk spends its cycles (round 4): the stamped diagnostic
instance (msiren_f16x3w_timeline) records per slot s_memtime at  0 slot start | 1 body start (unit bookkeeping, the next
pass's modulation fetch done) | 2 body end (192 MFMAs + the gap schedule) | 4 modulation rows stored | 5 pass-id pipeline |
6 slot end, and s_memrealtime (100 MHz) at 3.  Core cycles per phase by slot flavour, and the clock the chip held.

    python tools/timeline_ws_slot.py [slices=8]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

slices = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
for _ in range(300):
    _lib.check(m._lib.msiren_forward_mods_dev(m._h, d_m.ptr, B, d_o.ptr))
m.sync()
for _ in range(5):
    _lib.check(m._lib.msiren_f16x3w_timeline(m._h, d_m.ptr, B, d_o.ptr, st.ctypes.data))
t = st.astype(np.int64)
# slots 16..95 of every workgroup: steady state.  Slot k of a pass of 4 units x 4 layers: layer = 1 + (k // 4) % 4, unit = k % 4
# (MSIREN_WS_RUN order: for l: for b) -- flavour by position in the 16-slot pass
sl = np.arange(16, 95)
seg = {"setup 0->1": (0, 1), "body 1->2": (1, 2), "mods store 2->4": (2, 4), "pass id 4->5": (4, 5), "tail 5->6": (5, 6), "slot 0->6": (0, 6)}
print(f"{slices} slice(s), {grid} workgroups; cycles per slot phase (s_memtime), median over workgroups, steady slots 16..94")
k16 = sl % 16
names = {}
for k in range(16):
    names[k] = f"layer {1 + k // 4} unit {k % 4}"
hdr = "slot in pass".ljust(18) + "".join(n.rjust(18) for n in seg) + "   next-start gap"
print(hdr)
for k in range(16):
    idx = sl[k16 == k]
    row = names[k].ljust(18)
    for n, (a, b) in seg.items():
        row += f"{np.median(t[:, idx, b] - t[:, idx, a]):18.0f}"
    nxt = np.median(t[:, idx + 1, 0] - t[:, idx, 6])
    row += f"{nxt:18.0f}"
    print(row)
tot = np.median(t[:, 94, 0] - t[:, 16, 0]) / 78.0
rt = np.median(t[:, 94, 3] - t[:, 16, 3]) / 78.0
print(f"slot period {tot:.0f} cycles = {rt / 100:.3f} us -> clock {tot / rt * 100:.0f} MHz;  192 MFMAs x 16.08 = 3087 cycles")
