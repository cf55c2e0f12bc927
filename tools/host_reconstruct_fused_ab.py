import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from mri_inr_amd import ModulatedSiren, synthetic as syn
sd = syn.make_state_dict(seed=7, trained_like=True)
def mk(v):
    os.environ["MSIREN_TILING_FUSED"] = v
    m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
    m.load_state_dict(sd); m.to("cuda"); return m
ms = {"fused": mk("1"), "separate": mk("0")}
for mask in (0, 1):
    img = syn.make_slice(3, 320, 320, brain_mask=bool(mask))
    res = {k: [] for k in ms}
    for rep in range(6):
        for k, m in ms.items():
            for _ in range(30): m.reconstruct(img)
            t0 = time.perf_counter()
            for _ in range(500): m.reconstruct(img)
            res[k].append((time.perf_counter() - t0) / 500 * 1e6)
    for k in ms: print(f"mask={mask} {k:9s}: " + " ".join(f"{x:6.1f}" for x in res[k]) + f"   median {np.median(res[k]):6.1f} us", flush=True)
