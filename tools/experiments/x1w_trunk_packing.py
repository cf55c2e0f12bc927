"""Config 5: how well do back-to-back launches of the weight-stationary trunk pack when nothing else is queued between them?
Trunk-only calls (msiren_forward_mods_dev) on one stream and alternating over two; 320x320 slices of 400 tiles."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

H, L, Z = 512, 10, 128
sd = syn.make_state_dict(seed=21, dim_hidden=H, num_layers=L, latent_dim=Z, with_encoder=False)
sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
m = ModulatedSiren(2, H, 1, L, Z, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine", residual=True, precision="bf16")
m.load_state_dict(sd, strict=False); m.to("cuda")
for slices in (1, 2):
    B = 400 * slices
    mods = syn.make_mods(3, L, B, H, lo=0.1, hi=0.6)
    d_m = m.device_array(mods.shape).copy_from(mods)
    d_o = [m.device_array((B, 24, 24)) for _ in range(2)]
    for streams in (1, 2):
        _lib.check(m._lib.msiren_set_streams(m._h, streams))
        for _ in range(30): _lib.check(m._lib.msiren_forward_mods_dev(m._h, d_m.ptr, B, d_o[0].ptr))
        m.sync()
        ms = C.c_float(); n = 300 // slices
        _lib.check(m._lib.msiren_timer_start(m._h))
        for k in range(n): _lib.check(m._lib.msiren_forward_mods_dev(m._h, d_m.ptr, B, d_o[k & 1].ptr))
        _lib.check(m._lib.msiren_timer_stop(m._h, C.byref(ms)))
        per = ms.value / n
        print(f"{slices} slice(s) per call, {streams} stream(s): {per:.4f} ms per call = {slices * 0.1024 / per:.1f} Mpixel/s  ({m.last_trunk_kernel()})")
