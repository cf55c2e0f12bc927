#!/usr/bin/env python3
"""Slice -> slice (the reference's metrics_error pattern, src/util/error.py:231-249): one synchronous call per 320 x 320 slice.
Device-resident call + sync against the host-pointer call (numpy -> numpy), with and without a brain mask, by MSIREN_RECON_ZC (0 = the
staged copies of rounds 1-4; 1 = reconstruction stored in place; 2 = image read in place; 4 = image by DMA from page-locked pages)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn  # noqa: E402

sd = syn.make_state_dict(seed=7, trained_like=True)
for mask in (False, True):
    img = syn.make_slice(3, 320, 320, brain_mask=mask)
    for name, env in (("host call (default)", {}), ("host call, MSIREN_TILING_FUSED=0", {"MSIREN_TILING_FUSED": "0"}),
                      ("host call, MSIREN_RECON_ZC=0 (staged copies)", {"MSIREN_RECON_ZC": "0"}),
                      ("device call + sync", None), ("device call + sync, MSIREN_TILING_FUSED=0", {"MSIREN_TILING_FUSED": "0", "dev": "1"})):
        os.environ.pop("MSIREN_TILING_FUSED", None)
        os.environ.pop("MSIREN_RECON_ZC", None)
        dev = env is None or "dev" in env
        os.environ.update({k: v for k, v in (env or {}).items() if k != "dev"})
        m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
        m.load_state_dict(sd); m.to("cuda")
        if dev:
            d_i = m.device_array((1, 320, 320)).copy_from(img[None])
            d_o = m.device_array((1, 320, 320))
            def call():
                _lib.check(m._lib.msiren_reconstruct_slices_dev(m._h, d_i.ptr, 1, 320, 320, d_o.ptr))
                m.sync()
        else:
            def call():
                return m.reconstruct(img)
        for _ in range(30):
            call()
        n = 400
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        dt = (time.perf_counter() - t0) / n
        print(f"mask={int(mask)} {name:46s}: {dt * 1e6:7.1f} us per slice = {320 * 320 / dt / 1e6:.1f} Mpixel/s", flush=True)
        del m
