import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_chain import make, latent_dev, chain_info
from mri_inr_amd import synthetic as syn
sd = syn.make_state_dict(seed=7, trained_like=True)
on, off = make(sd, True), make(sd, False)
for B in (1, 16, 40):
    z = np.random.default_rng(B).standard_normal((B, 256)).astype(np.float32)
    (oa, ma), (ob, mb) = latent_dev(on, z), latent_dev(off, z)
    for l in range(5):
        d = np.abs(ma[l] - mb[l])
        bad = np.argwhere(ma[l] != mb[l])
        print(f"B={B} layer {l}: max diff {d.max():.3e}, differing {len(bad)} of {d.size}; first {bad[:4].tolist()}; on {ma[l].ravel()[:4]} off {mb[l].ravel()[:4]}")
    print(chain_info(on))
