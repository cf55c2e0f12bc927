"""Host numpy -> numpy call latency by batch size (synchronous msiren_forward_tiles): median and spread of 200 calls."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from mri_inr_amd import ModulatedSiren, synthetic as syn
sd = syn.make_state_dict(seed=7, trained_like=True)
m = ModulatedSiren(2,256,1,5,256,1.0,30.0,True,0.1,True,"custom",None,32,16,24,"cuda","sine")
m.load_state_dict(sd); m.to("cuda")
for B in (1, 2, 3, 4, 6, 8, 12, 16, 32, 64, 128, 400):
    t = np.random.default_rng(0).random((B,32,32), dtype=np.float32)
    for _ in range(10): m(t)
    xs = []
    for _ in range(200):
        t0 = time.perf_counter(); m(t); xs.append((time.perf_counter() - t0) * 1e6)
    xs = np.array(xs)
    print(f"B={B:4d} in {t.nbytes/1024:7.1f} KB out {B*576*4/1024:7.1f} KB: median {np.median(xs):7.1f} us  p10 {np.percentile(xs,10):7.1f}  p90 {np.percentile(xs,90):7.1f}")
