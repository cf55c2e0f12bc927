import sys, time, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn
sd = syn.make_state_dict(seed=7, trained_like=True)
prec = sys.argv[1] if len(sys.argv) > 1 else "auto"   # auto (split-fp16) | fp32
m = ModulatedSiren(2,256,1,5,256,1.0,30.0,True,0.1,True,"custom",None,32,16,24,"cuda","sine", precision=prec)
m.load_state_dict(sd); m.to("cuda")
print("precision", prec)
for B in (1, 8, 40, 400):
    t = np.random.default_rng(0).random((B,32,32), dtype=np.float32)
    d_t = m.device_array(t.shape).copy_from(t); d_o = m.device_array((B,24,24))
    for _ in range(5): _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_t.ptr, B, d_o.ptr))
    m.sync()
    ms = C.c_float(); n = 50
    _lib.check(m._lib.msiren_timer_start(m._h))
    for _ in range(n): _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_t.ptr, B, d_o.ptr))
    _lib.check(m._lib.msiren_timer_stop(m._h, C.byref(ms)))
    for _ in range(5): m(t)   # (the first host-pointer call of a size sets up its staging: not part of the steady figure)
    t0 = time.perf_counter()
    for _ in range(n): m(t)
    host = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B}: device {ms.value/n*1e3:.1f} us per forward (back-to-back);  host numpy->numpy call {host*1e3:.1f} us")
