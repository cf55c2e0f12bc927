"""What page-locking a pageable numpy buffer per call would cost (hipHostRegister / hipHostUnregister of 0.9 - 2.5 MB): the price of letting the
kernels read / write a PAGEABLE caller buffer in place, as they do with page-locked ones."""
import ctypes as C, numpy as np, time
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipInit(0)
hip.hipSetDevice(0)
d = C.c_void_p(); hip.hipMalloc(C.byref(d), 4 << 20)
for mb in (0.9, 1.6, 2.5):
    n = int(mb * (1 << 20)) // 4
    ts, tu = [], []
    for rep in range(30):
        a = np.random.random(n).astype(np.float32)   # fresh pageable memory, touched
        t0 = time.perf_counter(); rc = hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), C.c_uint(0)); t1 = time.perf_counter()
        assert rc == 0, rc
        rc = hip.hipHostUnregister(C.c_void_p(a.ctypes.data)); t2 = time.perf_counter()
        ts.append((t1 - t0) * 1e6); tu.append((t2 - t1) * 1e6)
    print(f"{mb} MB: hipHostRegister median {np.median(ts):.0f} us, hipHostUnregister median {np.median(tu):.0f} us")
