#!/usr/bin/env python3
"""What the HIP runtime says about base / size of page-locked host memory: hipHostMalloc blocks and hipHostRegister'ed ranges, queried at
the base and at interior pointers, through hipMemGetAddressRange, hipMemPtrGetInfo and hipPointerGetAttribute(RANGE_START_ADDR / RANGE_SIZE).
argv[1] == 'torch': import torch first (torch's bundled runtime)."""
import ctypes as C
import sys

import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
v = C.c_int()
hip.hipRuntimeGetVersion(C.byref(v))
print("runtime", v.value)
hip.hipSetDevice(0)
RANGE_START, RANGE_SIZE = 11, 12


def q(name, p):
    base, size = C.c_void_p(), C.c_size_t()
    r1 = hip.hipMemGetAddressRange(C.byref(base), C.byref(size), C.c_void_p(p))
    s2 = C.c_size_t()
    r2 = hip.hipMemPtrGetInfo(C.c_void_p(p), C.byref(s2))
    b3, s3 = C.c_void_p(), C.c_size_t()
    r3 = hip.hipPointerGetAttribute(C.byref(b3), RANGE_START, C.c_void_p(p))
    r4 = hip.hipPointerGetAttribute(C.byref(s3), RANGE_SIZE, C.c_void_p(p))
    print(f"  {name:34s} p={p:#x}: GetAddressRange rc={r1} base={base.value or 0:#x} size={size.value} | MemPtrGetInfo rc={r2} size={s2.value} | "
          f"attr RANGE_START rc={r3} {b3.value or 0:#x} RANGE_SIZE rc={r4} {s3.value}")


n = 1 << 20
hp = C.c_void_p()
assert hip.hipHostMalloc(C.byref(hp), C.c_size_t(n), C.c_uint(0)) == 0
print("hipHostMalloc block", hex(hp.value), n)
q("base", hp.value)
q("interior +4096", hp.value + 4096)
q("interior +12345", hp.value + 12345)
a = np.zeros(n // 4 + 64, np.float32)
reg = a.ctypes.data + 16 * 4
assert hip.hipHostRegister(C.c_void_p(reg), C.c_size_t(n), C.c_uint(0)) == 0
dp = C.c_void_p()
hip.hipHostGetDevicePointer(C.byref(dp), C.c_void_p(reg), C.c_uint(0))
print("hipHostRegister range", hex(reg), n, "device pointer", hex(dp.value))
q("host base", reg)
q("host interior +8192", reg + 8192)
q("device base", dp.value)
q("device interior +8192", dp.value + 8192)
q("host, just before", reg - 4)
q("host, just behind", reg + n)
hip.hipHostUnregister(C.c_void_p(reg))
q("pageable", a.ctypes.data)
