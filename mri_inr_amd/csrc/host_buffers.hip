// libmsiren.so, host side: classification of a caller's host range (host_buffers.h).
#include "host_buffers.h"

namespace mh {

// Device address of page-locked host memory; nullptr for ordinary pageable memory.
void* host_pinned_dev(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // (pageable memory is "invalid value" to the runtime: not an error of ours)
        return nullptr;
    }
    return a.type == hipMemoryTypeHost ? a.devicePointer : nullptr;
}

HostKind host_range_kind(const void* host, size_t bytes, void** dev) {
    *dev = nullptr;
    if (!host || !bytes) return HOST_PAGEABLE;
    void* const d = host_pinned_dev(host);
    if (!d) return (bytes > 1 && host_pinned_dev((const char*)host + bytes - 1)) ? HOST_PARTIAL : HOST_PAGEABLE;
    // both ends inside page-locked memory is not enough (two allocations, pageable bytes in between; on this platform the device
    // address of page-locked memory usually EQUALS its host address, so "d_last == d + bytes - 1" proves nothing): the allocation
    // that holds the first byte must hold the last one -- its start and size from the runtime.  hipPointerGetAttribute's RANGE_START_ADDR
    // / RANGE_SIZE answer for hipHostMalloc blocks and hipHostRegister'ed ranges alike, at interior pointers too, on both runtimes a
    // process can end up on; hipMemGetAddressRange reports base 0 for registered memory (tools/ptr_range_probe.py, profiles/r6/05_*).
    void* start = nullptr;
    size_t size = 0;
    if (hipPointerGetAttribute(&start, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)host) != hipSuccess ||
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)host) != hipSuccess || !start) {
        (void)hipGetLastError();
        return HOST_PARTIAL;  // (the runtime cannot name the allocation: do not trust the range)
    }
    if ((uintptr_t)host < (uintptr_t)start || (uintptr_t)host + bytes > (uintptr_t)start + size) return HOST_PARTIAL;
    *dev = d;
    return HOST_PINNED;
}

DrainOnExit::~DrainOnExit() {
    if (!armed || !h) return;
    for (auto& c : h->sc)
        if (c.s) (void)hipStreamSynchronize(c.s);
}

}  // namespace mh
