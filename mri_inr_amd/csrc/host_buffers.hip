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
    // that holds the first byte must hold the last one -- base and size of it from the runtime.
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)d) != hipSuccess) {
        (void)hipGetLastError();
        return HOST_PARTIAL;  // (the runtime cannot name the allocation: do not trust the range)
    }
    if ((uintptr_t)d + bytes > (uintptr_t)base + size) return HOST_PARTIAL;
    *dev = d;
    return HOST_PINNED;
}

DrainOnExit::~DrainOnExit() {
    if (!armed || !h) return;
    for (auto& c : h->sc)
        if (c.s) (void)hipStreamSynchronize(c.s);
}

}  // namespace mh
