// What a caller's host buffer is to a synchronous entry point, and the RAII pieces a call wraps around it (msiren.hip uses them).
#pragma once
#include <cstring>

#include "host_ctx.h"

namespace mh {

// ---- the caller's host buffers -------------------------------------------------------------------------------------------
// A host range handed to a synchronous entry point is one of three things, decided per call from what the HIP runtime says about it
// (nothing is cached, nothing of the caller's is ever registered or unregistered by this library -- round 5's per-call hipHostRegister
// of pageable buffers is gone: profiles/r6/01_*):
//   HOST_PINNED    the WHOLE range lies inside ONE page-locked allocation (msiren_host_alloc, hipHostMalloc, a caller's hipHostRegister,
//                  a pinned torch tensor): kernels and DMA copies work on it in place through `dev`;
//   HOST_PAGEABLE  no byte of it is page-locked as far as its two ends tell: copied by the runtime (hipMemcpyAsync on the pointer);
//   HOST_PARTIAL   it begins or ends inside a page-locked allocation that does not contain all of it (a caller's own partial
//                  hipHostRegister; two registrations with pageable bytes between them): the runtime refuses a copy whose range leaves
//                  the registration it starts in ("invalid argument": tools/soak.py found it in round 5) and a kernel would fault on the
//                  pageable part, so the call goes through a page-locked bounce buffer of its own -- rare, slow, correct.
enum HostKind { HOST_PAGEABLE = 0, HOST_PINNED = 1, HOST_PARTIAL = 2 };

void* host_pinned_dev(const void* p);  // device address of page-locked host memory; nullptr for ordinary pageable memory
HostKind host_range_kind(const void* host, size_t bytes, void** dev);

// Page-locked memory of one call's own (the bounce buffer of a HOST_PARTIAL range)
struct HostBounce {
    void* p = nullptr;
    HostBounce() = default;
    HostBounce(const HostBounce&) = delete;
    HostBounce& operator=(const HostBounce&) = delete;
    ~HostBounce() { if (p) (void)hipHostFree(p); }
    void* alloc(size_t n) {
        if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
        }
        return p;
    }
};

// A caller's input / output buffer of one synchronous call: `as<T>()` is what the call's copies use (the caller's pointer, or the
// bounce buffer of a HOST_PARTIAL range), `dev<T>()` the device view of a HOST_PINNED range (nullptr otherwise: no in-place access).
class HostSrc {
    HostBounce b_;
    const void* p_;
    void* dev_ = nullptr;
    bool ok_ = true;

public:
    HostSrc(const void* host, size_t n) : p_(host) {
        if (!host || !n) return;
        if (host_range_kind(host, n, &dev_) != HOST_PARTIAL) return;
        if (b_.alloc(n)) { std::memcpy(b_.p, host, n); p_ = b_.p; dev_ = host_pinned_dev(b_.p); } else ok_ = false;
    }
    bool ok() const { return ok_; }
    template <typename T> const T* as() const { return (const T*)p_; }
    template <typename T> const T* dev() const { return (const T*)dev_; }
};
class HostDst {
    HostBounce b_;
    void* user_;
    void* p_;
    void* dev_ = nullptr;
    size_t n_;
    bool ok_ = true;

public:
    HostDst(void* host, size_t n) : user_(host), p_(host), n_(n) {
        if (!host || !n) return;
        if (host_range_kind(host, n, &dev_) != HOST_PARTIAL) return;
        if (b_.alloc(n)) { p_ = b_.p; dev_ = host_pinned_dev(b_.p); } else ok_ = false;
    }
    bool ok() const { return ok_; }
    template <typename T> T* as() const { return (T*)p_; }
    template <typename T> T* dev() const { return (T*)dev_; }
    void finish() const { if (b_.p) std::memcpy(user_, b_.p, n_); }  // (behind the stream's synchronisation)
};
#define HOSTBUF_OK(x) do { if (!(x).ok()) return ::mh::fail(MSIREN_E_HIP, "no page-locked memory for a bounce buffer"); } while (0)

// A synchronous call that leaves early (a failed launch, a failed copy) may have copies in flight on the caller's buffers or on a bounce
// buffer that is about to be freed: declared BEHIND the HostSrc / HostDst objects, so it runs before they go, it waits for the handle's
// streams unless the call has done so itself (disarm()).
struct DrainOnExit {
    msiren_ctx* h;
    bool armed = true;
    explicit DrainOnExit(msiren_ctx* hh) : h(hh) {}
    void disarm() { armed = false; }
    ~DrainOnExit();
};


}  // namespace mh
