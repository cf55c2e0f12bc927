// libmsiren.so, host side: multi-GPU plumbing (include/msiren.h, "multi-GPU").  librccl is dlopen'ed by the first msiren_comm_* call:
// a single-GPU host never loads it.  One collective on the data path of the scale-out: the broadcast of the weight blob at load time.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "host_ctx.h"
#include "weights_blob.h"

namespace mh {
namespace {

// ---- RCCL (dlopen'ed on first use; types from <rccl/rccl.h>) -----------------------------------------
struct Rccl {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {std::getenv("MSIREN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.dl) break;
            r.err = dlerror();
        }
        if (!r.dl) return;
        bool ok = true;
        auto sym = [&](const char* n) {
            void* p = dlsym(r.dl, n);
            if (!p) {
                ok = false;
                r.err = std::string("missing symbol ") + n;
            }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            dlclose(r.dl);
            r.dl = nullptr;
        }
    });
    return r.dl ? &r : nullptr;
}

int need_rccl(Rccl** out) {
    Rccl* r = rccl();
    if (!r) {
        return fail(MSIREN_E_STATE, "librccl could not be loaded (multi-GPU entry points need it; set MSIREN_RCCL_LIB to its path)");
    }
    *out = r;
    return 0;
}

#define NCCLCHK(r_, expr)                                                                         \
    do {                                                                                          \
        ncclResult_t e_ = (expr);                                                                 \
        if (e_ != ncclSuccess)                                                                    \
            return fail(MSIREN_E_HIP, "%s failed: %s (%s:%d)", #expr, (r_)->GetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Flat image of the state_dict (weights_blob.h): header + one presence flag per expected key + every expected tensor.
// msiren_weights_export / _import hand it to the caller; msiren_broadcast_weights sends it through one ncclBroadcast and
// every receiving rank goes through import_blob() -- the same code a single-card test can drive.
size_t bcast_elems(msiren_ctx* h) { return msiren::blob_elems(h->expected); }

void bcast_pack(msiren_ctx* h, std::vector<float>& flat) {
    flat.resize(bcast_elems(h));
    msiren::blob_pack(h->expected, h->tensors, flat.data());
}

// blob -> tensors of the handle (replacing what it held) -> commit
int import_blob(msiren_ctx* h, const float* flat, size_t n) {
    std::string err;
    const int rc = msiren::blob_unpack(h->expected, flat, n, h->tensors, &err);
    if (rc) return fail(rc == -3 || rc == -1 ? MSIREN_E_SHAPE : MSIREN_E_INVALID, "%s", err.c_str());
    h->committed = false;
    return msiren_commit_weights(h);
}

}  // namespace
}  // namespace mh

using namespace mh;

extern "C" {

// ---- multi-GPU (include/msiren.h, "multi-GPU") ---------------------------------------------------------
int msiren_comm_unique_id(void* id_out, size_t bytes) {
    Rccl* r;
    int rc = need_rccl(&r);
    if (rc) return rc;
    if (!id_out || bytes < sizeof(ncclUniqueId)) return fail(MSIREN_E_INVALID, "id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    ncclUniqueId id;
    NCCLCHK(r, r->GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return 0;
}

int msiren_comm_init_rank(msiren_handle h, const void* id, size_t bytes, int32_t nranks, int32_t rank) {
    int rc = check(h, false);
    if (rc) return rc;
    Rccl* r;
    if ((rc = need_rccl(&r))) return rc;
    if (!id || bytes < sizeof(ncclUniqueId)) return fail(MSIREN_E_INVALID, "id must be the %zu bytes of msiren_comm_unique_id", sizeof(ncclUniqueId));
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(MSIREN_E_INVALID, "bad rank %d of %d", rank, nranks);
    if (h->comm) return fail(MSIREN_E_STATE, "the handle already belongs to a communicator (msiren_comm_destroy first)");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    NCCLCHK(r, r->CommInitRank(&h->comm, nranks, uid, rank));
    h->comm_n = nranks;
    h->comm_rank = rank;
    return 0;
}

int msiren_comm_init_all(msiren_handle* hs, int32_t n) {
    if (!hs || n < 1) return fail(MSIREN_E_INVALID, "bad arguments");
    Rccl* r;
    int rc = need_rccl(&r);
    if (rc) return rc;
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        if (!hs[i]) return fail(MSIREN_E_INVALID, "null handle %d", i);
        if (hs[i]->comm) return fail(MSIREN_E_STATE, "handle %d already belongs to a communicator", i);
        devs[i] = hs[i]->cfg.device;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return fail(MSIREN_E_INVALID, "handles %d and %d share device %d: one rank per GPU", j, i, devs[i]);
    }
    std::vector<ncclComm_t> comms(n);
    NCCLCHK(r, r->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        hs[i]->comm = comms[i];
        hs[i]->comm_n = n;
        hs[i]->comm_rank = i;
    }
    return 0;
}

static int broadcast_weights_group(msiren_handle* hs, int n, int32_t root) {
    Rccl* r;
    int rc = need_rccl(&r);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        if ((rc = check(hs[i], false))) return rc;
        if (!hs[i]->comm) return fail(MSIREN_E_STATE, "no communicator: call msiren_comm_init_rank / msiren_comm_init_all first");
        if (root < 0 || root >= hs[i]->comm_n) return fail(MSIREN_E_INVALID, "root %d out of range (%d ranks)", root, hs[i]->comm_n);
    }
    // every rank derives the layout from its own configuration: it must be the same model everywhere
    const size_t elems = bcast_elems(hs[0]);
    std::vector<float> flat;
    for (int i = 0; i < n; ++i) {
        msiren_ctx* h = hs[i];
        if (bcast_elems(h) != elems) return fail(MSIREN_E_SHAPE, "handles of one communicator describe different models");
        HIPCHK(hipSetDevice(h->cfg.device));
        if ((rc = sync_all(h)) || (rc = ensure(h, h->ws_comm, elems * sizeof(float)))) return rc;
        if (h->comm_rank == root) {
            bcast_pack(h, flat);
            HIPCHK(hipMemcpyAsync(h->ws_comm.p, flat.data(), elems * sizeof(float), hipMemcpyHostToDevice, h->sc[0].s));
            HIPCHK(hipStreamSynchronize(h->sc[0].s));  // `flat` is reused below
        }
    }
    if (n > 1) NCCLCHK(r, r->GroupStart());
    for (int i = 0; i < n; ++i) {
        msiren_ctx* h = hs[i];
        HIPCHK(hipSetDevice(h->cfg.device));
        NCCLCHK(r, r->Broadcast(h->ws_comm.p, h->ws_comm.p, elems, ncclFloat32, root, h->comm, h->sc[0].s));
    }
    if (n > 1) NCCLCHK(r, r->GroupEnd());
    for (int i = 0; i < n; ++i) {
        msiren_ctx* h = hs[i];
        HIPCHK(hipSetDevice(h->cfg.device));
        if (h->comm_rank != root) {
            flat.resize(elems);
            HIPCHK(hipMemcpyAsync(flat.data(), h->ws_comm.p, elems * sizeof(float), hipMemcpyDeviceToHost, h->sc[0].s));
            HIPCHK(hipStreamSynchronize(h->sc[0].s));
            if ((rc = import_blob(h, flat.data(), flat.size()))) return rc;  // unpack + commit
        } else {
            HIPCHK(hipStreamSynchronize(h->sc[0].s));
            if ((rc = msiren_commit_weights(h))) return rc;
        }
    }
    return 0;
}

int msiren_broadcast_weights(msiren_handle h, int32_t root) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    return broadcast_weights_group(&h, 1, root);
}

int msiren_broadcast_weights_all(msiren_handle* hs, int32_t n, int32_t root) {
    if (!hs || n < 1) return fail(MSIREN_E_INVALID, "bad arguments");
    for (int i = 0; i < n; ++i)
        if (!hs[i]) return fail(MSIREN_E_INVALID, "null handle %d", i);
    return broadcast_weights_group(hs, n, root);
}

int msiren_weights_blob_size(msiren_handle h, size_t* n_floats) {
    if (!h || !n_floats) return fail(MSIREN_E_INVALID, "null argument");
    *n_floats = bcast_elems(h);
    return 0;
}

int msiren_weights_export(msiren_handle h, float* blob_host, size_t n_floats) {
    if (!h || !blob_host) return fail(MSIREN_E_INVALID, "null argument");
    if (n_floats != bcast_elems(h))
        return fail(MSIREN_E_SHAPE, "blob buffer holds %zu floats, this configuration's blob has %zu (msiren_weights_blob_size)", n_floats, bcast_elems(h));
    msiren::blob_pack(h->expected, h->tensors, blob_host);
    return 0;
}

int msiren_weights_import(msiren_handle h, const float* blob_host, size_t n_floats) {
    int rc = check(h, false);
    if (rc) return rc;
    if (!blob_host) return fail(MSIREN_E_INVALID, "null argument");
    if ((rc = sync_all(h))) return rc;
    return import_blob(h, blob_host, n_floats);
}

int msiren_comm_allreduce_max_f64(msiren_handle h, double* inout, int32_t n) {
    int rc = check(h, false);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !inout)) return fail(MSIREN_E_INVALID, "bad arguments");
    if ((rc = sync_all(h))) return rc;
    if (!h->comm || n == 0) return 0;  // a communicator of one: the maximum is the input
    Rccl* r;
    if ((rc = need_rccl(&r))) return rc;
    if ((rc = ensure(h, h->ws_comm, (size_t)n * sizeof(double)))) return rc;
    HIPCHK(hipMemcpyAsync(h->ws_comm.p, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->sc[0].s));
    NCCLCHK(r, r->AllReduce(h->ws_comm.p, h->ws_comm.p, (size_t)n, ncclFloat64, ncclMax, h->comm, h->sc[0].s));
    HIPCHK(hipMemcpyAsync(inout, h->ws_comm.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->sc[0].s));
    HIPCHK(hipStreamSynchronize(h->sc[0].s));
    return 0;
}

int msiren_comm_barrier(msiren_handle h) {
    int rc = check(h, false);
    if (rc) return rc;
    if (!h->comm) return sync_all(h);  // a communicator of one
    double token = 0.0;
    return msiren_comm_allreduce_max_f64(h, &token, 1);
}

int msiren_comm_info(msiren_handle h, int32_t* nranks, int32_t* rank) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    if (nranks) *nranks = h->comm ? h->comm_n : 1;
    if (rank) *rank = h->comm ? h->comm_rank : 0;
    return 0;
}

int msiren_comm_destroy(msiren_handle h) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    if (!h->comm) return 0;
    Rccl* r;
    int rc = need_rccl(&r);
    if (rc) return rc;
    (void)hipSetDevice(h->cfg.device);
    (void)sync_all(h);
    NCCLCHK(r, r->CommDestroy(h->comm));
    h->comm = nullptr;
    h->comm_n = 1;
    h->comm_rank = 0;
    return 0;
}

}  // extern "C"

namespace mh {
int comm_destroy(msiren_ctx* h) { return msiren_comm_destroy(h); }
}  // namespace mh
