// TEST INFRASTRUCTURE -- a stand-in for librccl.so that lets TWO rank processes share ONE card.
//
// Real RCCL refuses two ranks on one device ("invalid usage"), so on the 1-GPU test box the receive side of
// msiren_broadcast_weights (device blob -> host -> unpack -> commit on a non-root rank) could never execute.  This
// library implements the nine entry points libmsiren dlopens (mri_inr_amd/csrc/msiren.hip: struct Rccl) with the
// collectives carried by files in $RCCL_STUB_DIR: same signatures, same device-pointer semantics (the payload is read
// from / written to HIP device memory on the caller's stream), no xGMI.  Selected with MSIREN_RCCL_LIB; never shipped,
// never used by bench.py's default path.  Build: tests/test_gpu_multi.py:_build_rccl_stub().
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>

struct ncclComm {
    std::string dir, token;
    int nranks = 1, rank = 0;
    unsigned seq = 0;
};

namespace {

const double TIMEOUT_S = 120.0;

std::string stub_dir() {
    const char* d = std::getenv("RCCL_STUB_DIR");
    return d && *d ? d : "/tmp";
}

bool write_file(const std::string& path, const void* data, size_t bytes) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes;
    std::fclose(f);
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;  // atomic publish
}

bool read_file_when_there(const std::string& path, void* data, size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size == bytes) {
            FILE* f = std::fopen(path.c_str(), "rb");
            if (f) {
                const bool ok = bytes == 0 || std::fread(data, 1, bytes, f) == bytes;
                std::fclose(f);
                if (ok) return true;
            }
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
}

size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclFloat64: case ncclInt64: case ncclUint64: return 8;
        case ncclFloat32: case ncclInt32: case ncclUint32: return 4;
        case ncclFloat16: case ncclBfloat16: return 2;
        default: return 1;
    }
}

template <typename T>
void reduce(T* acc, const T* x, size_t n, ncclRedOp_t op) {
    for (size_t i = 0; i < n; ++i) {
        if (op == ncclMax) acc[i] = x[i] > acc[i] ? x[i] : acc[i];
        else if (op == ncclMin) acc[i] = x[i] < acc[i] ? x[i] : acc[i];
        else acc[i] = acc[i] + x[i];
    }
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "stub-%ld-%ld", (long)getpid(), (long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    auto* c = new ncclComm();
    c->dir = stub_dir();
    c->token = std::string(id.internal, strnlen(id.internal, sizeof id.internal));
    c->nranks = nranks;
    c->rank = rank;
    // collective: every rank announces itself and waits for the others (like the real bootstrap)
    char one = 1;
    if (!write_file(c->dir + "/" + c->token + ".init.r" + std::to_string(rank), &one, 1)) return ncclSystemError;
    for (int r = 0; r < nranks; ++r)
        if (!read_file_when_there(c->dir + "/" + c->token + ".init.r" + std::to_string(r), &one, 1)) return ncclSystemError;
    *comm = c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    (void)devlist;
    if (!comms || ndev != 1) return ncclInvalidUsage;  // the stub is for one process per rank
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    return ncclCommInitRank(&comms[0], 1, id, 0);
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root,
                                                                  ncclComm_t comm, hipStream_t stream) {
    if (!comm || root < 0 || root >= comm->nranks) return ncclInvalidArgument;
    const size_t bytes = count * dtype_bytes(datatype);
    const std::string path = comm->dir + "/" + comm->token + ".bcast" + std::to_string(comm->seq++);
    std::vector<unsigned char> host(bytes);
    if (comm->rank == root) {
        if (hipMemcpyAsync(host.data(), sendbuff, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
        if (comm->nranks > 1 && !write_file(path, host.data(), bytes)) return ncclSystemError;
        if (recvbuff != sendbuff && hipMemcpyAsync(recvbuff, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    } else {
        if (!read_file_when_there(path, host.data(), bytes)) return ncclSystemError;
        if (hipMemcpyAsync(recvbuff, host.data(), bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;  // `host` dies with this frame
    }
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                                                                  ncclComm_t comm, hipStream_t stream) {
    if (!comm) return ncclInvalidArgument;
    if (datatype != ncclFloat64 && datatype != ncclFloat32) return ncclInvalidArgument;
    const size_t bytes = count * dtype_bytes(datatype);
    const std::string base = comm->dir + "/" + comm->token + ".allred" + std::to_string(comm->seq++) + ".r";
    std::vector<unsigned char> mine(bytes), other(bytes);
    if (hipMemcpyAsync(mine.data(), sendbuff, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (comm->nranks > 1 && !write_file(base + std::to_string(comm->rank), mine.data(), bytes)) return ncclSystemError;
    for (int r = 0; r < comm->nranks; ++r) {
        if (r == comm->rank) continue;
        if (!read_file_when_there(base + std::to_string(r), other.data(), bytes)) return ncclSystemError;
        if (datatype == ncclFloat64) reduce((double*)mine.data(), (const double*)other.data(), count, op);
        else reduce((float*)mine.data(), (const float*)other.data(), count, op);
    }
    if (hipMemcpyAsync(recvbuff, mine.data(), bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() { return ncclSuccess; }
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) {
    return r == ncclSuccess ? "no error" : r == ncclSystemError ? "rccl stub: file rendezvous failed or timed out" : "rccl stub: error";
}

}  // extern "C"
