// libmsiren.so -- the C ABI declared in include/msiren.h: lifecycle, weights, the forward and slice entry points, memory, timing, info.
// Owns the device context (one device, up to three streams), the weight store keyed by the reference's state_dict names and the grow-only
// device workspaces; the launch sequence lives in launch_dispatch.hip, the weight layouts in weights_pack.hip (host_ctx.h: the map).
// Nothing here falls back to the CPU: every forward entry point launches HIP kernels or fails.
#include <dlfcn.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "host_ctx.h"
#include "host_buffers.h"
#include "host_plan.h"
#include "weights_blob.h"

namespace mh {

namespace {
thread_local std::string g_err;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
const char* last_error() { return g_err.c_str(); }

int use_device(msiren_ctx* h) {
    HIPCHK(hipSetDevice(h->cfg.device));
    return 0;
}

int ensure(msiren_ctx* h, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t cap = bytes + bytes / 4 + 256;
    HIPCHK(hipMalloc(&b.p, cap));
    b.cap = cap;
    (void)h;
    return 0;
}

int upload(float** dst, const std::vector<float>& src) {
    if (*dst) HIPCHK(hipFree(*dst));
    *dst = nullptr;
    HIPCHK(hipMalloc((void**)dst, src.size() * sizeof(float)));
    HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

bool take_range_flag(msiren_ctx* h) {
    if (!h->status_host || !*h->status_host) return false;
    *h->status_host = 0;
    h->range_events++;
    return true;
}

int sync_all(msiren_ctx* h) {
    for (auto& c : h->sc)
        if (c.s) HIPCHK(hipStreamSynchronize(c.s));
    (void)take_range_flag(h);  // informational: the outputs are the exact-fp32 trunk's already
    return 0;
}

template <typename F>
int with_range_fallback(msiren_ctx* h, F&& run) {
    const int rc = run();
    (void)take_range_flag(h);
    return rc;
}

// asynchronous forward entry points rotate over the configured streams
void next_stream(msiren_ctx* h) {
    if (h->nstreams > 1) h->cur = (h->cur + 1) % h->nstreams;
}

// event pairs recorded since the last collection -> totals (the streams have been synchronised by the caller)
int profile_collect(msiren_ctx* h) {
    for (size_t i = 0; i < h->prof_used; ++i) {
        const auto& r = h->prof_events[i];
        if (r.kernel < 0) continue;  // (the launch between the pair failed)
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        h->prof_ms += ms;
        h->prof_launches++;
        auto& k = h->prof_kernels[r.kernel];
        k.ms += ms;
        k.launches++;
        k.coords += r.coords;
    }
    h->prof_used = 0;
    return 0;
}

int check(msiren_ctx* h, bool need_commit) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    if (need_commit && !h->committed) return fail(MSIREN_E_STATE, "weights not committed: call msiren_set_tensor for every net.* key, then msiren_commit_weights");
    return use_device(h);
}


}  // namespace mh

using namespace mh;

// =================================================================================================
extern "C" {

int msiren_abi_version(void) { return MSIREN_ABI_VERSION; }

const char* msiren_last_error(void) { return mh::last_error(); }

int msiren_device_count(int32_t* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (count) *count = n;
    return 0;
}

int msiren_create(const msiren_config* cfg, msiren_handle* out) {
    if (!cfg || !out) return fail(MSIREN_E_INVALID, "null argument");
    if (cfg->abi_version != MSIREN_ABI_VERSION)
        return fail(MSIREN_E_INVALID, "ABI version mismatch: header %d, library %d", cfg->abi_version, MSIREN_ABI_VERSION);
    if (cfg->dim_in != 2) return fail(MSIREN_E_INVALID, "dim_in must be 2 (the coordinate grid is a 2-D meshgrid), got %d", cfg->dim_in);
    if (cfg->dim_out != 1) return fail(MSIREN_E_INVALID, "dim_out must be 1 (the reference's squeeze(2)+rearrange only works for 1), got %d", cfg->dim_out);
    if (cfg->dim_hidden < 1 || cfg->dim_hidden > 512) return fail(MSIREN_E_INVALID, "dim_hidden must be in [1,512], got %d", cfg->dim_hidden);
    if (cfg->num_layers < 1 || cfg->num_layers > 64) return fail(MSIREN_E_INVALID, "num_layers must be in [1,64], got %d", cfg->num_layers);
    if (cfg->latent_dim < 1) return fail(MSIREN_E_INVALID, "latent_dim must be positive, got %d", cfg->latent_dim);
    if (cfg->siren_patch_size < 2) return fail(MSIREN_E_INVALID, "siren_patch_size must be >= 2, got %d", cfg->siren_patch_size);
    if (cfg->inner_patch_size < 1 || cfg->outer_patch_size < cfg->inner_patch_size)
        return fail(MSIREN_E_INVALID, "need outer_patch_size >= inner_patch_size >= 1");
    if (cfg->activation != MSIREN_ACT_SINE && cfg->activation != MSIREN_ACT_MORLET) return fail(MSIREN_E_INVALID, "unknown activation %d", cfg->activation);
    if (cfg->precision < MSIREN_PREC_F32 || cfg->precision > MSIREN_PREC_F16)
        return fail(MSIREN_E_INVALID, "unknown precision %d", cfg->precision);
    if (cfg->w0 == 0.f || cfg->w0_initial == 0.f) return fail(MSIREN_E_INVALID, "w0 and w0_initial must be non-zero");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(MSIREN_E_INVALID, "device %d out of range (%d visible)", cfg->device, ndev);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MSIREN_E_INVALID, "device %d is %s; libmsiren is built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);
    auto* h = new msiren_ctx();
    h->cfg = *cfg;
    h->H = cfg->dim_hidden;
    h->HP = (cfg->dim_hidden + 127) / 128 * 128;
    h->L = cfg->num_layers;
    h->Z = cfg->latent_dim;
    h->S = cfg->siren_patch_size;
    h->P = h->S * h->S;
    h->O = cfg->outer_patch_size;
    h->I = cfg->inner_patch_size;
    h->num_cus = prop.multiProcessorCount;
    if (const char* e = std::getenv("MSIREN_F16_HALF")) h->half_allowed = std::atoi(e) != 0;
    if (const char* e = std::getenv("MSIREN_HOST_PIPE_MIN")) h->host_pipe_min = std::max(128, std::atoi(e));
    if (const char* e = std::getenv("MSIREN_QUEUE_START")) h->queue_start = (unsigned)std::strtoul(e, nullptr, 0);
    if (const char* e = std::getenv("MSIREN_F16_WS")) h->f16_ws = std::atoi(e) != 0;
    if (const char* e = std::getenv("MSIREN_TRACE_HOST")) h->trace_host = std::atoi(e);
    if (const char* e = std::getenv("MSIREN_PROLOGUE_F16X3")) h->em_enabled = std::atoi(e) != 0;
    if (const char* e = std::getenv("MSIREN_EM_DEPTH")) h->em_depth = std::atoi(e);
    declare_expected(h);
    hipError_t e = hipSetDevice(cfg->device);
    for (auto& c : h->sc)
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->status_host, 64, hipHostMallocMapped);
    if (e == hipSuccess) {
        for (int i = 0; i < 16; ++i) h->status_host[i] = 0;  // [0] f16x3 domain guard (informational), [8] the flag word of synchronous host calls
        e = hipHostGetDevicePointer((void**)&h->status_dev, (void*)h->status_host, 0);
    }

    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) {
        msiren_destroy(h);  // releases whatever was created
        return fail(MSIREN_E_HIP, "context creation failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return 0;
}

int msiren_destroy(msiren_handle h) {
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    for (auto& c : h->sc)
        if (c.s) (void)hipStreamSynchronize(c.s);
    if (h->comm) (void)comm_destroy(h);
    if (h->status_host) (void)hipHostFree((void*)h->status_host);
    if (h->ws_comm.p) (void)hipFree(h->ws_comm.p);
    if (h->d_wp16n) (void)hipFree(h->d_wp16n);
    if (h->d_emw) (void)hipFree(h->d_emw);
    if (h->d_emc2) (void)hipFree(h->d_emc2);
    for (void* q : {h->d_woutx1, h->d_wpx1n, h->d_wpx1w, (void*)h->d_bias32x1})
        if (q) (void)hipFree(q);
    float* ptrs[] = {h->d_dump, h->d_s0t512, h->d_s0t, h->d_bias16, h->d_wout16, h->d_grid, h->d_l0, h->d_wp, h->d_bias, h->d_wout, h->d_modw, h->d_modw_rm, h->d_modb, h->d_encw, h->d_foldw, h->d_embias};
    for (float* p : ptrs)
        if (p) (void)hipFree(p);
    std::vector<DevBuf*> bufs = {&h->ws_out, &h->ws_tiles, &h->ws_in, &h->ws_img};
    for (auto& c : h->sc)
        for (DevBuf* b : {&c.cscratch, &c.mods, &c.modpad, &c.latent, &c.patches, &c.keep, &c.rec, &c.queue, &c.feat, &c.plan}) bufs.push_back(b);
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& pr : h->prof_events) {
        (void)hipEventDestroy(pr.a);
        (void)hipEventDestroy(pr.b);
    }
    for (auto& c : h->sc)
        if (c.ev_join) (void)hipEventDestroy(c.ev_join);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (auto& c : h->sc)
        if (c.s) (void)hipStreamDestroy(c.s);
    delete h;
    return 0;
}

int msiren_set_tensor(msiren_handle h, const char* name, const float* host_data, size_t n) {
    if (!h || !name || (!host_data && n)) return fail(MSIREN_E_INVALID, "null argument");
    auto it = h->expected.find(name);
    if (it == h->expected.end()) return fail(MSIREN_E_INVALID, "Unexpected key in state_dict: \"%s\"", name);
    if (it->second != n)
        return fail(MSIREN_E_SHAPE, "size mismatch for %s: got %zu elements, the configuration implies %zu", name, n, it->second);
    h->tensors[name].assign(host_data, host_data + n);
    h->committed = false;
    return 0;
}

int msiren_get_tensor(msiren_handle h, const char* name, float* host_out, size_t n) {
    if (!h || !name || (!host_out && n)) return fail(MSIREN_E_INVALID, "null argument");
    auto ex = h->expected.find(name);
    if (ex == h->expected.end()) return fail(MSIREN_E_INVALID, "Unexpected key in state_dict: \"%s\"", name);
    auto it = h->tensors.find(name);
    if (it == h->tensors.end()) return fail(MSIREN_E_STATE, "tensor %s has not been set", name);
    if (it->second.size() != n) return fail(MSIREN_E_SHAPE, "size mismatch for %s: asked for %zu elements, it has %zu", name, n, it->second.size());
    std::copy(it->second.begin(), it->second.end(), host_out);
    return 0;
}

int msiren_commit_weights(msiren_handle h) {
    int rc = check(h, false);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    if ((rc = pack_trunk(h))) return rc;
    if ((rc = pack_trunk_f16x3(h))) return rc;
    if ((rc = pack_trunk_x1(h))) return rc;
    if ((rc = pack_fold_weights(h))) return rc;
    rc = pack_modulator(h);
    if (rc < 0) return rc;
    h->have_modulator = (rc == 0);
    rc = pack_encoder(h);
    if (rc < 0) return rc;
    h->have_encoder = (rc == 0);
    if ((rc = pack_prologue_f16x3(h))) return rc;
    h->committed = true;
    return 0;
}

int msiren_forward_mods_dev(msiren_handle h, const float* mods_dev, int64_t B, float* out_dev) {
    int rc = check(h);
    if (rc) return rc;
    next_stream(h);
    if (B < 0 || (B > 0 && (!mods_dev || !out_dev))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    return launch_trunk(h, mods_dev, B, out_dev);
}

namespace {
struct SoloCall {  // marks a synchronous single-stream host call for its duration (trunk choice: use_f16x3w, ring depth)
    msiren_ctx* h;
    explicit SoloCall(msiren_ctx* hh, bool on = true) : h(hh) { if (h) h->solo = on; }
    ~SoloCall() { if (h) h->solo = false; }
};
}  // namespace

static int msiren_forward_mods_impl(msiren_handle h, const float* mods_host, int64_t B, float* out_host) {
    int rc = check(h);
    if (rc) return rc;
    SoloCall solo(h);
    if (B < 0 || (B > 0 && (!mods_host || !out_host))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    if (B == 0) return 0;
    const size_t nm = (size_t)h->L * B * h->H * sizeof(float), no = (size_t)B * h->P * sizeof(float);
    if ((rc = ensure(h, h->sc[h->cur].mods, nm)) || (rc = ensure(h, h->ws_out, no))) return rc;
    const HostSrc src(mods_host, nm);
    const HostDst dst(out_host, no);
    HOSTBUF_OK(src);
    HOSTBUF_OK(dst);
    DrainOnExit drain(h);
    HIPCHK(hipMemcpyAsync(h->sc[h->cur].mods.p, src.as<float>(), nm, hipMemcpyHostToDevice, h->sc[h->cur].s));
    if ((rc = launch_trunk(h, (const float*)h->sc[h->cur].mods.p, B, (float*)h->ws_out.p))) return rc;
    HIPCHK(hipMemcpyAsync(dst.as<float>(), h->ws_out.p, no, hipMemcpyDeviceToHost, h->sc[h->cur].s));
    HIPCHK(hipStreamSynchronize(h->sc[h->cur].s));
    drain.disarm();
    dst.finish();
    return 0;
}

int msiren_forward_mods(msiren_handle h, const float* mods_host, int64_t B, float* out_host) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    return with_range_fallback(h, [&] { return msiren_forward_mods_impl(h, mods_host, B, out_host); });
}

int msiren_forward_latent_dev(msiren_handle h, const float* z_dev, int64_t B, float* out_dev, float* mods_out_dev) {
    int rc = check(h);
    if (rc) return rc;
    next_stream(h);
    if (B < 0 || (B > 0 && (!z_dev || !out_dev))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    return forward_latent_dev(h, z_dev, B, out_dev, mods_out_dev);
}

static int msiren_forward_latent_impl(msiren_handle h, const float* z_host, int64_t B, float* out_host, float* mods_out_host) {
    int rc = check(h);
    if (rc) return rc;
    SoloCall solo(h);
    if (B < 0 || (B > 0 && (!z_host || !out_host))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    if (B == 0) return 0;
    const size_t nz = (size_t)B * h->Z * sizeof(float), no = (size_t)B * h->P * sizeof(float);
    const size_t nm = (size_t)h->L * B * h->H * sizeof(float);
    if ((rc = ensure(h, h->sc[h->cur].latent, nz)) || (rc = ensure(h, h->ws_out, no)) || (rc = ensure(h, h->sc[h->cur].mods, nm))) return rc;
    const HostSrc src(z_host, nz);
    const HostDst dst(out_host, no), dst_mods(mods_out_host, nm);
    HOSTBUF_OK(src);
    HOSTBUF_OK(dst);
    HOSTBUF_OK(dst_mods);
    DrainOnExit drain(h);
    HIPCHK(hipMemcpyAsync(h->sc[h->cur].latent.p, src.as<float>(), nz, hipMemcpyHostToDevice, h->sc[h->cur].s));
    if ((rc = forward_latent_dev(h, (const float*)h->sc[h->cur].latent.p, B, (float*)h->ws_out.p, (float*)h->sc[h->cur].mods.p))) return rc;
    HIPCHK(hipMemcpyAsync(dst.as<float>(), h->ws_out.p, no, hipMemcpyDeviceToHost, h->sc[h->cur].s));
    if (mods_out_host) HIPCHK(hipMemcpyAsync(dst_mods.as<float>(), h->sc[h->cur].mods.p, nm, hipMemcpyDeviceToHost, h->sc[h->cur].s));
    HIPCHK(hipStreamSynchronize(h->sc[h->cur].s));
    drain.disarm();
    dst.finish();
    dst_mods.finish();
    return 0;
}

int msiren_forward_latent(msiren_handle h, const float* z_host, int64_t B, float* out_host, float* mods_out_host) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    return with_range_fallback(h, [&] { return msiren_forward_latent_impl(h, z_host, B, out_host, mods_out_host); });
}

// ---- the two producers alone: model.encoder(tiles) and model.modulator(z) of the reference (modulated_siren.py:420, 416) ----
int msiren_encode_tiles_dev(msiren_handle h, const float* tiles_dev, int64_t B, float* z_dev) {
    int rc = check(h);
    if (rc) return rc;
    next_stream(h);
    if (B < 0 || (B > 0 && (!tiles_dev || !z_dev))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    if (h->O != 32) return fail(MSIREN_E_INVALID, "the custom encoder is hard-wired to 32x32 tiles (siren_encoder.py:499), outer_patch_size=%d", h->O);
    return launch_encoder(h, tiles_dev, B, z_dev);
}

int msiren_modulate_dev(msiren_handle h, const float* z_dev, int64_t B, float* mods_dev) {
    int rc = check(h);
    if (rc) return rc;
    next_stream(h);
    if (B < 0 || (B > 0 && (!z_dev || !mods_dev))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    return launch_modulator(h, z_dev, B, mods_dev);
}

int msiren_encode_tiles(msiren_handle h, const float* tiles_host, int64_t B, float* z_host) {
    int rc = check(h);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!tiles_host || !z_host))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    if (h->O != 32) return fail(MSIREN_E_INVALID, "the custom encoder is hard-wired to 32x32 tiles (siren_encoder.py:499), outer_patch_size=%d", h->O);
    if (B == 0) return 0;
    const size_t nt = (size_t)B * h->O * h->O * sizeof(float), nz = (size_t)B * h->Z * sizeof(float);
    auto& c = h->sc[h->cur];
    if ((rc = ensure(h, h->ws_tiles, nt)) || (rc = ensure(h, c.latent, nz))) return rc;
    const HostSrc src(tiles_host, nt);
    const HostDst dst(z_host, nz);
    HOSTBUF_OK(src);
    HOSTBUF_OK(dst);
    DrainOnExit drain(h);
    HIPCHK(hipMemcpyAsync(h->ws_tiles.p, src.as<float>(), nt, hipMemcpyHostToDevice, c.s));
    if ((rc = launch_encoder(h, (const float*)h->ws_tiles.p, B, (float*)c.latent.p))) return rc;
    HIPCHK(hipMemcpyAsync(dst.as<float>(), c.latent.p, nz, hipMemcpyDeviceToHost, c.s));
    HIPCHK(hipStreamSynchronize(c.s));
    drain.disarm();
    dst.finish();
    return 0;
}

int msiren_modulate(msiren_handle h, const float* z_host, int64_t B, float* mods_host) {
    int rc = check(h);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!z_host || !mods_host))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    if (B == 0) return 0;
    const size_t nz = (size_t)B * h->Z * sizeof(float), nm = (size_t)h->L * B * h->H * sizeof(float);
    auto& c = h->sc[h->cur];
    if ((rc = ensure(h, c.latent, nz)) || (rc = ensure(h, c.mods, nm))) return rc;
    const HostSrc src(z_host, nz);
    const HostDst dst(mods_host, nm);
    HOSTBUF_OK(src);
    HOSTBUF_OK(dst);
    DrainOnExit drain(h);
    HIPCHK(hipMemcpyAsync(c.latent.p, src.as<float>(), nz, hipMemcpyHostToDevice, c.s));
    if ((rc = launch_modulator(h, (const float*)c.latent.p, B, (float*)c.mods.p))) return rc;
    HIPCHK(hipMemcpyAsync(dst.as<float>(), c.mods.p, nm, hipMemcpyDeviceToHost, c.s));
    HIPCHK(hipStreamSynchronize(c.s));
    drain.disarm();
    dst.finish();
    return 0;
}

int msiren_forward_tiles_dev(msiren_handle h, const float* tiles_dev, int64_t B, float* out_dev) {
    int rc = check(h);
    if (rc) return rc;
    next_stream(h);
    if (B < 0 || (B > 0 && (!tiles_dev || !out_dev))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    if (h->O != 32) return fail(MSIREN_E_INVALID, "the custom encoder is hard-wired to 32x32 tiles (siren_encoder.py:499), outer_patch_size=%d", h->O);
    return forward_tiles_dev(h, tiles_dev, B, out_dev);
}

static int msiren_forward_tiles_impl(msiren_handle h, const float* tiles_host, int64_t B, float* out_host) {
    int rc = check(h);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!tiles_host || !out_host))) return fail(MSIREN_E_INVALID, "bad arguments (B=%lld)", (long long)B);
    if (h->O != 32) return fail(MSIREN_E_INVALID, "the custom encoder is hard-wired to 32x32 tiles (siren_encoder.py:499), outer_patch_size=%d", h->O);
    if (B == 0) return 0;
    const size_t nt = (size_t)B * h->O * h->O * sizeof(float), no = (size_t)B * h->P * sizeof(float);
    if ((rc = ensure(h, h->ws_tiles, nt)) || (rc = ensure(h, h->ws_out, no))) return rc;
    // From host_pipe_min tiles (2400 = six slices) up the call pipelines itself (round 5).  The device side of a slice is ~325 us; uploading
    // its 1.6 MB first and downloading its 0.9 MB afterwards added ~90 us in front and behind.  Patches are independent
    // (modulated_siren.py:435-457), so the batch is cut into chunks that alternate between the handle's two streams:
    //     H2D_0 | launch_0 | H2D_1 | launch_1 | D2H_0 | H2D_2 | launch_2 | D2H_1 | ... | D2H_last
    // (a pageable copy blocks the host until it is done -- so each is issued where the device has other work queued).
    // Chunk 0 is SMALL (112 tiles = two rounds of the register-resident trunk): its upload is short, so the device starts early,
    // and its trunk runs while the next chunk's tiles arrive and its encoder / Modulator run beside it: 8 slices per call 2.57 -> 2.27 ms.
    // Every chunk but the last takes the register-resident trunk (room beside it for the next chunk's prologue), the last one the
    // weight-stationary trunk (the faster kernel; nothing is left to run beside it but the previous chunk's download).  Below the
    // threshold ONE chunk whose kernels read / write page-locked caller buffers in place is faster: 800 tiles 658 against 818 us,
    // 1600 tiles 1220 against 1227, 3200 tiles 2353 against 2284 (profiles/r5/04_host_call_pipelining.txt).  All trunk and prologue
    // instances give the same bits, so the cut does not change results (tests/test_gpu_host_calls.py).
    using Chunk = msiren::HostChunk;
    std::vector<Chunk> plan;
    const int cur0 = h->cur;
    const bool pipelined = B >= h->host_pipe_min && use_f16x3(h) && !h->x1_ready && h->L == 5 && h->em_enc && h->em_mod && ws_capable(h, B);
    if (pipelined) plan = msiren::pipelined_host_plan(B, h->host_first, h->host_piece, cur0);  // (host_plan.h: unit-tested on the CPU)
    else plan.push_back({0, B, cur0, 0, false});
    const int nchunks = (int)plan.size();
    const size_t tile_elems = (size_t)h->O * h->O;
    // In place (round 5): where the caller's OUTPUT array is page-locked memory (msiren_host_alloc; the Python mirror's outputs come from a
    // recycling pool of such blocks by default; a pinned torch tensor) the trunk stores its 0.9 MB per slice straight into it over the course
    // of its 265 us -- no download, no wait for one behind the stream; page-locked TILES are read in place by the conv kernel.  Pageable
    // memory (a plain numpy array) is copied by the runtime.  One-chunk calls only: same box, 400 tiles: 390 us with both copies, 369 with
    // the output in place, 360 with the tiles in place as well; a cut call of 3 200 tiles: 2.24 ms with copies (they run beside the other
    // chunk's kernels anyway), 2.35-2.87 ms in place (profiles/r5/04_host_call_pipelining.txt).
    const HostSrc src(tiles_host, nt);  // (a range that is page-locked in part goes through a bounce buffer: host_range_kind)
    const HostDst dst(out_host, no);
    HOSTBUF_OK(src);
    HOSTBUF_OK(dst);
    DrainOnExit drain(h);  // (an early return waits for what is in flight on these buffers before they go)
    tiles_host = src.as<float>();
    out_host = dst.as<float>();
    float* out_zc_ = nchunks == 1 ? dst.dev<float>() : nullptr;
    const float* in_zc_ = nchunks == 1 ? src.dev<float>() : nullptr;
    float* const out_zc = out_zc_;
    const float* const in_zc = in_zc_;
    float* const out_base = out_zc ? out_zc : (float*)h->ws_out.p;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto us = [&]() { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); };
    std::vector<double> tr_h2d(nchunks, 0.0), tr_launch(nchunks, 0.0), tr_d2h(nchunks, 0.0);
    SoloCall solo(h);
    struct Restore {  // the launchers address the stream through h->cur, the trunk through h->trunk_force, the prologue's ring through h->em_beside
        msiren_ctx* h;
        int cur;
        ~Restore() { h->cur = cur; h->trunk_force = 0; h->em_beside = false; h->trunk_after = nullptr; h->host_check_now = false; h->hc.armed = false; }
    } restore{h, cur0};
    h->host_check_now = nchunks == 1;
    auto download = [&](int k) {
        const Chunk& c = plan[k];
        tr_d2h[k] = us();
        if (out_zc) return;
        hipError_t e = hipMemcpyAsync(out_host + (size_t)c.lo * h->P, out_base + (size_t)c.lo * h->P, (size_t)c.n * h->P * sizeof(float),
                                      hipMemcpyDeviceToHost, h->sc[c.stream].s);
        if (e != hipSuccess && !rc) rc = fail(MSIREN_E_HIP, "hipMemcpyAsync(D2H): %s", hipGetErrorString(e));
        tr_d2h[k] = us();
    };
    for (int k = 0; k < nchunks && !rc; ++k) {
        const Chunk& c = plan[k];
        h->cur = c.stream;
        h->trunk_force = c.trunk;
        h->em_beside = c.beside;
        const float* d_t = in_zc ? in_zc + (size_t)c.lo * tile_elems : (const float*)h->ws_tiles.p + (size_t)c.lo * tile_elems;
        if (!in_zc) {
            hipError_t e = hipMemcpyAsync((void*)d_t, tiles_host + (size_t)c.lo * tile_elems, (size_t)c.n * tile_elems * sizeof(float), hipMemcpyHostToDevice, h->sc[c.stream].s);
            if (e != hipSuccess) rc = fail(MSIREN_E_HIP, "hipMemcpyAsync(H2D): %s", hipGetErrorString(e));
        }
        tr_h2d[k] = us();
        // (the weight-stationary trunk owns its CUs: queued beside the previous chunk's conditional exact-fp32 launch it would start first,
        //  and that launch -- and the download behind it -- would wait for it to end)
        if (pipelined && c.trunk == 2 && k >= 1 && !rc) h->trunk_after = h->sc[plan[k - 1].stream].ev_join;
        if (!rc) rc = forward_tiles_dev(h, d_t, c.n, out_base + (size_t)c.lo * h->P);
        if (pipelined && !rc) {
            auto& sc = h->sc[c.stream];
            if (!sc.ev_join) { hipError_t e2 = hipEventCreateWithFlags(&sc.ev_join, hipEventDisableTiming); if (e2 != hipSuccess) rc = fail(MSIREN_E_HIP, "hipEventCreate: %s", hipGetErrorString(e2)); }
            if (!rc) { hipError_t e2 = hipEventRecord(sc.ev_join, sc.s); if (e2 != hipSuccess) rc = fail(MSIREN_E_HIP, "hipEventRecord: %s", hipGetErrorString(e2)); }
        }
        tr_launch[k] = us();
        // (pipelined: the previous chunk's download is issued once this chunk's work is queued behind it on the other stream;
        //  otherwise all downloads follow all launches, as a pageable D2H blocks the host until its chunk is done)
        if (pipelined && k >= 1 && !rc) download(k - 1);
    }
    for (int k = pipelined ? nchunks - 1 : 0; k < nchunks && !rc; ++k) download(k);
    h->cur = cur0;
    int rs = sync_all(h);
    if (!rc && !rs && h->hc.armed && (unsigned)h->status_host[8] == h->hc.epoch) {
        // the trunk met a modulation outside the fp16 domain: the batch once more on the exact-fp32 trunk (the conditional kernel, its
        // condition pointed at the word that has just been read), the download once more if there is one
        h->hc.armed = false;
        h->cur = plan[0].stream;
        rc = launch_trunk_f32_cond(h, h->hc.mods, h->hc.B, h->hc.out, h->status_dev + 8, h->hc.epoch);
        if (!rc) download(0);
        h->cur = cur0;
        rs = sync_all(h);
    }
    if (!rs) drain.disarm();
    if (!rc && !rs) dst.finish();
    if (h->trace_host) {
        std::fprintf(stderr, "msiren_forward_tiles B=%lld chunks=%d%s (us since entry): ", (long long)B, nchunks, pipelined ? " pipelined" : "");
        for (int k = 0; k < nchunks; ++k) std::fprintf(stderr, "[%lld tiles: h2d %.0f launched %.0f d2h %.0f] ", (long long)plan[k].n, tr_h2d[k], tr_launch[k], tr_d2h[k]);
        std::fprintf(stderr, "synced %.0f\n", us());
    }
    return rc ? rc : rs;
}

int msiren_forward_tiles(msiren_handle h, const float* tiles_host, int64_t B, float* out_host) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    return with_range_fallback(h, [&] { return msiren_forward_tiles_impl(h, tiles_host, B, out_host); });
}

static int msiren_reconstruct_slices_impl(msiren_handle h, const float* images_host, int64_t n, int32_t height, int32_t width, float* recon_host) {
    int rc = check(h);
    if (rc) return rc;
    SoloCall solo(h);
    if (n < 0 || (n > 0 && (!images_host || !recon_host))) return fail(MSIREN_E_INVALID, "bad arguments");
    if (n == 0) return 0;
    int32_t nV, nH;
    if ((rc = msiren_recon_shape(h, height, width, &nV, &nH))) return rc;
    const size_t ni = (size_t)n * height * width * sizeof(float);
    const size_t nr = (size_t)n * nV * h->I * nH * h->I * sizeof(float);
    if ((rc = ensure(h, h->ws_in, ni)) || (rc = ensure(h, h->ws_img, nr))) return rc;
    // As in msiren_forward_tiles: where the caller's reconstruction array is page-locked memory (the Python mirror's outputs are, by default)
    // the fold stores straight into it; the image always arrives by a copy (DMA from page-locked memory, through the runtime from pageable
    // memory): read in place every pixel would cross the link four times (32 x 32 tiles at a stride of 16; profiles/r5/09_*).
    auto& sc = h->sc[h->cur];
    const HostSrc src(images_host, ni);
    const HostDst dst(recon_host, nr);
    HOSTBUF_OK(src);
    HOSTBUF_OK(dst);
    DrainOnExit drain(h);
    float* const d_rec = dst.dev<float>() ? dst.dev<float>() : (float*)h->ws_img.p;
    HIPCHK(hipMemcpyAsync(h->ws_in.p, src.as<float>(), ni, hipMemcpyHostToDevice, sc.s));
    if ((rc = reconstruct_on_current_stream(h, (const float*)h->ws_in.p, n, height, width, d_rec))) return rc;
    if (d_rec == (float*)h->ws_img.p) HIPCHK(hipMemcpyAsync(dst.as<float>(), h->ws_img.p, nr, hipMemcpyDeviceToHost, sc.s));
    HIPCHK(hipStreamSynchronize(sc.s));
    drain.disarm();
    dst.finish();
    return 0;
}

int msiren_reconstruct_slices(msiren_handle h, const float* images_host, int64_t n, int32_t height, int32_t width, float* recon_host) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    return with_range_fallback(h, [&] { return msiren_reconstruct_slices_impl(h, images_host, n, height, width, recon_host); });
}

int msiren_set_streams(msiren_handle h, int32_t n) {
    int rc = check(h, false);
    if (rc) return rc;
    if (n < 1 || n > 3) return fail(MSIREN_E_INVALID, "streams must be 1, 2 or 3, got %d", n);
    if ((rc = sync_all(h))) return rc;
    h->nstreams = n;
    h->cur = 0;
    return 0;
}

int msiren_sync(msiren_handle h) {
    int rc = check(h, false);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    return 0;
}

int msiren_dev_alloc(msiren_handle h, size_t bytes, void** dev_ptr) {
    int rc = check(h, false);
    if (rc) return rc;
    if (!dev_ptr) return fail(MSIREN_E_INVALID, "null argument");
    *dev_ptr = nullptr;
    if (bytes == 0) return 0;
    HIPCHK(hipMalloc(dev_ptr, bytes));
    return 0;
}

int msiren_dev_free(msiren_handle h, void* dev_ptr) {
    int rc = check(h, false);
    if (rc) return rc;
    if (dev_ptr) {
        if ((rc = sync_all(h))) return rc;
        HIPCHK(hipFree(dev_ptr));
    }
    return 0;
}

int msiren_host_alloc(msiren_handle h, size_t bytes, void** host_ptr) {
    int rc = check(h, false);
    if (rc) return rc;
    if (!host_ptr) return fail(MSIREN_E_INVALID, "null argument");
    *host_ptr = nullptr;
    if (bytes == 0) return 0;
    HIPCHK(hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    return 0;
}

int msiren_host_free(msiren_handle h, void* host_ptr) {
    if (!h) {  // a block that has outlived its handle (msiren_destroy waited for the handle's streams: nothing of it is in flight)
        if (host_ptr) HIPCHK(hipHostFree(host_ptr));
        return 0;
    }
    int rc = check(h, false);
    if (rc) return rc;
    if (host_ptr) {
        if ((rc = sync_all(h))) return rc;  // (a copy to or from it may still be in flight)
        HIPCHK(hipHostFree(host_ptr));
    }
    return 0;
}

int msiren_memcpy_h2d(msiren_handle h, void* dst_dev, const void* src_host, size_t bytes) {
    int rc = check(h, false);
    if (rc) return rc;
    if (bytes == 0) return 0;
    if ((rc = sync_all(h))) return rc;
    const HostSrc src(src_host, bytes);
    HOSTBUF_OK(src);
    DrainOnExit drain(h);
    HIPCHK(hipMemcpyAsync(dst_dev, src.as<void>(), bytes, hipMemcpyHostToDevice, h->sc[h->cur].s));
    HIPCHK(hipStreamSynchronize(h->sc[h->cur].s));
    drain.disarm();
    return 0;
}

int msiren_memcpy_d2h(msiren_handle h, void* dst_host, const void* src_dev, size_t bytes) {
    int rc = check(h, false);
    if (rc) return rc;
    if (bytes == 0) return 0;
    if ((rc = sync_all(h))) return rc;
    const HostDst dst(dst_host, bytes);
    HOSTBUF_OK(dst);
    DrainOnExit drain(h);
    HIPCHK(hipMemcpyAsync(dst.as<void>(), src_dev, bytes, hipMemcpyDeviceToHost, h->sc[h->cur].s));
    HIPCHK(hipStreamSynchronize(h->sc[h->cur].s));
    drain.disarm();
    dst.finish();
    return 0;
}

int msiren_timer_start(msiren_handle h) {
    int rc = check(h, false);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    HIPCHK(hipEventRecord(h->ev0, h->sc[0].s));
    return 0;
}

int msiren_timer_stop(msiren_handle h, float* elapsed_ms) {
    int rc = check(h, false);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    HIPCHK(hipEventRecord(h->ev1, h->sc[0].s));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return 0;
}

int msiren_profile_enable(msiren_handle h, int32_t on) {
    int rc = check(h, false);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    h->profile = on != 0;
    h->prof_used = 0;
    h->prof_launches = 0;
    h->prof_ms = 0.0;
    h->prof_kernels.clear();
    return 0;
}

int msiren_profile_read(msiren_handle h, int64_t* launches, double* trunk_ms_total) {
    int rc = check(h, false);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    if ((rc = profile_collect(h))) return rc;
    if (launches) *launches = h->prof_launches;
    if (trunk_ms_total) *trunk_ms_total = h->prof_ms;
    return 0;
}

int msiren_profile_read_kernel(msiren_handle h, int32_t index, char* name128, int64_t* launches, double* ms_total, int64_t* coords_total) {
    int rc = check(h, false);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    if ((rc = profile_collect(h))) return rc;
    if (index < 0 || index >= (int32_t)h->prof_kernels.size())
        return fail(MSIREN_E_INVALID, "profile: %d trunk instance(s) were launched since msiren_profile_enable, index %d asked for",
                    (int)h->prof_kernels.size(), index);
    const auto& k = h->prof_kernels[index];
    if (name128) std::snprintf(name128, 128, "%s", k.name.c_str());
    if (launches) *launches = k.launches;
    if (ms_total) *ms_total = k.ms;
    if (coords_total) *coords_total = k.coords;
    return 0;
}

int msiren_last_trunk_kernel(msiren_handle h, char* name128) {
    if (!h || !name128) return fail(MSIREN_E_INVALID, "null argument");
    std::snprintf(name128, 128, "%s", h->last_trunk);
    return 0;
}

int msiren_device_info(msiren_handle h, char* name256, int32_t* cus, int32_t* mhz, uint64_t* hbm) {
    int rc = check(h, false);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->cfg.device));
    if (name256) {
        std::snprintf(name256, 256, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (mhz) *mhz = prop.clockRate / 1000;
    if (hbm) *hbm = (uint64_t)prop.totalGlobalMem;
    return 0;
}

int msiren_runtime_info(int32_t* runtime_version, int32_t* built_against, int32_t* driver_version, char* lib_path, size_t lib_path_bytes) {
    // Which HIP runtime this process's libmsiren calls end up in.  The library's only HIP dependency is NEEDED libamdhip64.so.7; a
    // PyTorch-ROCm wheel ships its own libamdhip64.so under the SAME soname (torch/lib, ROCm 7.0 in this image), so in a process that
    // imported torch first the dynamic loader resolves every hip* call of this library to torch's copy and its libhsa-runtime64 -- not
    // to /opt/rocm's.  dladdr on a HIP entry point names the file that is really mapped.
    int rv = 0, dv = 0;
    if (hipRuntimeGetVersion(&rv) != hipSuccess) { (void)hipGetLastError(); rv = 0; }
    if (hipDriverGetVersion(&dv) != hipSuccess) { (void)hipGetLastError(); dv = 0; }
    if (runtime_version) *runtime_version = rv;
    if (built_against) *built_against = HIP_VERSION;
    if (driver_version) *driver_version = dv;
    if (lib_path && lib_path_bytes) {
        Dl_info di{};
        const char* name = (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) ? di.dli_fname : "";
        std::snprintf(lib_path, lib_path_bytes, "%s", name);
    }
    return 0;
}

int msiren_host_range_kind(const void* host_ptr, size_t bytes, int32_t* kind) {
    if (!kind) return fail(MSIREN_E_INVALID, "null argument");
    void* dev = nullptr;
    *kind = (int32_t)host_range_kind(host_ptr, bytes, &dev);
    return 0;
}

int msiren_device_pci(msiren_handle h, char* busid32) {
    int rc = check(h, false);
    if (rc) return rc;
    if (!busid32) return fail(MSIREN_E_INVALID, "null argument");
    HIPCHK(hipDeviceGetPCIBusId(busid32, 32, h->cfg.device));
    return 0;
}

int msiren_range_events(msiren_handle h, int64_t* count) {
    if (!h || !count) return fail(MSIREN_E_INVALID, "null argument");
    *count = h->range_events;
    return 0;
}

int msiren_flops_per_coord(msiren_handle h, double* flops) {
    if (!h || !flops) return fail(MSIREN_E_INVALID, "null argument");
    const double H = h->H, L = h->L;
    *flops = 2.0 * 2.0 * H + (L - 1.0) * 2.0 * H * H + 2.0 * H;
    return 0;
}

}  // extern "C"
