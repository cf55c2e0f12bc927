// libmsiren.so -- host side of the C ABI declared in include/msiren.h.
//
// Owns: the device context (one device, one stream), the weight store keyed by the reference's
// state_dict names, the host-side packing of weights into kernel layouts, grow-only device
// workspaces, and the launch sequence
//     [tiling] -> encoder -> modulator -> fused SIREN trunk -> [weighted fold]
// Nothing here falls back to the CPU: every forward entry point launches HIP kernels or fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: librccl is dlopen'ed by the first msiren_comm_* call

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/msiren.h"
#include "encoder_modulator.hip.h"
#include "encoder_modulator_f16x3.hip.h"
#include "mfma_probe.hip.h"
#include "pass_queue.h"
#include "host_plan.h"
#include "weights_blob.h"
#include "siren_trunk_f16x3n.hip.h"
#include "siren_trunk_f16x3h.hip.h"
#include "siren_trunk_f16x3w.hip.h"
#include "siren_trunk_f32.hip.h"
#include "siren_trunk_x1n.hip.h"
#include "siren_trunk_x1w.hip.h"
#include "tiling.hip.h"
#include "trunk_instances.h"  // the trunk kernels are compiled in their own translation units (k_*.hip)

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(MSIREN_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct msiren_ctx {
    msiren_config cfg{};
    int H = 0, HP = 0, L = 0, Z = 0, S = 0, P = 0, O = 0, I = 0;
    // Up to three streams with private scratch: with msiren_set_streams(h, 2) consecutive *_dev forward
    // calls alternate between them, so the under-occupied tail of one call's persistent trunk kernel
    // overlaps the encoder / modulator / trunk start of the next call.  Three (round 5): call k+2's prologue no longer queues
    // behind call k's trunk -- for a trunk that OWNS its CUs (config 5: 1.76 rounds per slice) the next trunk is then ready when the
    // half-empty last round begins.
    struct StreamCtx {
        hipStream_t s = nullptr;
        DevBuf mods, modpad, latent, patches, keep, rec, queue, feat, plan;
        DevBuf cscratch;  // split-fp16 Modulator: the latent part of layers 1.., lane-private (encoder_modulator_f16x3.hip.h)
        hipEvent_t ev_join = nullptr;  // a host call that pipelines itself: this stream's chunk has been enqueued
        msiren::PassQueue pq;  // host view of the never-reset pass counter (pass_queue.h)
    } sc[3];
    int cur = 0, nstreams = 1;
    bool solo = false;     // a synchronous host-pointer call is running on ONE stream: nothing of this handle is to run beside its trunk
    // which split-fp16 trunk a launch takes: 0 = launch_trunk_f16x3's own rule; 1 = register-resident with room beside it
    // (ring of 3); 2 = weight-stationary.  Set per chunk by a host call that pipelines itself (host_plan.h).
    int trunk_force = 0;
    hipEvent_t trunk_after = nullptr;  // the next trunk launch waits for this event first (a pipelined host call: the weight-stationary
                                       // trunk of the last chunk behind the other stream's conditional launch, which cannot run beside it)
    bool em_beside = false;  // the prologue being launched runs beside a trunk of this call (a pipelined host call's chunks): shallow weight ring
    static constexpr int lin_tile_min = 1024;  // rows from which the exact-fp32 Linear layers use the 32 x 32-tile kernel (a quarter of it for >= 512 outputs)
    char last_trunk[96] = "";  // name of the trunk instance launched last (msiren_last_trunk_kernel)
    const int* plan = nullptr;  // device-side list of non-black patches in effect (slice pipeline only)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, std::vector<float>> tensors;  // state_dict, host copies
    std::map<std::string, size_t> expected;             // key -> element count
    std::vector<float> grid_host;                       // the coordinate grid in effect (state_dict's, or rebuilt)
    bool committed = false, have_modulator = false, have_encoder = false;
    // trunk
    float *d_grid = nullptr, *d_l0 = nullptr, *d_wp = nullptr, *d_bias = nullptr, *d_wout = nullptr;
    float bout = 0.f, cg0 = 0.f, cg = 0.f;
    // split-fp16 trunk (MSIREN_PREC_F16X3)
    void* d_wp16n = nullptr;  // weight stream of the 16x16x32 kernel (default)
    int lds_attr_f16n[2][4] = {};
    int lds_attr_f16h[2][2] = {};  // half-unit instances (num_layers = 5 only)
    int lds_attr_f16w[2] = {};     // weight-stationary instances ([activation])
    int lds_attr_f32[4] = {}, lds_attr_x1 = 0;  // exact-fp32 trunk ([activation][residual]) / single-product 16-bit trunk
    // f16x3 domain guard: a word in host memory the trunk kernels set when a scaled modulation does not fit fp16
    volatile int* status_host = nullptr;
    int* status_dev = nullptr;
    unsigned range_epoch = 0;      // number of the split-fp16 trunk launch in flight (what it writes to its stream's flag word)
    // Synchronous one-chunk msiren_forward_tiles calls (round 5): the host is going to wait for the stream anyway, so the trunk raises its flag in
    // HOST memory (status_host[8]) and the call looks at it after the wait -- no conditional launch (4.4 us of kernel + a launch gap per call);
    // a flagged call enqueues the exact-fp32 trunk then and waits once more (profiles/r5/12_*).  Asynchronous calls keep the conditional launch.
    bool host_check_now = false;   // set by the call for the launch_trunk it reaches
    struct { const float* mods = nullptr; int64_t B = 0; float* out = nullptr; unsigned epoch = 0; bool armed = false; } hc;
    int64_t range_events = 0;      // synchronisations that found the conditional exact-fp32 trunk had run, since create
    float* d_dump = nullptr;       // 256 floats: where lanes of the weight-stationary trunk that have nothing to store write
    int trace_host = 0;            // MSIREN_TRACE_HOST=1: msiren_forward_tiles prints the host-side timeline of the call (stderr)
    int f16_ws = 1;                // the weight-stationary trunk runs single-stream launches (MSIREN_F16_WS=0: never; tests, A/B)
    float *d_bias16 = nullptr, *d_wout16 = nullptr, *d_s0t = nullptr;
    float mscale16[16] = {0};  // 16x16 kernel: factor of each layer's modulation row (the NEXT layer's weight scale, inverted)
    bool f16x3_ready = false;
    // single-product 16-bit trunk (MSIREN_PREC_BF16 / MSIREN_PREC_F16), H = 512
    void *d_woutx1 = nullptr, *d_wpx1n = nullptr;  // last_layer.weight (fp16); weight stream of siren_trunk_x1n.hip.h
    void* d_wpx1w = nullptr;       // weight stream of siren_trunk_x1w.hip.h (weight-stationary: 64 KB per layer, N-pass and wave)
    int lds_attr_x1w = 0;

    float* d_bias32x1 = nullptr;   // bias rows: fp32, in revolutions x the layer's weight scale
    float* d_s0t512 = nullptr;
    float winvx1[64] = {0};
    bool x1_ready = false;
    int num_cus = 256;
    // environment knobs (the whole list): read ONCE, at msiren_create -- not on the launch path
    int half_allowed = 1;      // MSIREN_F16_HALF=0: never use the half-unit instance (tests: instance selection)
    int host_pipe_min = 2400;  // MSIREN_HOST_PIPE_MIN: tiles from which a host call cuts itself into chunks (below: one chunk, buffers in place; profiles/r5/04_*)
    static constexpr int host_first = 112, host_piece = 400;  // tiles in the first / the further chunks of a pipelined host call (host_plan.h)
    unsigned queue_start = 0;  // MSIREN_QUEUE_START: initial value of the never-reset pass counters (tests: wrap-around)
    // modulator: transposed weights so that consecutive threads read consecutive outputs
    float *d_modw = nullptr, *d_modb = nullptr, *d_modw_rm = nullptr;  // transposed / as stored (row-major)
    // encoder
    float *d_encw = nullptr, *d_c3w_rm = nullptr, *d_fcw_rm = nullptr;  // the latter two point into d_encw
    msiren::EncoderParams enc{};
    // encoder tail + Modulator in split-fp16 arithmetic, one launch (encoder_modulator_f16x3.hip.h); every precision but fp32
    void* d_emw = nullptr;         // packed weight streams of the four waves
    void* d_emc2 = nullptr;        // conv2's MFMA A fragments
    float* d_embias = nullptr;     // [conv3 64][fc Z][modulator L x H]
    float em_winv_c3 = 1.f, em_winv_fc = 1.f, em_winv_z[64] = {0}, em_winv_h[64] = {0};
    int em_wave_stride = 0, em_zp_start = 0;
    bool em_enc = false, em_mod = false;  // which halves of the stream are packed (the checkpoint's key set decides)
    int em_depth = 0;              // MSIREN_EM_DEPTH=2|4|8: force the weight-ring depth of the split-fp16 prologue (tests: same bits at every depth)
    int em_enabled = 1;            // MSIREN_PROLOGUE_F16X3=0: the exact-fp32 launches per layer on a split-fp16 handle (tests, A/B)
    float* d_foldw = nullptr;  // (S,S) overlap-add weights
    // workspaces
    DevBuf ws_out, ws_tiles, ws_in, ws_img;  // staging of the host-pointer entry points
    // profiling
    bool profile = false;
    int64_t prof_launches = 0;
    double prof_ms = 0.0;
    struct ProfRec { hipEvent_t a, b; int kernel; int64_t coords; };
    struct ProfKernel { std::string name; int64_t launches = 0, coords = 0; double ms = 0.0; };
    std::vector<ProfRec> prof_events;
    std::vector<ProfKernel> prof_kernels;  // totals per trunk instance since msiren_profile_enable(h, 1), in order of first launch
    size_t prof_used = 0;
    // multi-GPU: RCCL communicator this handle is a rank of (msiren_comm_*), staging buffer of its collectives
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_n = 1;
    DevBuf ws_comm;
};

namespace {

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

void declare_expected(msiren_ctx* h) {
    auto& e = h->expected;
    const size_t H = h->H, Z = h->Z, L = h->L;
    e["grid"] = (size_t)h->P * 2;
    for (size_t l = 0; l < L; ++l) {
        const std::string p = "net.layers." + std::to_string(l);
        e[p + ".weight"] = H * (l == 0 ? 2 : H);
        if (h->cfg.use_bias) e[p + ".bias"] = H;
        const std::string m = "modulator.layers." + std::to_string(l) + ".0";
        e[m + ".weight"] = H * (l == 0 ? Z : H + Z);
        e[m + ".bias"] = H;
    }
    e["net.last_layer.weight"] = H;
    if (h->cfg.use_bias) e["net.last_layer.bias"] = 1;
    const std::string en = "encoder.encoder.encoder.";
    e[en + "0.weight"] = 16 * 1 * 3 * 3;
    e[en + "0.bias"] = 16;
    e[en + "2.weight"] = 32 * 16 * 3 * 3;
    e[en + "2.bias"] = 32;
    e[en + "4.weight"] = 64 * 32 * 8 * 8;
    e[en + "4.bias"] = 64;
    e[en + "7.weight"] = Z * 64;
    e[en + "7.bias"] = Z;
}

const std::vector<float>* get(msiren_ctx* h, const std::string& k) {
    auto it = h->tensors.find(k);
    return it == h->tensors.end() ? nullptr : &it->second;
}

// ---- trunk packing --------------------------------------------------------------------------
// Everything is scaled by w0/(2*pi) in double before rounding to fp32, so that the kernel's
// accumulator is the sine argument in revolutions (see siren_trunk_f32.hip.h).
int pack_trunk(msiren_ctx* h) {
    const int H = h->H, HP = h->HP, L = h->L;
    const int TT = HP / 128, QN = HP / 8;
    const double two_pi = 6.283185307179586476925286766559;
    const double c0 = (double)h->cfg.w0_initial / two_pi, c = (double)h->cfg.w0 / two_pi;
    std::string missing;
    auto need = [&](const std::string& k) -> const std::vector<float>* {
        const auto* v = get(h, k);
        if (!v) missing += (missing.empty() ? "" : ", ") + k;
        return v;
    };
    std::vector<const std::vector<float>*> W(L), Bv(L);
    for (int l = 0; l < L; ++l) {
        W[l] = need("net.layers." + std::to_string(l) + ".weight");
        Bv[l] = h->cfg.use_bias ? need("net.layers." + std::to_string(l) + ".bias") : nullptr;
    }
    const auto* Wo = need("net.last_layer.weight");
    const auto* Bo = h->cfg.use_bias ? need("net.last_layer.bias") : nullptr;
    if (!missing.empty())
        return fail(MSIREN_E_STATE, "Missing key(s) in state_dict: %s", missing.c_str());

    std::vector<float>& grid = h->grid_host;  // the layer-0 tables of the 16-bit trunks are built from it as well
    if (const auto* g = get(h, "grid")) {
        grid = *g;
    } else {  // the reference registers it as a buffer (modulated_siren.py:427-433); rebuild it if a checkpoint lacks it
        grid.resize((size_t)h->P * 2);
        const int S = h->S;
        std::vector<float> lin(S);
        const float step = S > 1 ? (1.0f - (-1.0f)) / (float)(S - 1) : 0.f;
        for (int i = 0; i < S; ++i) lin[i] = (i < S / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(S - 1 - i));
        for (int a = 0; a < S; ++a)
            for (int b2 = 0; b2 < S; ++b2) {
                grid[((size_t)a * S + b2) * 2 + 0] = lin[a];
                grid[((size_t)a * S + b2) * 2 + 1] = lin[b2];
            }
    }

    std::vector<float> l0((size_t)HP * 4, 0.f);
    for (int f = 0; f < H; ++f) {
        l0[(size_t)f * 4 + 0] = (float)((double)(*W[0])[(size_t)f * 2 + 0] * c0);
        l0[(size_t)f * 4 + 1] = (float)((double)(*W[0])[(size_t)f * 2 + 1] * c0);
        l0[(size_t)f * 4 + 2] = Bv[0] ? (float)((double)(*Bv[0])[f] * c0) : 0.f;
    }
    const int nh = L > 1 ? L - 1 : 0;
    std::vector<float> wp((size_t)std::max(nh, 1) * 4 * QN * TT * 256, 0.f);
    std::vector<float> bias((size_t)std::max(nh, 1) * HP, 0.f);
    for (int l = 1; l < L; ++l) {
        const std::vector<float>& w = *W[l];
        for (int wave = 0; wave < 4; ++wave)
            for (int q = 0; q < QN; ++q)
                for (int tt = 0; tt < TT; ++tt)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int f = wave * 32 * TT + 32 * tt + (lane & 31);
                        float* dst = &wp[(((((size_t)(l - 1) * 4 + wave) * QN + q) * TT + tt) * 64 + lane) * 4];
                        for (int j = 0; j < 4; ++j) {
                            const int k = 8 * q + 4 * (lane >> 5) + j;
                            dst[j] = (f < H && k < H) ? (float)((double)w[(size_t)f * H + k] * c) : 0.f;
                        }
                    }
        if (Bv[l])
            for (int f = 0; f < H; ++f) bias[(size_t)(l - 1) * HP + f] = (float)((double)(*Bv[l])[f] * c);
    }
    std::vector<float> wout(HP, 0.f);
    for (int f = 0; f < H; ++f) wout[f] = (float)((double)(*Wo)[f] * c);
    h->bout = Bo ? (float)((double)(*Bo)[0] * c) : 0.f;
    // Morlet: exp(-0.5 p^2) with p = r * 2pi / w  ->  exp2(cg * r^2)
    const double log2e = 1.4426950408889634;
    h->cg0 = (float)(-0.5 * log2e * (two_pi / h->cfg.w0_initial) * (two_pi / h->cfg.w0_initial));
    h->cg = (float)(-0.5 * log2e * (two_pi / h->cfg.w0) * (two_pi / h->cfg.w0));

    int rc;
    if ((rc = upload(&h->d_grid, grid))) return rc;
    if ((rc = upload(&h->d_l0, l0))) return rc;
    if ((rc = upload(&h->d_wp, wp))) return rc;
    if ((rc = upload(&h->d_bias, bias))) return rc;
    if ((rc = upload(&h->d_wout, wout))) return rc;
    return 0;
}

// ---- split-fp16 trunk packing ------------------------------------------------------------------
// Chunk (layer l, feature tile t) = [16 k-steps][hi|lo][64 lanes][8 x f16]; lane (r = lane&31, h = lane>>5),
// element j of k-step s multiplies feature  kf = 32*(s>>1) + 16*(s&1) + 8*(j>>2) + 4*h + (j&3)  of the
// previous layer -- the order in which the previous layer's accumulator registers hold them.
// Weights are scaled by w0/2pi and by 2^e (e per layer, max|W| -> [8192, 16384)) before the split.
uint16_t f32_to_f16_rne(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}
float f16_to_f32(uint16_t u) {
    _Float16 h;
    std::memcpy(&h, &u, 2);
    return (float)h;
}

int pack_trunk_f16x3(msiren_ctx* h) {
    h->f16x3_ready = false;
    const int H = h->H, L = h->L;
    if (h->cfg.precision != MSIREN_PREC_F16X3 || H != 256 || L < 2 || msiren::F16Lds<3>::total(L) > 160 * 1024) return 0;
    const double two_pi = 6.283185307179586476925286766559;
    const double c = (double)h->cfg.w0 / two_pi;
    std::vector<uint16_t> wpn((size_t)(L - 1) * 8 * 16 * 2 * 64 * 8);
    std::vector<float> bias((size_t)(L - 1) * 256, 0.f), wout(256, 0.f);
    for (int l = 1; l < L; ++l) {
        const std::vector<float>& w = *get(h, "net.layers." + std::to_string(l) + ".weight");
        double mx = 0.0;
        for (float v : w) mx = std::max(mx, std::fabs((double)v * c));
        // 16x16x32 kernel (siren_trunk_f16x3n.hip.h): chunk (l, t) = [8 k-steps][2 sub-tiles][hi|lo][64 lanes][8 x f16];
        // lane (r = lane & 15, q = lane >> 4), element j of k-step s of sub-tile u: output feature 32 t + 16 u + r,
        // input feature 32 s + 16 (j >> 2) + 4 q + (j & 3).  Scale 2^a with rms|W'| ~ 0.1 (a is undone on the
        // activation side, through the previous layer's modulation row, so the accumulator is the sine argument).
        {
            double sq = 0.0;
            for (float v : w) sq += ((double)v * c) * ((double)v * c);
            const double rmsw = std::sqrt(sq / (double)w.size());
            int a = 0;
            if (rmsw > 0.0) a = (int)std::lround(std::log2(0.1 / rmsw));
            if (mx > 0.0) a = std::min(a, (int)std::floor(std::log2(32768.0 / mx)));  // stay inside fp16
            a = std::max(-14, std::min(a, 30));
            const double scn = std::ldexp(c, a);
            h->mscale16[l - 1] = (float)std::ldexp(1.0, -a);  // row l-1 of the modulation table
            for (int t = 0; t < 8; ++t)
                for (int s2 = 0; s2 < 8; ++s2)
                    for (int u = 0; u < 2; ++u)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int f = 32 * t + 16 * u + (lane & 15);
                                const int k = 32 * s2 + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
                                const float ws = (float)((double)w[(size_t)f * H + k] * scn);
                                const uint16_t hi = f32_to_f16_rne(ws);
                                const uint16_t lo = f32_to_f16_rne(ws - f16_to_f32(hi));
                                const size_t base = (((((size_t)(l - 1) * 8 + t) * 8 + s2) * 2 + u) * 2) * 64 * 8;
                                wpn[base + (size_t)lane * 8 + j] = hi;
                                wpn[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
                            }
        }
        if (const auto* b = h->cfg.use_bias ? get(h, "net.layers." + std::to_string(l) + ".bias") : nullptr)
            for (int f = 0; f < H; ++f) bias[(size_t)(l - 1) * 256 + f] = (float)((double)(*b)[f] * c);
    }
    h->mscale16[L - 1] = 1.0f;  // the last hidden layer's output meets last_layer unscaled
    const auto* Wo = get(h, "net.last_layer.weight");
    for (int f = 0; f < H; ++f) wout[f] = (float)((double)(*Wo)[f] * c);
    if (h->d_wp16n) HIPCHK(hipFree(h->d_wp16n));
    h->d_wp16n = nullptr;
    HIPCHK(hipMalloc(&h->d_wp16n, wpn.size() * 2));
    HIPCHK(hipMemcpy(h->d_wp16n, wpn.data(), wpn.size() * 2, hipMemcpyHostToDevice));
    int rc;
    if ((rc = upload(&h->d_bias16, bias))) return rc;
    if ((rc = upload(&h->d_wout16, wout))) return rc;
    {   // layer-0 activation table S0T[f/4][p][f%4] = act0(w0_initial * (W0 x_p + b0)), fp64 -> fp32
        const auto& W0 = *get(h, "net.layers.0.weight");
        const auto* B0 = h->cfg.use_bias ? get(h, "net.layers.0.bias") : nullptr;
        const std::vector<float>& grid = h->grid_host;  // pack_trunk ran first
        if (grid.size() != (size_t)h->P * 2) return fail(MSIREN_E_STATE, "grid buffer missing");
        std::vector<float> tab((size_t)64 * h->P * 4);
        const bool morlet = h->cfg.activation == MSIREN_ACT_MORLET;
        for (int f = 0; f < 256; ++f)
            for (int pidx = 0; pidx < h->P; ++pidx) {
                // the pre-activation is formed in fp32 like F.linear does, the activation in fp64
                const float pre = std::fmaf(grid[(size_t)pidx * 2 + 1], W0[(size_t)f * 2 + 1],
                                            std::fmaf(grid[(size_t)pidx * 2], W0[(size_t)f * 2], B0 ? (*B0)[f] : 0.f));
                double a = std::sin((double)h->cfg.w0_initial * (double)pre);
                if (morlet) a *= std::exp(-0.5 * (double)pre * (double)pre);
                tab[((size_t)(f / 4) * h->P + pidx) * 4 + (f & 3)] = (float)a;
            }
        if ((rc = upload(&h->d_s0t, tab))) return rc;
    }
    h->f16x3_ready = true;
    return 0;
}

// ---- single-product 16-bit trunk packing (H = 512) ---------------------------------------------
uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int pack_trunk_x1(msiren_ctx* h) {
    h->x1_ready = false;
    const int H = h->H, L = h->L;
    const bool bf = h->cfg.precision == MSIREN_PREC_BF16;
    if (!(h->cfg.precision == MSIREN_PREC_BF16 || h->cfg.precision == MSIREN_PREC_F16)) return 0;
    // (the kernel launch_trunk_x1_kernel will pick: weight-stationary from 3 layers on, depths 2..11; register-resident 2..10)
    const int lds_need = L >= 3 ? msiren::X1wLds::total(L) : msiren::X1nLds<3>::total(L);
    if (H != 512 || L < 2 || L > 65 || lds_need > 160 * 1024)
        return fail(MSIREN_E_INVALID, "precision bf16/f16 (single-product trunk) needs dim_hidden = 512 and 2 <= num_layers <= 11: its tables must fit the 160 KB LDS; got H=%d L=%d", H, L);
    const double two_pi = 6.283185307179586476925286766559;
    const double c = (double)h->cfg.w0 / two_pi;
    std::vector<uint16_t> wpn((size_t)(L - 1) * 16 * 16 * 2 * 64 * 8), wout(512, 0);  // chunk (l, t) = [16 k-steps][2 sub-tiles][64 lanes][8]
    std::vector<float> bias32((size_t)(L - 1) * 512, 0.f);
    std::vector<uint16_t> wpw(wpn.size());
    for (int l = 1; l < L; ++l) {
        const std::vector<float>& w = *get(h, "net.layers." + std::to_string(l) + ".weight");
        int e = 0;
        if (!bf) {  // fp16: scale max|W| into [8192, 16384); bf16 has fp32's exponent range
            double mx = 0.0;
            for (float v : w) mx = std::max(mx, std::fabs((double)v * c));
            if (mx > 0.0) e = std::max(-14, std::min((int)std::floor(std::log2(16384.0 / mx)), 30));
        }
        const double sc = std::ldexp(c, e);
        h->winvx1[l - 1] = (float)std::ldexp(1.0, -e);
        // lane (r = lane & 15, q = lane >> 4), element j of k-step s of sub-tile u: output feature 32 t + 16 u + r, input
        // feature 32 s + 16 (j >> 2) + 4 q + (j & 3) (the order of siren_trunk_f16x3n.hip.h)
        for (int t = 0; t < 16; ++t)
            for (int s = 0; s < 16; ++s)
                for (int u = 0; u < 2; ++u)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int f = 32 * t + 16 * u + (lane & 15);
                            const int k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
                            const float ws = (float)((double)w[(size_t)f * H + k] * sc);
                            wpn[(((((size_t)(l - 1) * 16 + t) * 16 + s) * 2 + u) * 64 + lane) * 8 + j] = bf ? f32_to_bf16_rne(ws) : f32_to_f16_rne(ws);
                        }
        // weight-stationary kernel: block ((l - 1) * 2 + n) * 4 + wave = [16 k-steps][4 tiles][64 lanes][8]: output feature
        // 256 n + 64 wave + 16 t + r, input feature as above
        for (int n = 0; n < 2; ++n)
            for (int wv = 0; wv < 4; ++wv)
                for (int s = 0; s < 16; ++s)
                    for (int t = 0; t < 4; ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int f = 256 * n + 64 * wv + 16 * t + (lane & 15);
                                const int k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
                                const float ws = (float)((double)w[(size_t)f * H + k] * sc);
                                wpw[((((((size_t)(l - 1) * 2 + n) * 4 + wv) * 16 + s) * 4 + t) * 64 + lane) * 8 + j] = bf ? f32_to_bf16_rne(ws) : f32_to_f16_rne(ws);
                            }
        if (const auto* b = h->cfg.use_bias ? get(h, "net.layers." + std::to_string(l) + ".bias") : nullptr)
            for (int f = 0; f < H; ++f) bias32[(size_t)(l - 1) * 512 + f] = (float)((double)(*b)[f] * sc);  // (x 2^e: the accumulator is scaled like the weights)
    }
    const auto* Wo = get(h, "net.last_layer.weight");
    for (int f = 0; f < H; ++f) wout[f] = f32_to_f16_rne((float)((double)(*Wo)[f] * c));
    const auto& W0 = *get(h, "net.layers.0.weight");
    const auto* B0 = h->cfg.use_bias ? get(h, "net.layers.0.bias") : nullptr;
    const std::vector<float>* g = &h->grid_host;  // pack_trunk ran first
    if (g->size() != (size_t)h->P * 2) return fail(MSIREN_E_STATE, "grid buffer missing");
    std::vector<float> tab((size_t)128 * h->P * 4);  // layer-0 activation table S0T[f/4][p][f%4] = act0(w0_initial * (W0 x_p + b0))
    const bool morlet = h->cfg.activation == MSIREN_ACT_MORLET;
    for (int f = 0; f < 512; ++f)
        for (int pidx = 0; pidx < h->P; ++pidx) {
            const float pre = std::fmaf((*g)[(size_t)pidx * 2 + 1], W0[(size_t)f * 2 + 1],
                                        std::fmaf((*g)[(size_t)pidx * 2], W0[(size_t)f * 2], B0 ? (*B0)[f] : 0.f));
            double a = std::sin((double)h->cfg.w0_initial * (double)pre);
            if (morlet) a *= std::exp(-0.5 * (double)pre * (double)pre);
            tab[((size_t)(f / 4) * h->P + pidx) * 4 + (f & 3)] = (float)a;
        }
    auto up16 = [&](void** dst, const std::vector<uint16_t>& v) -> int {
        if (*dst) HIPCHK(hipFree(*dst));
        *dst = nullptr;
        HIPCHK(hipMalloc(dst, v.size() * 2));
        HIPCHK(hipMemcpy(*dst, v.data(), v.size() * 2, hipMemcpyHostToDevice));
        return 0;
    };
    int rc;
    if ((rc = up16(&h->d_wpx1n, wpn)) || (rc = up16(&h->d_wpx1w, wpw)) || (rc = up16(&h->d_woutx1, wout))) return rc;
    if ((rc = upload(&h->d_bias32x1, bias32)) || (rc = upload(&h->d_s0t512, tab))) return rc;
    h->x1_ready = true;
    return 0;
}

// ---- modulator / encoder packing ------------------------------------------------------------
int pack_modulator(msiren_ctx* h) {
    const int H = h->H, Z = h->Z, L = h->L;
    // transposed: Wt[l][k][f], k over [hidden(H) ; latent(Z)] (layer 0: latent only), so that a
    // thread per output feature reads consecutive addresses
    size_t total = 0;
    for (int l = 0; l < L; ++l) total += (size_t)(l == 0 ? Z : H + Z) * H;
    std::vector<float> wt(total), bb((size_t)L * H);
    size_t off = 0;
    for (int l = 0; l < L; ++l) {
        const auto* w = get(h, "modulator.layers." + std::to_string(l) + ".0.weight");
        const auto* b = get(h, "modulator.layers." + std::to_string(l) + ".0.bias");
        if (!w || !b) return 1;  // not present: latent/tiles entry points stay unavailable
        const int K = (l == 0 ? Z : H + Z);
        for (int f = 0; f < H; ++f)
            for (int k = 0; k < K; ++k) wt[off + (size_t)k * H + f] = (*w)[(size_t)f * K + k];
        for (int f = 0; f < H; ++f) bb[(size_t)l * H + f] = (*b)[f];
        off += (size_t)K * H;
    }
    std::vector<float> rm(total);
    off = 0;
    for (int l = 0; l < L; ++l) {
        const auto* w = get(h, "modulator.layers." + std::to_string(l) + ".0.weight");
        std::copy(w->begin(), w->end(), rm.begin() + off);
        off += w->size();
    }
    int rc;
    if ((rc = upload(&h->d_modw, wt))) return rc;
    if ((rc = upload(&h->d_modw_rm, rm))) return rc;
    if ((rc = upload(&h->d_modb, bb))) return rc;
    return 0;
}

int pack_encoder(msiren_ctx* h) {
    const std::string en = "encoder.encoder.encoder.";
    const char* keys[8] = {"0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias", "7.weight", "7.bias"};
    const std::vector<float>* t[8];
    for (int i = 0; i < 8; ++i) {
        t[i] = get(h, en + keys[i]);
        if (!t[i]) return 1;
    }
    const int Z = h->Z;
    // one blob: [c1w 16x9][c1b 16][c2w (144,32) transposed][c2b 32][c3w (2048,64) transposed][c3b 64]
    //           [fcw (64,Z) transposed][fcb Z]
    std::vector<float> blob;
    auto push = [&](const std::vector<float>& v) {
        size_t o = blob.size();
        blob.insert(blob.end(), v.begin(), v.end());
        while (blob.size() % 4) blob.push_back(0.f);
        return o;
    };
    msiren::EncoderParams ep{};
    size_t o_c1w = push(*t[0]);
    size_t o_c1b = push(*t[1]);
    std::vector<float> c2t((size_t)144 * 32);
    for (int o = 0; o < 32; ++o)
        for (int k = 0; k < 144; ++k) c2t[(size_t)k * 32 + o] = (*t[2])[(size_t)o * 144 + k];
    size_t o_c2w = push(c2t);
    size_t o_c2b = push(*t[3]);
    std::vector<float> c3t((size_t)2048 * 64);
    for (int o = 0; o < 64; ++o)
        for (int k = 0; k < 2048; ++k) c3t[(size_t)k * 64 + o] = (*t[4])[(size_t)o * 2048 + k];
    size_t o_c3w = push(c3t);
    size_t o_c3b = push(*t[5]);
    std::vector<float> fct((size_t)64 * Z);
    for (int o = 0; o < Z; ++o)
        for (int k = 0; k < 64; ++k) fct[(size_t)k * Z + o] = (*t[6])[(size_t)o * 64 + k];
    size_t o_fcw = push(fct);
    size_t o_fcb = push(*t[7]);
    size_t o_c3rm = push(*t[4]);  // (64, 2048) and (Z, 64) as stored: operands of the batched MFMA GEMMs
    size_t o_fcrm = push(*t[6]);
    int rc;
    if ((rc = upload(&h->d_encw, blob))) return rc;
    ep.c1w = h->d_encw + o_c1w;
    ep.c1b = h->d_encw + o_c1b;
    ep.c2w = h->d_encw + o_c2w;
    ep.c2b = h->d_encw + o_c2b;
    ep.c3w = h->d_encw + o_c3w;
    ep.c3b = h->d_encw + o_c3b;
    ep.fcw = h->d_encw + o_fcw;
    ep.fcb = h->d_encw + o_fcb;
    ep.Z = Z;
    h->enc = ep;
    h->d_c3w_rm = h->d_encw + o_c3rm;
    h->d_fcw_rm = h->d_encw + o_fcrm;
    return 0;
}

// ---- encoder tail + Modulator, split-fp16 (encoder_modulator_f16x3.hip.h) ------------------------------------------------
// Per wave one stream of k-steps in the order the kernel consumes them, each [tile 0 hi | tile 0 lo | tile 1 hi | tile 1 lo]
// x [64 lanes][8 x f16]; lane (m = lane & 15, q = lane >> 4), element j: output feature 16 T + m, input
// k(s, q, j) = 32 s + 16 (j >> 2) + 4 q + (j & 3) of k-step s.  Sections: conv3 (32 k-steps: the wave's K half of its tile
// pair), Linear(64, Z) (NPZ passes x 4 k-steps, the upper two zero), the latent part of every Modulator layer (L NPH passes
// x Z / 32), the hidden part of layers 1.. ((L - 1) NPH passes x H / 32).  Each layer is scaled by the power of two that
// brings max|W| into [2^13, 2^14) before the hi / lo split.
int pack_prologue_f16x3(msiren_ctx* h) {
    h->em_enc = h->em_mod = false;
    const int H = h->H, Z = h->Z, L = h->L;
    const bool enc = h->have_encoder && h->O == 32, mod = h->have_modulator;  // (a trunk + Modulator checkpoint has no encoder.* keys)
    if (!h->em_enabled || h->cfg.precision == MSIREN_PREC_F32 || (!enc && !mod)) return 0;
    if (!((H == 256 && Z == 256) || (H == 512 && Z == 128)) || L > 64) return 0;  // the instantiated (NPH, NPZ) pairs
    const int NPH = H / 128, NPZ = Z / 128, KH = H / 32, KZ = Z / 32;
    auto scale_of = [](const float* w, size_t n0, size_t stride, size_t rows, size_t cols) {  // exponent a: max|w| 2^a in [2^13, 2^14)
        double mx = 0.0;
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) mx = std::max(mx, std::fabs((double)w[n0 + r * stride + c]));
        if (!(mx > 0.0) || !std::isfinite(mx)) return 0;
        int e = 0;
        (void)std::frexp(mx, &e);  // mx = f 2^e, f in [0.5, 1)
        return std::max(-100, std::min(100, 14 - e));
    };
    const msiren::EmStreamLayout lay = msiren::em_stream_layout(NPH, NPZ, L, enc, mod, msiren::EM_C3_KSTEPS / 2, msiren::EM_FC_KSTEPS);  // (host_plan.h)
    const int zp_start = lay.zp_start, nk = lay.total;
    std::vector<uint16_t> ws(((size_t)4 * nk + msiren::EM_MAX_DEPTH) * 4 * 64 * 8, 0);  // (+ padding: the ring prefetches past the end)
    auto put = [&](int wave, int g, int t, int lane, int j, double v) {  // v already scaled
        const float f = (float)v;
        const uint16_t hi = f32_to_f16_rne(f), lo = f32_to_f16_rne(f - f16_to_f32(hi));
        const size_t base = (((size_t)wave * nk + g) * 4 + 2 * t) * 64 * 8 + (size_t)lane * 8 + j;
        ws[base] = hi;
        ws[base + 64 * 8] = lo;
    };
    auto kin = [](int s, int q, int j) { return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3); };
    std::vector<float> bias((size_t)64 + Z + (size_t)L * H, 0.f);
    if (enc) {
    const std::string en = "encoder.encoder.encoder.";
    const std::vector<float>&W3 = *get(h, en + "4.weight"), &B3 = *get(h, en + "4.bias"), &Wf = *get(h, en + "7.weight"), &Bf = *get(h, en + "7.bias");
    // conv2 as MFMA A fragments: lane (m, q), element j of k-step s: channel 16 mt + m, tap 2 s + (q >> 1), input channel 8 (q & 1) + j
    {
        const std::vector<float>& W2 = *get(h, en + "2.weight");  // (32, 16, 3, 3)
        const int a2 = scale_of(W2.data(), 0, 144, 32, 144);
        std::vector<uint16_t> c2((size_t)2 * 5 * 2 * 64 * 8, 0);
        for (int mt = 0; mt < 2; ++mt)
            for (int ks = 0; ks < 5; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int tap = 2 * ks + (lane >> 5), ci = 8 * ((lane >> 4) & 1) + j, o = 16 * mt + (lane & 15);
                        const float f = tap < 9 ? (float)std::ldexp((double)W2[(size_t)o * 144 + ci * 9 + tap], a2) : 0.f;
                        const uint16_t hi = f32_to_f16_rne(f), lo = f32_to_f16_rne(f - f16_to_f32(hi));
                        const size_t base = ((size_t)(mt * 5 + ks) * 2) * 64 * 8 + (size_t)lane * 8 + j;
                        c2[base] = hi;
                        c2[base + 64 * 8] = lo;
                    }
        if (h->d_emc2) HIPCHK(hipFree(h->d_emc2));
        h->d_emc2 = nullptr;
        HIPCHK(hipMalloc(&h->d_emc2, c2.size() * 2));
        HIPCHK(hipMemcpy(h->d_emc2, c2.data(), c2.size() * 2, hipMemcpyHostToDevice));
        h->enc.c2f16 = h->d_emc2;
        h->enc.c2_winv = (float)std::ldexp(1.0, -a2);
    }
    // conv3: its k order is the order in which the conv kernel's threads hold the features (encoder_conv_f16x3_kernel<VARIANT>)
    const int a3 = scale_of(W3.data(), 0, 2048, 64, 2048);
    h->em_winv_c3 = (float)std::ldexp(1.0, -a3);
    for (int wave = 0; wave < 4; ++wave)
        for (int ks = 0; ks < msiren::EM_C3_KSTEPS / 2; ++ks)
            for (int t = 0; t < 2; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int s2 = (msiren::EM_C3_KSTEPS / 2) * (wave >> 1) + ks, f = 32 * (wave & 1) + 16 * t + (lane & 15);
                        int k;  // torch's flattened (channel, position) index of element (k-step s2, q = lane >> 4, j) of the conv kernel's images
                        {
                            const int cw = s2 >> 4, cl = 4 * (s2 & 15) + (lane >> 4);  // the conv kernel's (wave, lane) that stored this piece
                            k = (16 * (cw & 1) + 4 * (cl >> 4) + (j & 3)) * 64 + 16 * (2 * (cw >> 1) + (j >> 2)) + (cl & 15);
                        }
                        put(wave, ks, t, lane, j, std::ldexp((double)W3[(size_t)f * 2048 + k], a3));
                    }
    for (int f = 0; f < 64; ++f) bias[f] = B3[f];
    const int af = scale_of(Wf.data(), 0, 64, Z, 64);
    h->em_winv_fc = (float)std::ldexp(1.0, -af);
    for (int wave = 0; wave < 4; ++wave)
        for (int pz = 0; pz < NPZ; ++pz)
            for (int ks = 0; ks < 2; ++ks)  // (k-steps 2, 3 of a pass stay zero)
                for (int t = 0; t < 2; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int f = 128 * pz + 32 * wave + 16 * t + (lane & 15);
                            put(wave, msiren::EM_C3_KSTEPS / 2 + pz * msiren::EM_FC_KSTEPS + ks, t, lane, j,
                                std::ldexp((double)Wf[(size_t)f * 64 + kin(ks, lane >> 4, j)], af));
                        }
    for (int f = 0; f < Z; ++f) bias[64 + f] = Bf[f];
    }
    for (int l = 0; l < L && mod; ++l) {
        const std::vector<float>& W = *get(h, "modulator.layers." + std::to_string(l) + ".0.weight");
        const std::vector<float>& Bm = *get(h, "modulator.layers." + std::to_string(l) + ".0.bias");
        const int Kh = l == 0 ? 0 : H, K = Kh + Z;
        const int az = scale_of(W.data(), (size_t)Kh, (size_t)K, H, Z);
        h->em_winv_z[l] = (float)std::ldexp(1.0, -az);
        for (int wave = 0; wave < 4; ++wave)
            for (int ph = 0; ph < NPH; ++ph)
                for (int ks = 0; ks < KZ; ++ks)
                    for (int t = 0; t < 2; ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int f = 128 * ph + 32 * wave + 16 * t + (lane & 15);
                                put(wave, zp_start + (l * NPH + ph) * KZ + ks, t, lane, j,
                                    std::ldexp((double)W[(size_t)f * K + Kh + kin(ks, lane >> 4, j)], az));
                            }
        if (l > 0) {
            const int ah = scale_of(W.data(), 0, (size_t)K, H, H);
            h->em_winv_h[l] = (float)std::ldexp(1.0, -ah);
            for (int wave = 0; wave < 4; ++wave)
                for (int ph = 0; ph < NPH; ++ph)
                    for (int ks = 0; ks < KH; ++ks)
                        for (int t = 0; t < 2; ++t)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j) {
                                    const int f = 128 * ph + 32 * wave + 16 * t + (lane & 15);
                                    put(wave, zp_start + L * NPH * KZ + ((l - 1) * NPH + ph) * KH + ks, t, lane, j,
                                        std::ldexp((double)W[(size_t)f * K + kin(ks, lane >> 4, j)], ah));
                                }
        }
        for (int f = 0; f < H; ++f) bias[(size_t)64 + Z + (size_t)l * H + f] = Bm[f];
    }
    if (h->d_emw) HIPCHK(hipFree(h->d_emw));
    h->d_emw = nullptr;
    HIPCHK(hipMalloc(&h->d_emw, ws.size() * 2));
    HIPCHK(hipMemcpy(h->d_emw, ws.data(), ws.size() * 2, hipMemcpyHostToDevice));
    int rc;
    if ((rc = upload(&h->d_embias, bias))) return rc;
    h->em_wave_stride = nk * 256;
    h->em_zp_start = zp_start;
    h->em_enc = enc;
    h->em_mod = mod;
    return 0;
}

int pack_fold_weights(msiren_ctx* h) {
    // w[i][j] = exp(-0.1 * dist((i,j), centre)) / max   (src/util/tiling.py:67-88; fp64 maths
    // rounded to fp32 element-wise, then divided by the fp32 maximum, as the reference does)
    const int S = h->S;
    std::vector<float> w((size_t)S * S);
    const double c = (S - 1) / 2.0;
    float mx = 0.f;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            const double d = std::sqrt((i - c) * (i - c) + (j - c) * (j - c));
            w[(size_t)i * S + j] = (float)std::exp(-0.1 * d);
            mx = std::max(mx, w[(size_t)i * S + j]);
        }
    for (auto& v : w) v = v / mx;
    return upload(&h->d_foldw, w);
}

// ---- launches ---------------------------------------------------------------------------------
template <int HP>
int launch_trunk_hp(msiren_ctx* h, const msiren::TrunkParams& p, int grid) {
    const size_t lds = (size_t)HP * 256 + (size_t)HP * 16;  // X image + layer-0 rows
    const int act = h->cfg.activation, res = h->cfg.residual;
#define MSIREN_LAUNCH(A, R)                                                                        \
    do {                                                                                           \
        auto k = msiren::siren_trunk_f32_kernel<HP, A, R>;                                         \
        int& done = h->lds_attr_f32[(A) * 2 + (R)]; /* one instance per handle: HP, activation and residual are the handle's */ \
        if (lds > 64 * 1024 && done < (int)lds) {                                                  \
            HIPCHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            done = (int)lds;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, h->sc[h->cur].s, p);                           \
        std::snprintf(h->last_trunk, sizeof h->last_trunk, "siren_trunk_f32_kernel<%d,%d,%d>", HP, A, R); \
    } while (0)
    if (act == MSIREN_ACT_MORLET) {
        if (res) MSIREN_LAUNCH(1, 1); else MSIREN_LAUNCH(1, 0);
    } else {
        if (res) MSIREN_LAUNCH(0, 1); else MSIREN_LAUNCH(0, 0);
    }
#undef MSIREN_LAUNCH
    HIPCHK(hipGetLastError());
    return 0;
}

msiren::TrunkParams make_trunk_params(msiren_ctx* h, const float* mods, int stride, int64_t B, float* out_dev) {
    msiren::TrunkParams p{};
    p.grid = h->d_grid;
    p.l0 = h->d_l0;
    p.wp = h->d_wp;
    p.bias = h->d_bias;
    p.wout = h->d_wout;
    p.mods = mods;
    p.out = out_dev;
    p.bout = h->bout;
    p.cg0 = h->cg0;
    p.cg = h->cg;
    p.B = (int)B;
    p.P = h->P;
    p.L = h->L;
    p.mod_stride = stride;
    p.chunks = (h->P + 63) / 64;
    p.stamps = nullptr;
    p.plan = h->plan;
    return p;
}

// Pass queue of the persistent trunks.  Workgroup g starts with pass g; every executed pass performs exactly
// one atomicAdd on the counter, so a launch of n passes advances it by n: the counter is never reset, the
// host hands each launch the value it will find (no memset node per call).  The host value moves only once
// the launch has been accepted (queue_launched); a failure in between leaves it where the device counter is.
int ensure_queue(msiren_ctx* h) {
    auto& c = h->sc[h->cur];
    if (c.queue.p) return 0;
    int rc = ensure(h, c.queue, 256);
    if (rc) return rc;
    // test knob: start the never-reset counter just below 2^32 (or 2^31) to exercise its wrap-around
    const unsigned start = h->queue_start;
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)c.queue.p, (int)start, 16, c.s));          // [0..15]: the pass counter's line
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)((int*)c.queue.p + 16), 0, 48, c.s));     // [16]: the domain guard's flag word; [32]: the slice pipeline's ticket counter
    c.pq.reset(start);
    return 0;
}

int queue_for_launch(msiren_ctx* h, int64_t npasses, int** counter, unsigned* base) {
    auto& c = h->sc[h->cur];
    int rc = ensure_queue(h);
    if (rc) return rc;
    *counter = (int*)c.queue.p;
    *base = c.pq.begin(npasses);
    return 0;
}

int queue_launched(msiren_ctx* h, int rc) {
    if (rc == 0) h->sc[h->cur].pq.commit();
    else h->sc[h->cur].pq.abort();
    return rc;
}

// After a launch whose number of passes only the device knows (black patches skipped): reset the counter.
int queue_reset_after_plan_launch(msiren_ctx* h, bool by_the_next_kernel = false) {
    auto& c = h->sc[h->cur];
    if (!c.queue.p) return 0;
    if (!by_the_next_kernel) HIPCHK(hipMemsetAsync(c.queue.p, 0, 4, c.s));  // (else: weighted_fold_kernel's reset_word)
    c.pq.reset(0);
    return 0;
}

template <int R>
int launch_trunk_f16x3n_r(msiren_ctx* h, const msiren::TrunkF16Params& p, int grid) {
    const int lds = msiren::F16Lds<R>::total(h->L);
    const bool mor = h->cfg.activation == MSIREN_ACT_MORLET;
    const bool l5 = h->L == 5;  // the YAML depth has its own straight-line instance (siren_trunk_f16x3n.hip.h: LFIX)
    using Kern = void (*)(msiren::TrunkF16Params);
    const Kern k = l5 ? (mor ? (Kern)msiren::siren_trunk_f16x3n_kernel<1, R, 5> : (Kern)msiren::siren_trunk_f16x3n_kernel<0, R, 5>)
                      : (mor ? (Kern)msiren::siren_trunk_f16x3n_kernel<1, R, 0> : (Kern)msiren::siren_trunk_f16x3n_kernel<0, R, 0>);
    int& done = h->lds_attr_f16n[R == 4 ? 1 : 0][(mor ? 1 : 0) + (l5 ? 2 : 0)];
    if (done < lds) {
        HIPCHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        done = lds;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, h->sc[h->cur].s, p);
    HIPCHK(hipGetLastError());
    std::snprintf(h->last_trunk, sizeof h->last_trunk, "siren_trunk_f16x3n_kernel<%d,%d,%d>", mor ? 1 : 0, R, l5 ? 5 : 0);
    return 0;
}

// half-unit instance (siren_trunk_f16x3h.hip.h): 16 coordinates per wave; depth-5 models only
template <int R>
int launch_trunk_f16x3h_r(msiren_ctx* h, const msiren::TrunkF16Params& p, int grid) {
    const int lds = msiren::F16Lds<R>::total(h->L);
    const bool mor = h->cfg.activation == MSIREN_ACT_MORLET;
    using Kern = void (*)(msiren::TrunkF16Params);
    const Kern k = mor ? (Kern)msiren::siren_trunk_f16x3h_kernel<1, R, 5> : (Kern)msiren::siren_trunk_f16x3h_kernel<0, R, 5>;
    int& done = h->lds_attr_f16h[R == 4 ? 1 : 0][mor ? 1 : 0];
    if (done < lds) {
        HIPCHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        done = lds;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, h->sc[h->cur].s, p);
    HIPCHK(hipGetLastError());
    std::snprintf(h->last_trunk, sizeof h->last_trunk, "siren_trunk_f16x3h_kernel<%d,%d,5>", mor ? 1 : 0, R);
    return 0;
}

// weight-stationary trunk (siren_trunk_f16x3w.hip.h): passes of 2..4 units, laid out by ws_schedule
int launch_trunk_f16x3w(msiren_ctx* h, const float* mods_dev, int64_t B, float* out_dev) {
    msiren::TrunkWsParams p{};
    if (!h->d_dump) HIPCHK(hipMalloc((void**)&h->d_dump, 256 * sizeof(float)));
    p.dump = h->d_dump;
    p.s0t = h->d_s0t;
    p.wp = (const _Float16*)h->d_wp16n;
    p.bias = h->d_bias16;
    p.wout = h->d_wout16;
    p.mods = mods_dev;
    p.out = out_dev;
    for (int i = 0; i < 16; ++i) p.mscale[i] = h->mscale16[i];
    p.bout = h->bout;
    p.cg0 = h->cg0;
    p.cg = h->cg;
    p.B = (int)B;
    p.P = h->P;
    p.L = h->L;
    p.plan = h->plan;
    const int upp = (h->P + 31) / 32;
    const int64_t units = B * upp;
    if (units > 0x3fffffffLL) return fail(MSIREN_E_INVALID, "batch too large for one launch: B=%lld", (long long)B);
    p.units_per_patch = upp;
    p.unit_base = 0;
    p.total_units = (int)units;
    {   // unit / upp as a multiply-high: k = 30 + ceil(log2 upp), m = ceil(2^k / upp) (exact for units < 2^30)
        int lg = 0;
        while ((1 << lg) < upp) ++lg;
        p.div_k = 30 + lg;
        p.div_m = (unsigned)(((1ULL << p.div_k) + (unsigned)upp - 1) / (unsigned)upp);
    }
    // small batches: one pass of 2 units per workgroup (latency); otherwise one workgroup per CU
    const int grid = (int)std::min<int64_t>(h->num_cus, (units + 1) / 2);
    const msiren::WsSchedule sch = msiren::ws_schedule(units, grid);
    int rc = queue_for_launch(h, sch.npasses(), &p.pass_counter, &p.pass_base);
    if (rc) return rc;
    p.status = h->host_check_now ? h->status_dev + 8 : p.pass_counter + 16;  // the stream's flag word, behind the pass counter's line (or the host's)
    p.status_val = (int)h->range_epoch;
    const int lds = msiren::WsLds<4>::total(h->L);
    const bool mor = h->cfg.activation == MSIREN_ACT_MORLET;
    using Kern = void (*)(msiren::TrunkWsParams);
    const Kern k = mor ? (Kern)msiren::siren_trunk_f16x3w_kernel<1, 4> : (Kern)msiren::siren_trunk_f16x3w_kernel<0, 4>;
    int& done = h->lds_attr_f16w[mor ? 1 : 0];
    if (done < lds) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return queue_launched(h, fail(MSIREN_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e)));
        done = lds;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, h->sc[h->cur].s, p);
    hipError_t e = hipGetLastError();
    std::snprintf(h->last_trunk, sizeof h->last_trunk, "siren_trunk_f16x3w_kernel<%d,4>", mor ? 1 : 0);
    return queue_launched(h, e == hipSuccess ? 0 : fail(MSIREN_E_HIP, "trunk launch: %s", hipGetErrorString(e)));
}

// The weight-stationary trunk is the faster kernel on its own (it owns the whole register file and LDS of its CUs, so
// nothing can run beside it); with two streams the register-resident trunk wins because the next call's encoder and
// modulator run beside it.  Depths 3..5 (its unit images + tables must fit the LDS); modulation buffer below 4 GB.
bool ws_capable(msiren_ctx* h, int64_t B) {
    static_assert(msiren::WsLds<4>::total(msiren::WS_MAX_L) <= 160 * 1024, "unit images + tables of the deepest supported model must fit the LDS");
    return h->f16_ws && h->L >= msiren::WS_MIN_L && h->L <= msiren::WS_MAX_L && (int64_t)h->L * B * 256 * 4 < (1LL << 32);
}
bool use_f16x3w(msiren_ctx* h, int64_t B) { return ws_capable(h, B) && (h->nstreams == 1 || h->solo); }

int launch_trunk_f16x3(msiren_ctx* h, const float* mods_dev, int64_t B, float* out_dev) {
    const int upp_ = (h->P + 31) / 32;
    // (small batches of depth-5 models keep the half-unit instance: twice the waves, lower latency)
    if (h->trunk_force == 2 ||
        (h->trunk_force == 0 && use_f16x3w(h, B) && !(h->L == 5 && h->half_allowed && !h->plan && B * upp_ <= 2 * (int64_t)h->num_cus)))
        return launch_trunk_f16x3w(h, mods_dev, B, out_dev);
    msiren::TrunkF16Params p{};
    p.grid = h->d_grid;
    p.l0 = h->d_l0;
    p.s0t = h->d_s0t;
    p.wp = (const _Float16*)h->d_wp16n;
    p.bias = h->d_bias16;
    p.wout = h->d_wout16;
    p.mods = mods_dev;
    p.out = out_dev;
    for (int i = 0; i < 16; ++i) p.winv[i] = h->mscale16[i];
    p.bout = h->bout;
    p.cg0 = h->cg0;
    p.cg = h->cg;
    p.B = (int)B;
    p.P = h->P;
    p.L = h->L;
    p.plan = h->plan;
    const int upp = (h->P + 31) / 32;
    const int64_t units = B * upp;
    if (units > 0x3fffffffLL) return fail(MSIREN_E_INVALID, "batch too large for one launch: B=%lld", (long long)B);
    // R = 3 leaves ~35 KB of LDS per CU free, enough for an encoder / modulator workgroup of the NEXT
    // call (other stream) to run beside the persistent trunk workgroup; R = 4 fills the CU.
    int ring = ((h->nstreams > 1 && !h->solo) || h->trunk_force == 1) ? 3 : 4;
    // depths other than 5 run the loop form of the kernel: with a ring of 3 hipcc gives it all 512 registers (and 188 bytes of
    // scratch per lane), so nothing could run beside it anyway -- the ring of 4 has neither (164 + 240 registers)
    if (ring == 3 && h->L != 5) ring = 4;
    const bool r4 = ring >= 4 && msiren::F16Lds<4>::total(h->L) <= 160 * 1024;
    const int cus = h->num_cus;

    // One launch of a piece of the batch: units [base, base + count) of `per_wave` coordinates each.  The pass queue
    // (workgroup g starts with pass g, further passes come from the counter) is claimed per launch.
    auto launch_piece = [&](bool half, int64_t base, int64_t count) -> int {
        p.units_per_patch = half ? (h->P + 15) / 16 : upp;
        p.unit_base = (int)base;
        p.total_units = (int)count;
        const int64_t passes = (count + 3) / 4;
        const int grid = (int)std::min<int64_t>(cus, passes);
        int rc = queue_for_launch(h, passes, &p.pass_counter, &p.pass_base);
        if (rc) return rc;
        p.status = h->host_check_now ? h->status_dev + 8 : p.pass_counter + 16;  // the stream's flag word, behind the pass counter's line (or the host's)
        p.status_val = (int)h->range_epoch;
        if (half) return queue_launched(h, r4 ? launch_trunk_f16x3h_r<4>(h, p, grid) : launch_trunk_f16x3h_r<3>(h, p, grid));
        return queue_launched(h, r4 ? launch_trunk_f16x3n_r<4>(h, p, grid) : launch_trunk_f16x3n_r<3>(h, p, grid));
    };

    // Half-unit instance (16 coordinates per wave, twice the waves) for small batches: everything fits in one round even
    // as half-units, so the extra waves are free and the latency drops (a single tile: 76 -> 66 us).  Needs the unit count
    // on the host (no black-tile plan) and the depth-5 instance.
    // Measured and dropped, twice: running the ragged last round of a big launch (one 320x320 slice = 7.03 rounds of
    // 256 x 4 waves) as half-units so that the main launch's workgroups finish together -- (1) as a second launch behind
    // the main one on the same stream: 0.306 vs 0.295 ms per slice; (2) queued beside it on the handle's idle second
    // stream (event fork / join, no launch gap): 0.315 vs 0.289 ms.  A half-unit pass on an otherwise idle chip is not
    // half a round (its weight-fragment reads are those of a full unit; prologue and layer 0 do not shrink), and the
    // cross-stream dependency costs more than the tail it removes.
    const bool half_ok = !h->plan && h->L == 5 && h->half_allowed;
    if (half_ok && units <= 2 * (int64_t)cus) return launch_piece(true, 0, B * ((h->P + 15) / 16));
    return launch_piece(false, 0, units);
}

int launch_trunk_x1_kernel(msiren_ctx* h, const msiren::TrunkX1Params& p, int grid);

int launch_trunk_x1(msiren_ctx* h, const float* mods_dev, int64_t B, float* out_dev) {
    msiren::TrunkX1Params p{};
    p.s0t = h->d_s0t512;
    p.wp = (const unsigned short*)h->d_wpx1n;
    p.bias32 = h->d_bias32x1;
    p.wout = (const _Float16*)h->d_woutx1;
    p.mods = mods_dev;
    p.out = out_dev;
    for (int i = 0; i < 64; ++i) p.winv[i] = h->winvx1[i];
    p.bout = h->bout;
    p.cg0 = h->cg0;
    p.cg = h->cg;
    p.B = (int)B;
    p.P = h->P;
    p.L = h->L;
    p.units_per_patch = (h->P + 31) / 32;
    const int64_t units = B * p.units_per_patch;
    if (units > 0x7fffffffLL) return fail(MSIREN_E_INVALID, "batch too large for one launch: B=%lld", (long long)B);
    p.total_units = (int)units;
    p.plan = h->plan;
    const bool ws = h->L >= 3;
    // (one-stream handles: the balanced grid -- the same rounds on fewer CUs, 1 % faster alone; two streams: every CU, so that the
    //  next call's trunk can start in the half-empty last round -- measured 111.2 against 109.3 Mpixel/s, profiles/r4/09_*)
    const int cus_x1 = h->num_cus;
    const bool balance = ws && (h->nstreams == 1 || h->solo);
    const int grid = balance ? msiren::x1w_balanced_grid(units, cus_x1) : (int)std::min<int64_t>(cus_x1, (units + 3) / 4);
    // (the weight-stationary kernel lays its passes out itself: x1w_schedule, 4-unit passes and a last round of 2-unit ones)
    msiren::X1wSchedule sch = msiren::x1w_schedule(units, grid);
    const int64_t npasses = ws ? (int64_t)sch.n4 + sch.n2 : (units + 3) / 4;
    int rc = queue_for_launch(h, npasses, &p.pass_counter, &p.pass_base);
    if (rc) return rc;
    p.status = p.pass_counter + 16;  // the stream's flag word, behind the pass counter's line (fp16 operands: the domain guard)
    p.status_val = (int)h->range_epoch;
    return queue_launched(h, launch_trunk_x1_kernel(h, p, grid));
}

int launch_trunk_x1_kernel(msiren_ctx* h, const msiren::TrunkX1Params& p0, int grid) {
    const bool bf = h->cfg.precision == MSIREN_PREC_BF16, mor = h->cfg.activation == MSIREN_ACT_MORLET, res = h->cfg.residual != 0;
    msiren::TrunkX1Params p = p0;
    if (h->L >= 3) {  // weight-stationary (siren_trunk_x1w.hip.h; its layer pipeline needs a hidden layer before the final one)
        p.wp = (const unsigned short*)h->d_wpx1w;
        const int lds = msiren::X1wLds::total(h->L);
#define MSIREN_X1W_LAUNCH(BF, A, RS)                                                                 \
    do {                                                                                             \
        auto k = msiren::siren_trunk_x1w_kernel<BF, A, RS>;                                          \
        if (h->lds_attr_x1w < lds) {                                                                 \
            HIPCHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
            h->lds_attr_x1w = lds;                                                                   \
        }                                                                                            \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, h->sc[h->cur].s, p);                       \
        std::snprintf(h->last_trunk, sizeof h->last_trunk, "siren_trunk_x1w_kernel<%d,%d,%d>", BF, A, RS); \
    } while (0)
        if (bf) {
            if (mor) { if (res) MSIREN_X1W_LAUNCH(1, 1, 1); else MSIREN_X1W_LAUNCH(1, 1, 0); }
            else     { if (res) MSIREN_X1W_LAUNCH(1, 0, 1); else MSIREN_X1W_LAUNCH(1, 0, 0); }
        } else {
            if (mor) { if (res) MSIREN_X1W_LAUNCH(0, 1, 1); else MSIREN_X1W_LAUNCH(0, 1, 0); }
            else     { if (res) MSIREN_X1W_LAUNCH(0, 0, 1); else MSIREN_X1W_LAUNCH(0, 0, 0); }
        }
#undef MSIREN_X1W_LAUNCH
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int lds = msiren::X1nLds<3>::total(h->L);
#define MSIREN_X1N_LAUNCH(BF, A, RS)                                                                 \
    do {                                                                                             \
        auto k = msiren::siren_trunk_x1n_kernel<BF, A, RS, 3>;                                       \
        if (h->lds_attr_x1 < lds) { /* one instance per handle (precision, activation, residual are the handle's) */ \
            HIPCHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
            h->lds_attr_x1 = lds;                                                                    \
        }                                                                                            \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, h->sc[h->cur].s, p);                       \
        std::snprintf(h->last_trunk, sizeof h->last_trunk, "siren_trunk_x1n_kernel<%d,%d,%d,3>", BF, A, RS); \
    } while (0)
    if (bf) {
        if (mor) { if (res) MSIREN_X1N_LAUNCH(1, 1, 1); else MSIREN_X1N_LAUNCH(1, 1, 0); }
        else     { if (res) MSIREN_X1N_LAUNCH(1, 0, 1); else MSIREN_X1N_LAUNCH(1, 0, 0); }
    } else {
        if (mor) { if (res) MSIREN_X1N_LAUNCH(0, 1, 1); else MSIREN_X1N_LAUNCH(0, 1, 0); }
        else     { if (res) MSIREN_X1N_LAUNCH(0, 0, 1); else MSIREN_X1N_LAUNCH(0, 0, 0); }
    }
#undef MSIREN_X1N_LAUNCH
    HIPCHK(hipGetLastError());
    return 0;
}

bool use_f16x3(msiren_ctx* h) {
    return h->cfg.precision == MSIREN_PREC_F16X3 && h->f16x3_ready && !h->cfg.residual &&
           msiren::F16Lds<3>::total(h->L) <= 160 * 1024;
}

// msiren_profile_enable: a HIP event pair around every trunk launch, on the stream it is launched on
int profile_begin(msiren_ctx* h, hipEvent_t* end_event) {
    *end_event = nullptr;
    if (!h->profile) return 0;
    if (h->prof_used == h->prof_events.size()) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        h->prof_events.push_back({a, b, -1, 0});
    }
    HIPCHK(hipEventRecord(h->prof_events[h->prof_used].a, h->sc[h->cur].s));
    *end_event = h->prof_events[h->prof_used].b;
    h->prof_used++;
    return 0;
}

// closes the pair profile_begin opened: the launch in between was h->last_trunk over `coords` coordinates
int profile_end(msiren_ctx* h, hipEvent_t end_event, int64_t coords) {
    if (!end_event) return 0;
    HIPCHK(hipEventRecord(end_event, h->sc[h->cur].s));
    auto& r = h->prof_events[h->prof_used - 1];
    int k = 0;
    for (; k < (int)h->prof_kernels.size(); ++k)
        if (h->prof_kernels[k].name == h->last_trunk) break;
    if (k == (int)h->prof_kernels.size()) {
        h->prof_kernels.emplace_back();
        h->prof_kernels.back().name = h->last_trunk;
    }
    r.kernel = k;
    r.coords = coords;
    return 0;
}

// Behind every split-fp16 trunk launch, on the same stream: the exact-fp32 trunk over the same batch as a conditional launch
// (siren_trunk_f32_cond_kernel: 32 KB of LDS, <= 96 registers, so that it fits beside a register-resident trunk of the other
// stream) -- its <= 2 workgroups per CU read the stream's flag word and leave unless the f16x3 launch
// wrote its number there (a scaled modulation beyond fp16, a NaN / inf).  So the output buffer always holds what the
// reference's fp32 arithmetic computes (modulated_siren.py:215-233), on the asynchronous API as well; the flag in host memory is
// informational (msiren_range_events).
int launch_trunk_f32_cond(msiren_ctx* h, const float* mods_dev, int64_t B, float* out_dev, const int* flag_word = nullptr, unsigned flag_val = 0) {
    auto& c = h->sc[h->cur];
    const int cpp = (h->P + 31) / 32;
    if (B * (int64_t)cpp > 0x7fffffffLL) return fail(MSIREN_E_INVALID, "batch too large for one launch: B=%lld", (long long)B);
    msiren::TrunkParams p = make_trunk_params(h, mods_dev, h->H, B, out_dev);  // (f16x3 needs H = 256 = HP: no padding of the rows)
    p.cond = flag_word ? flag_word : (const int*)c.queue.p + 16;
    p.cond_val = (int)(flag_word ? flag_val : h->range_epoch);
    p.items = (int)(B * cpp);
    p.host_flag = h->status_dev;
    const int grid = (int)std::min<int64_t>(p.items, (int64_t)h->num_cus);
    if (h->cfg.activation == MSIREN_ACT_MORLET)
        hipLaunchKernelGGL(msiren::siren_trunk_f32_cond_kernel<1>, dim3(grid), dim3(256), 0, c.s, p);
    else
        hipLaunchKernelGGL(msiren::siren_trunk_f32_cond_kernel<0>, dim3(grid), dim3(256), 0, c.s, p);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_trunk(msiren_ctx* h, const float* mods_dev, int64_t B, float* out_dev) {
    if (B == 0) return 0;
    if (h->trunk_after) {
        HIPCHK(hipStreamWaitEvent(h->sc[h->cur].s, h->trunk_after, 0));
        h->trunk_after = nullptr;
    }
    if (use_f16x3(h) || h->x1_ready) {
        hipEvent_t e1 = nullptr;
        {
            int rc = profile_begin(h, &e1);
            if (rc) return rc;
        }
        const bool x1_f16 = h->x1_ready && h->cfg.precision == MSIREN_PREC_F16;  // (bf16 has fp32's exponent range: nothing to guard)
        if ((!h->x1_ready || x1_f16) && ++h->range_epoch == 0) h->range_epoch = 1;  // this launch's number (never 0: the flag word's rest state; unsigned: wraps)
        int rc = h->x1_ready ? launch_trunk_x1(h, mods_dev, B, out_dev) : launch_trunk_f16x3(h, mods_dev, B, out_dev);
        if (rc) return rc;
        if ((rc = profile_end(h, e1, B * h->P))) return rc;
        if (h->host_check_now && !h->x1_ready) {  // (the caller looks at the flag in host memory behind its wait for the stream)
            h->hc.mods = mods_dev;
            h->hc.B = B;
            h->hc.out = out_dev;
            h->hc.epoch = h->range_epoch;
            h->hc.armed = true;
            return 0;
        }
        if (x1_f16) {  // H = 512: the 64-coordinate exact-fp32 trunk as the conditional launch (its workgroups read the flag word and leave)
            const int chunks = (h->P + 63) / 64;
            if (B * (int64_t)chunks > 0x7fffffffLL) return fail(MSIREN_E_INVALID, "batch too large for one launch: B=%lld", (long long)B);
            msiren::TrunkParams p = make_trunk_params(h, mods_dev, h->H, B, out_dev);
            p.cond = (const int*)h->sc[h->cur].queue.p + 16;
            p.cond_val = (int)h->range_epoch;
            p.host_flag = h->status_dev;
            char keep[sizeof h->last_trunk];
            std::memcpy(keep, h->last_trunk, sizeof keep);  // (the profile names the 16-bit trunk, not its stand-in)
            rc = launch_trunk_hp<512>(h, p, (int)(B * chunks));
            std::memcpy(h->last_trunk, keep, sizeof keep);
            return rc;
        }
        return h->x1_ready ? 0 : launch_trunk_f32_cond(h, mods_dev, B, out_dev);
    }
    const int chunks = (h->P + 63) / 64;
    if (B * (int64_t)chunks > 0x7fffffffLL) return fail(MSIREN_E_INVALID, "batch too large for one launch: B=%lld", (long long)B);
    const float* mods = mods_dev;
    int stride = h->H;
    if (h->HP != h->H) {  // zero-pad the feature axis once so the kernel can use float4 loads
        int rc = ensure(h, h->sc[h->cur].modpad, (size_t)h->L * B * h->HP * sizeof(float));
        if (rc) return rc;
        const int64_t n = (int64_t)h->L * B * h->HP;
        hipLaunchKernelGGL(msiren::pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->sc[h->cur].s,
                           mods_dev, (float*)h->sc[h->cur].modpad.p, (int64_t)h->L * B, h->H, h->HP);
        HIPCHK(hipGetLastError());
        mods = (const float*)h->sc[h->cur].modpad.p;
        stride = h->HP;
    }
    msiren::TrunkParams p = make_trunk_params(h, mods, stride, B, out_dev);
    const int grid = (int)(B * chunks);

    hipEvent_t e1 = nullptr;
    int rc = profile_begin(h, &e1);
    if (rc) return rc;
    switch (h->HP) {
        case 128: rc = launch_trunk_hp<128>(h, p, grid); break;
        case 256: rc = launch_trunk_hp<256>(h, p, grid); break;
        case 384: rc = launch_trunk_hp<384>(h, p, grid); break;
        case 512: rc = launch_trunk_hp<512>(h, p, grid); break;
        default: return fail(MSIREN_E_INVALID, "dim_hidden=%d (padded %d) is not supported by the fp32 trunk (max 512)", h->H, h->HP);
    }
    if (rc) return rc;
    return profile_end(h, e1, B * h->P);
}

// One Linear layer over the batch on the matrix cores: 16 x 16 output tiles (latency sizes) or 32 x 32 (throughput sizes:
// half the operand bytes per FLOP).  Same arithmetic either way -- an output does not depend on the batch it came in.
int launch_linear(msiren_ctx* h, const msiren::ModulatorMfmaParams& mp) {
    hipStream_t s = h->sc[h->cur].s;
    // (default threshold: 1024 rows; a quarter of it for layers of >= 512 outputs -- at 400 rows the 16 x 16 kernel launches 800 workgroups
    //  per 512-wide layer and takes 10.8 us, the tiled one is 1.7 % of a config-5 step faster; 256-wide layers: 2.7 % slower.  Same bits.)
    const int tile_min = mp.H < 512 ? h->lin_tile_min : h->lin_tile_min / 4;
    // (the tiled kernel addresses rows with 32-bit element offsets: beyond 2^32 elements per operand the 16 x 16 kernel, same bits)
    const bool fits32 = (uint64_t)mp.B * (uint64_t)std::max(std::max(mp.Z, mp.H), mp.Kh) < (1ULL << 32);
    if (mp.B >= tile_min && fits32) {
        dim3 grid((unsigned)((mp.B + 31) / 32), (unsigned)((mp.H + 31) / 32));
        hipLaunchKernelGGL((msiren::linear_mfma_tile_kernel<2, 2>), grid, dim3(256), 0, s, mp);
    } else {
        dim3 grid((unsigned)((mp.B + 15) / 16), (unsigned)(mp.H / 16));
        hipLaunchKernelGGL(msiren::modulator_layer_mfma_kernel, grid, dim3(256), 0, s, mp);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// encoder tail + Modulator in ONE launch (plus the conv kernel in front when tiles are given): split-fp16 arithmetic,
// a row block of 16 patches per workgroup through every layer (encoder_modulator_f16x3.hip.h).
//   tiles -> [z_out] -> [mods]     (tiles_dev given)        z_in -> mods     (tiles_dev null)
template <int NPH, int NPZ>
int launch_prologue_f16x3_t(msiren_ctx* h, const float* tiles_dev, const float* z_in, int64_t B, float* z_out, float* mods_dev) {
    auto& c = h->sc[h->cur];
    const int64_t nblk = (B + msiren::EM_ROWS - 1) / msiren::EM_ROWS, rows16 = nblk * msiren::EM_ROWS;
    if (nblk > 0x7fffffffLL) return fail(MSIREN_E_INVALID, "batch too large for one launch: B=%lld", (long long)B);
    int rc;
    msiren::EmTailParams p{};
    if (tiles_dev) {
        if ((rc = ensure(h, c.feat, (size_t)rows16 * 2048 * 4 + (size_t)rows16 * 4 + msiren::EM_MAX_DEPTH * 2048))) return rc;  // (+ padding: conv3's B ring prefetches past the end)
        p.feat = (const msiren::em_u4*)c.feat.p;
        p.feat_inv = (const float*)((const char*)c.feat.p + (size_t)rows16 * 2048 * 4 + msiren::EM_MAX_DEPTH * 2048);
        h->enc.plan = h->plan;
        float* const finv = (float*)((char*)c.feat.p + (size_t)rows16 * 2048 * 4 + msiren::EM_MAX_DEPTH * 2048);
        hipLaunchKernelGGL(msiren::encoder_conv_f16x3_kernel<1>, dim3((unsigned)B), dim3(256), 0, c.s, h->enc, tiles_dev, (msiren::em_u4*)c.feat.p, finv);
        HIPCHK(hipGetLastError());
    }
    if (mods_dev) {
        if ((rc = ensure(h, c.cscratch, (size_t)nblk * std::max(1, h->L - 1) * NPH * 512 * 16))) return rc;
        p.cscratch = (msiren::em_f4*)c.cscratch.p;
    }
    p.wstream = (const msiren::em_u4*)h->d_emw;
    p.bias = h->d_embias;
    p.z_in = z_in;
    p.z_out = z_out;
    p.mods = mods_dev;
    p.winv_c3 = h->em_winv_c3;
    p.winv_fc = h->em_winv_fc;
    for (int l = 0; l < 64; ++l) {
        p.winv_z[l] = h->em_winv_z[l];
        p.winv_h[l] = h->em_winv_h[l];
    }
    p.B = (int)B;
    p.L = h->L;
    p.wave_stride = h->em_wave_stride;
    p.zp_start = h->em_zp_start;
    p.count = h->plan;
    const int lds = msiren::em_tail_lds_bytes<NPH, NPZ>();
    // ring depth 4 (more weight fragments in flight per wave) where the workgroups have their CUs to themselves; depth 2 (<= 96
    // registers, 33 KB of LDS) where they run beside the register-resident trunk of the other stream or many to a CU.  Same bits.
    const bool alone = (h->nstreams == 1 || h->solo) && !h->em_beside;
    int depth = alone ? (nblk <= (int64_t)h->num_cus ? 8 : 4) : 2;
    if (h->em_depth) depth = h->em_depth;
    // latency sizes of the H = 256 model: 64 more workgroups (8 per XCD) that only pull the 2.9 MB weight stream into the L2s (EmTailParams)
    p.row_blocks = (int)nblk;
    if (NPH == 2 && alone && nblk <= 64 && tiles_dev && mods_dev) {
        p.pf_blocks = 64;
        p.pf_lines = (unsigned)(((size_t)h->em_wave_stride * 4 * 16 / 8 + 1023) / 1024);
    }
    // (the halves alone -- model.encoder(tiles), model.modulator(z) -- have the ring of 4 only)
    const dim3 grid((unsigned)(nblk + p.pf_blocks)), wg(256);
    hipStream_t st = c.s;
    if (tiles_dev && !mods_dev) hipLaunchKernelGGL((msiren::latent_mods_f16x3_kernel<NPH, NPZ, 4, 1>), grid, wg, lds, st, p);
    else if (!tiles_dev) hipLaunchKernelGGL((msiren::latent_mods_f16x3_kernel<NPH, NPZ, 4, 2>), grid, wg, lds, st, p);
    else if constexpr (NPH > 2) {  // H = 512 (config 5): 12.6 MB of weights per workgroup; nothing runs beside its trunk anyway
        if (depth >= 8) hipLaunchKernelGGL((msiren::latent_mods_f16x3_kernel<NPH, NPZ, 8, 3>), grid, wg, lds, st, p);
        else hipLaunchKernelGGL((msiren::latent_mods_f16x3_kernel<NPH, NPZ, 4, 3>), grid, wg, lds, st, p);
    }
    else if (depth >= 8) hipLaunchKernelGGL((msiren::latent_mods_f16x3_kernel<NPH, NPZ, 8, 3>), grid, wg, lds, st, p);
    else if (depth >= 4) hipLaunchKernelGGL((msiren::latent_mods_f16x3_kernel<NPH, NPZ, 4, 3>), grid, wg, lds, st, p);
    else hipLaunchKernelGGL((msiren::latent_mods_f16x3_kernel<NPH, NPZ, 2, 3>), grid, wg, lds, st, p);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_prologue_f16x3(msiren_ctx* h, const float* tiles_dev, const float* z_in, int64_t B, float* z_out, float* mods_dev) {
    if (B == 0) return 0;
    if (h->H == 256) return launch_prologue_f16x3_t<2, 2>(h, tiles_dev, z_in, B, z_out, mods_dev);
    return launch_prologue_f16x3_t<4, 1>(h, tiles_dev, z_in, B, z_out, mods_dev);
}

int launch_modulator(msiren_ctx* h, const float* z_dev, int64_t B, float* mods_dev) {
    if (B == 0) return 0;
    if (!h->have_modulator) return fail(MSIREN_E_STATE, "modulator.* weights were not loaded");
    if (h->em_mod) return launch_prologue_f16x3(h, nullptr, z_dev, B, nullptr, mods_dev);
    size_t off = 0;
    const bool mfma_ok = (h->H % 16 == 0) && (h->Z % 16 == 0);
    for (int l = 0; l < h->L && mfma_ok; ++l) {
        const int Kh = (l == 0 ? 0 : h->H);
        msiren::ModulatorMfmaParams mp{};
        mp.w = h->d_modw_rm + off;
        mp.bias = h->d_modb + (size_t)l * h->H;
        mp.hprev = l == 0 ? nullptr : mods_dev + (size_t)(l - 1) * B * h->H;
        mp.z = z_dev;
        mp.out = mods_dev + (size_t)l * B * h->H;
        mp.B = (int)B;
        mp.H = h->H;
        mp.Z = h->Z;
        mp.Kh = Kh;
        mp.act = msiren::LIN_ACT_RELU;
        mp.count = h->plan;
        int rc = launch_linear(h, mp);
        if (rc) return rc;
        off += (size_t)(Kh + h->Z) * h->H;
    }
    if (mfma_ok) return 0;
    off = 0;
    for (int l = 0; l < h->L; ++l) {
        const int Kh = (l == 0 ? 0 : h->H);
        msiren::ModulatorLayerParams mp{};
        mp.wt = h->d_modw + off;
        mp.bias = h->d_modb + (size_t)l * h->H;
        mp.hprev = l == 0 ? nullptr : mods_dev + (size_t)(l - 1) * B * h->H;
        mp.z = z_dev;
        mp.out = mods_dev + (size_t)l * B * h->H;
        mp.B = (int)B;
        mp.H = h->H;
        mp.Z = h->Z;
        mp.Kh = Kh;
        mp.count = h->plan;
        dim3 grid((unsigned)((B + msiren::MOD_ROWS - 1) / msiren::MOD_ROWS), (unsigned)((h->H + 63) / 64));
        const size_t lds = (size_t)msiren::MOD_ROWS * (Kh + h->Z) * sizeof(float);
        hipLaunchKernelGGL(msiren::modulator_layer_kernel, grid, dim3(256), lds, h->sc[h->cur].s, mp);
        HIPCHK(hipGetLastError());
        off += (size_t)(Kh + h->Z) * h->H;
    }
    return 0;
}

int launch_encoder(msiren_ctx* h, const float* tiles_dev, int64_t B, float* z_dev) {
    if (B == 0) return 0;
    if (!h->have_encoder) return fail(MSIREN_E_STATE, "encoder.* weights were not loaded");
    if (h->em_enc) return launch_prologue_f16x3(h, tiles_dev, nullptr, B, z_dev, nullptr);
    hipStream_t s = h->sc[h->cur].s;
    h->enc.plan = h->plan;
    // small batches are launch-latency bound: one fused per-tile kernel instead of three launches
    if (h->Z % 16 != 0 || B < 48) {
        hipLaunchKernelGGL(msiren::encoder_kernel, dim3((unsigned)B), dim3(256), 0, s, h->enc, tiles_dev, z_dev);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // conv1+conv2 per tile, then conv3 == Linear(2048, 64) and Linear(64, Z) as GEMMs over the batch
    auto& c = h->sc[h->cur];
    int rc = ensure(h, c.feat, (size_t)B * (2048 + 64) * sizeof(float));
    if (rc) return rc;
    float* feat = (float*)c.feat.p;
    float* a3 = feat + (size_t)B * 2048;
    hipLaunchKernelGGL(msiren::encoder_conv_kernel, dim3((unsigned)B), dim3(256), 0, s, h->enc, tiles_dev, feat);
    HIPCHK(hipGetLastError());
    msiren::ModulatorMfmaParams mp{};
    mp.w = h->d_c3w_rm;
    mp.bias = h->enc.c3b;
    mp.z = feat;
    mp.out = a3;
    mp.B = (int)B;
    mp.H = 64;
    mp.Z = 2048;
    mp.act = msiren::LIN_ACT_LEAKY02;
    mp.count = h->plan;
    if ((rc = launch_linear(h, mp))) return rc;
    mp.w = h->d_fcw_rm;
    mp.bias = h->enc.fcb;
    mp.z = a3;
    mp.out = z_dev;
    mp.H = h->Z;
    mp.Z = 64;
    mp.act = msiren::LIN_ACT_NONE;
    return launch_linear(h, mp);
}

// encoder + modulator: tiles -> latent -> modulations
int launch_encoder_modulator(msiren_ctx* h, const float* tiles_dev, int64_t B, float* z_dev, float* mods_dev) {
    if (B == 0) return 0;
    if (h->em_enc && h->em_mod) return launch_prologue_f16x3(h, tiles_dev, nullptr, B, nullptr, mods_dev);  // (the latent stays in the workgroup)
    int rc = launch_encoder(h, tiles_dev, B, z_dev);
    if (rc) return rc;
    return launch_modulator(h, z_dev, B, mods_dev);
}

int forward_latent_dev(msiren_ctx* h, const float* z_dev, int64_t B, float* out_dev, float* mods_out_dev) {
    float* mods = mods_out_dev;
    if (!mods) {
        int rc = ensure(h, h->sc[h->cur].mods, (size_t)h->L * B * h->H * sizeof(float));
        if (rc) return rc;
        mods = (float*)h->sc[h->cur].mods.p;
    }
    int rc = launch_modulator(h, z_dev, B, mods);
    if (rc) return rc;
    return launch_trunk(h, mods, B, out_dev);
}

int forward_tiles_dev(msiren_ctx* h, const float* tiles_dev, int64_t B, float* out_dev) {
    int rc = ensure(h, h->sc[h->cur].latent, (size_t)B * h->Z * sizeof(float));
    if (rc) return rc;
    rc = ensure(h, h->sc[h->cur].mods, (size_t)h->L * B * h->H * sizeof(float));
    if (rc) return rc;
    float* mods = (float*)h->sc[h->cur].mods.p;
    rc = launch_encoder_modulator(h, tiles_dev, B, (float*)h->sc[h->cur].latent.p, mods);
    if (rc) return rc;
    return launch_trunk(h, mods, B, out_dev);
}

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
#define HOSTBUF_OK(x) do { if (!(x).ok()) return fail(MSIREN_E_HIP, "no page-locked memory for a bounce buffer"); } while (0)

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

// the f16x3 domain guard's flag in host memory: raised by a conditional exact-fp32 trunk launch that had to run
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

DrainOnExit::~DrainOnExit() {
    if (!armed || !h) return;
    for (auto& c : h->sc)
        if (c.s) (void)hipStreamSynchronize(c.s);
}

// Host-pointer (synchronous) calls.  (Until round 3 a call whose f16x3 trunk raised the domain flag was run again on the
// exact-fp32 trunk from here; since round 4 the re-run is a conditional launch on the stream itself, for every entry point.)
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

int check(msiren_ctx* h, bool need_commit = true) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    if (need_commit && !h->committed) return fail(MSIREN_E_STATE, "weights not committed: call msiren_set_tensor for every net.* key, then msiren_commit_weights");
    return use_device(h);
}


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

// =================================================================================================
extern "C" {

int msiren_abi_version(void) { return MSIREN_ABI_VERSION; }

const char* msiren_last_error(void) { return g_err.c_str(); }

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
    if (h->comm) (void)msiren_comm_destroy(h);
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
    HIPCHK(hipMemcpyAsync(h->sc[h->cur].mods.p, src.as<float>(), nm, hipMemcpyHostToDevice, h->sc[h->cur].s));
    if ((rc = launch_trunk(h, (const float*)h->sc[h->cur].mods.p, B, (float*)h->ws_out.p))) return rc;
    HIPCHK(hipMemcpyAsync(dst.as<float>(), h->ws_out.p, no, hipMemcpyDeviceToHost, h->sc[h->cur].s));
    HIPCHK(hipStreamSynchronize(h->sc[h->cur].s));
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
    HIPCHK(hipMemcpyAsync(h->sc[h->cur].latent.p, src.as<float>(), nz, hipMemcpyHostToDevice, h->sc[h->cur].s));
    if ((rc = forward_latent_dev(h, (const float*)h->sc[h->cur].latent.p, B, (float*)h->ws_out.p, (float*)h->sc[h->cur].mods.p))) return rc;
    HIPCHK(hipMemcpyAsync(dst.as<float>(), h->ws_out.p, no, hipMemcpyDeviceToHost, h->sc[h->cur].s));
    if (mods_out_host) HIPCHK(hipMemcpyAsync(dst_mods.as<float>(), h->sc[h->cur].mods.p, nm, hipMemcpyDeviceToHost, h->sc[h->cur].s));
    HIPCHK(hipStreamSynchronize(h->sc[h->cur].s));
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
    HIPCHK(hipMemcpyAsync(h->ws_tiles.p, src.as<float>(), nt, hipMemcpyHostToDevice, c.s));
    if ((rc = launch_encoder(h, (const float*)h->ws_tiles.p, B, (float*)c.latent.p))) return rc;
    HIPCHK(hipMemcpyAsync(dst.as<float>(), c.latent.p, nz, hipMemcpyDeviceToHost, c.s));
    HIPCHK(hipStreamSynchronize(c.s));
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
    HIPCHK(hipMemcpyAsync(c.latent.p, src.as<float>(), nz, hipMemcpyHostToDevice, c.s));
    if ((rc = launch_modulator(h, (const float*)c.latent.p, B, (float*)c.mods.p))) return rc;
    HIPCHK(hipMemcpyAsync(dst.as<float>(), c.mods.p, nm, hipMemcpyDeviceToHost, c.s));
    HIPCHK(hipStreamSynchronize(c.s));
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

int msiren_recon_shape(msiren_handle h, int32_t height, int32_t width, int32_t* nv, int32_t* nh) {
    if (!h) return fail(MSIREN_E_INVALID, "null handle");
    if (height < 1 || width < 1) return fail(MSIREN_E_INVALID, "bad image size %dx%d", height, width);
    if (nv) *nv = (height + h->I - 1) / h->I;
    if (nh) *nh = (width + h->I - 1) / h->I;
    return 0;
}

int msiren_image_to_patches_dev(msiren_handle h, const float* images_dev, int64_t n, int32_t height, int32_t width, float* patches_dev) {
    int rc = check(h, false);
    if (rc) return rc;
    if (n < 0 || height < 1 || width < 1) return fail(MSIREN_E_INVALID, "bad arguments");
    if (n == 0) return 0;
    const int pad = (h->O - h->I) / 2;
    const int vpad = (h->I - height % h->I) % h->I, hpad = (h->I - width % h->I) % h->I;
    // torch's reflect padding requires pad < dim (F.pad raises otherwise)
    if (pad + vpad >= height || pad + hpad >= width)
        return fail(MSIREN_E_INVALID, "image %dx%d is too small for reflect padding of %d/%d", height, width, pad + vpad, pad + hpad);
    const int nV = (height + vpad) / h->I, nH = (width + hpad) / h->I;
    const int64_t total = n * nV * nH * h->O * h->O;
    hipLaunchKernelGGL(msiren::image_to_patches_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->sc[h->cur].s,
                       images_dev, patches_dev, n, height, width, nV, nH, h->O, h->I, pad);
    HIPCHK(hipGetLastError());
    return 0;
}

int msiren_weighted_fold_dev(msiren_handle h, const float* tiles_dev, int64_t n, int32_t nV, int32_t nH, float* recon_dev) {
    int rc = check(h);
    if (rc) return rc;
    if (n < 0 || nV < 1 || nH < 1) return fail(MSIREN_E_INVALID, "bad arguments");
    if (n == 0) return 0;
    const int64_t total = n * nV * h->I * (int64_t)nH * h->I;
    hipLaunchKernelGGL(msiren::weighted_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->sc[h->cur].s,
                       tiles_dev, h->d_foldw, recon_dev, nullptr, nullptr, n, nV, nH, h->S, h->I, (h->S - h->I) / 2, (int*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

int msiren_black_patch_flags_dev(msiren_handle h, const float* tiles_dev, int64_t n_tiles, int64_t tile_elems, int32_t* flags_dev) {
    int rc = check(h, false);
    if (rc) return rc;
    if (n_tiles < 0 || tile_elems < 1 || tile_elems > (1 << 24) || (n_tiles > 0 && (!tiles_dev || !flags_dev))) return fail(MSIREN_E_INVALID, "bad arguments");
    if (n_tiles == 0) return 0;
    if (n_tiles > 0x7fffffffLL) return fail(MSIREN_E_INVALID, "too many tiles for one call: %lld", (long long)n_tiles);
    hipLaunchKernelGGL(msiren::black_flags_kernel, dim3((unsigned)n_tiles), dim3(256), 0, h->sc[h->cur].s, tiles_dev, flags_dev, (int)tile_elems);
    HIPCHK(hipGetLastError());
    return 0;
}

static int copy_rows(msiren_handle h, const float* src, const int32_t* idx, int64_t n_idx, int64_t row_elems, float* dst, int scatter) {
    if (n_idx == 0) return 0;
    if (n_idx > 0x7fffffffLL) return fail(MSIREN_E_INVALID, "too many rows for one call: %lld", (long long)n_idx);
    hipLaunchKernelGGL(msiren::copy_rows_kernel, dim3((unsigned)n_idx), dim3(256), 0, h->sc[h->cur].s, src, dst, idx, (int)row_elems, scatter);
    HIPCHK(hipGetLastError());
    return 0;
}

int msiren_gather_rows_dev(msiren_handle h, const float* src_dev, const int32_t* idx_dev, int64_t n_idx, int64_t row_elems, float* dst_dev) {
    int rc = check(h, false);
    if (rc) return rc;
    if (n_idx < 0 || row_elems < 1 || row_elems > (1 << 24) || (n_idx > 0 && (!src_dev || !idx_dev || !dst_dev))) return fail(MSIREN_E_INVALID, "bad arguments");
    return copy_rows(h, src_dev, idx_dev, n_idx, row_elems, dst_dev, 0);
}

int msiren_scatter_rows_dev(msiren_handle h, const float* src_dev, const int32_t* idx_dev, int64_t n_idx, int64_t n_rows, int64_t row_elems, float* dst_dev) {
    int rc = check(h, false);
    if (rc) return rc;
    if (n_idx < 0 || n_rows < n_idx || row_elems < 1 || row_elems > (1 << 24) || (n_rows > 0 && !dst_dev) || (n_idx > 0 && (!src_dev || !idx_dev)))
        return fail(MSIREN_E_INVALID, "bad arguments");
    if (n_rows == 0) return 0;
    HIPCHK(hipMemsetAsync(dst_dev, 0, (size_t)n_rows * row_elems * sizeof(float), h->sc[h->cur].s));  // rows no index names stay zeros
    return copy_rows(h, src_dev, idx_dev, n_idx, row_elems, dst_dev, 1);
}

int msiren_patches_to_image_dev(msiren_handle h, const float* tiles_dev, int64_t n, int32_t nV, int32_t nH, float* image_dev) {
    int rc = check(h, false);
    if (rc) return rc;
    if (n < 0 || nV < 1 || nH < 1 || (n > 0 && (!tiles_dev || !image_dev))) return fail(MSIREN_E_INVALID, "bad arguments");
    if (n == 0) return 0;
    const int64_t total = n * nV * h->I * (int64_t)nH * h->I;
    hipLaunchKernelGGL(msiren::weighted_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->sc[h->cur].s,
                       tiles_dev, nullptr, image_dev, nullptr, nullptr, n, nV, nH, h->O, h->I, (h->O - h->I) / 2, (int*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

// filter -> model -> reintegrate -> weighted fold on tiles that are already on the device (CURRENT stream)
// `images_dev` given: `patches` is scratch that image_to_patches fills; null: `patches` are the caller's tiles
static int reconstruct_tiles_on_current_stream(msiren_handle h, const float* images_dev, int32_t height, int32_t width, float* patches_rw, const float* patches_ro,
                                               int64_t n, int32_t nV, int32_t nH, float* recon_dev) {
    int rc;
    const int64_t NP = n * nV * nH;
    if (NP > 0x7fffffffLL) return fail(MSIREN_E_INVALID, "too many patches for one call: %lld", (long long)NP);
    if ((rc = ensure(h, h->sc[h->cur].keep, (size_t)(NP + 64) * sizeof(int)))) return rc;
    if ((rc = ensure(h, h->sc[h->cur].rec, (size_t)NP * h->P * sizeof(float)))) return rc;
    if ((rc = ensure(h, h->sc[h->cur].latent, (size_t)NP * h->Z * sizeof(float)))) return rc;
    if ((rc = ensure(h, h->sc[h->cur].mods, (size_t)h->L * NP * h->H * sizeof(float)))) return rc;
    int* black = (int*)h->sc[h->cur].keep.p;
    float* rec = (float*)h->sc[h->cur].rec.p;
    // The reference compacts the non-black tiles, runs the model on those only, and scatters zeros back
    // (tiling.py:244-303).  Same here, on the device: black flags -> list of kept patches (the "plan") ->
    // encoder / modulator / trunk over the kept patches only (their count stays on the device) -> the fold
    // looks each patch up through the plan and lets black ones contribute zeros.
    if ((rc = ensure(h, h->sc[h->cur].plan, (size_t)(2 + 2 * NP) * sizeof(int)))) return rc;
    int* plan = (int*)h->sc[h->cur].plan.p;
    hipStream_t st = h->sc[h->cur].s;
    const float* patches = images_dev ? patches_rw : patches_ro;
    const int pad = (h->O - h->I) / 2;
    // Round 5, synchronous host calls: tiling + flags + plan as ONE launch and the pass counter's reset inside the fold: 10 stream operations
    // per slice -> 7.  The host enqueues into an idle stream there, so every launch saved is ~3 us (370 against 379 us per slice, 263 against
    // 272 masked); back-to-back asynchronous calls run from a full queue and lose 0.6-1.5 % to the fused kernel's 400 device-wide fences, so
    // they keep the separate kernels (profiles/r5/13_*).  Same bits either way (the flag is summed in the same order).
    const bool fused = h->solo && (images_dev || patches_rw);
    if (fused) {
        if ((rc = ensure_queue(h))) return rc;
        msiren::TilingPlanParams tp{images_dev, patches_rw, black, plan, (unsigned*)h->sc[h->cur].queue.p + 32, (int)n, height, width, nV, nH, h->O, h->I, pad, (int)NP, (h->P + 31) / 32};
        hipLaunchKernelGGL(msiren::patches_flags_plan_kernel, dim3((unsigned)NP), dim3(256), 0, st, tp);
        HIPCHK(hipGetLastError());
    } else {
        if (images_dev && (rc = msiren_image_to_patches_dev(h, images_dev, n, height, width, patches_rw))) return rc;
        hipLaunchKernelGGL(msiren::black_flags_kernel, dim3((unsigned)NP), dim3(256), 0, st, patches, black, h->O * h->O);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(msiren::compact_flags_kernel, dim3(1), dim3(256), 0, st, black, (int)NP, (h->P + 31) / 32, plan);
        HIPCHK(hipGetLastError());
    }
    h->plan = plan;
    rc = launch_encoder_modulator(h, patches, NP, (float*)h->sc[h->cur].latent.p, (float*)h->sc[h->cur].mods.p);
    if (!rc) rc = launch_trunk(h, (const float*)h->sc[h->cur].mods.p, NP, rec);
    h->plan = nullptr;
    if (rc) return rc;
    if ((rc = queue_reset_after_plan_launch(h, fused))) return rc;
    const int64_t total = n * nV * h->I * (int64_t)nH * h->I;
    hipLaunchKernelGGL(msiren::weighted_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       rec, h->d_foldw, recon_dev, black, plan + 2 + NP, n, nV, nH, h->S, h->I, (h->S - h->I) / 2,
                       fused && h->sc[h->cur].queue.p ? (int*)h->sc[h->cur].queue.p : nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

// slice pipeline on the CURRENT stream (the host-pointer entry point enqueues its copies around it)
static int reconstruct_on_current_stream(msiren_handle h, const float* images_dev, int64_t n, int32_t height, int32_t width, float* recon_dev) {
    int rc;
    if (n < 0 || (n > 0 && (!images_dev || !recon_dev))) return fail(MSIREN_E_INVALID, "bad arguments");
    if (h->O != 32) return fail(MSIREN_E_INVALID, "the custom encoder is hard-wired to 32x32 tiles, outer_patch_size=%d", h->O);
    if (n == 0) return 0;
    int32_t nV, nH;
    if ((rc = msiren_recon_shape(h, height, width, &nV, &nH))) return rc;
    const int64_t NP = n * nV * nH;
    const int padr = (h->O - h->I) / 2;
    const int vpad = (h->I - height % h->I) % h->I, hpad = (h->I - width % h->I) % h->I;
    // torch's reflect padding requires pad < dim (F.pad raises otherwise): the rule of msiren_image_to_patches_dev
    if (padr + vpad >= height || padr + hpad >= width)
        return fail(MSIREN_E_INVALID, "image %dx%d is too small for reflect padding of %d/%d", height, width, padr + vpad, padr + hpad);
    if ((rc = ensure(h, h->sc[h->cur].patches, (size_t)NP * h->O * h->O * sizeof(float)))) return rc;
    float* patches = (float*)h->sc[h->cur].patches.p;
    return reconstruct_tiles_on_current_stream(h, images_dev, height, width, patches, nullptr, n, nV, nH, recon_dev);
}

int msiren_reconstruct_tiles_dev(msiren_handle h, const float* tiles_dev, int64_t n, int32_t nV, int32_t nH, float* recon_dev) {
    int rc = check(h);
    if (rc) return rc;
    next_stream(h);
    if (n < 0 || nV < 1 || nH < 1 || (n > 0 && (!tiles_dev || !recon_dev))) return fail(MSIREN_E_INVALID, "bad arguments");
    if (h->O != 32) return fail(MSIREN_E_INVALID, "the custom encoder is hard-wired to 32x32 tiles, outer_patch_size=%d", h->O);
    if (n == 0) return 0;
    return reconstruct_tiles_on_current_stream(h, nullptr, 0, 0, nullptr, tiles_dev, n, nV, nH, recon_dev);
}

int msiren_reconstruct_slices_dev(msiren_handle h, const float* images_dev, int64_t n, int32_t height, int32_t width, float* recon_dev) {
    int rc = check(h);
    if (rc) return rc;
    next_stream(h);
    return reconstruct_on_current_stream(h, images_dev, n, height, width, recon_dev);
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
    HIPCHK(hipMemcpyAsync(dst_dev, src.as<void>(), bytes, hipMemcpyHostToDevice, h->sc[h->cur].s));
    HIPCHK(hipStreamSynchronize(h->sc[h->cur].s));
    return 0;
}

int msiren_memcpy_d2h(msiren_handle h, void* dst_host, const void* src_dev, size_t bytes) {
    int rc = check(h, false);
    if (rc) return rc;
    if (bytes == 0) return 0;
    if ((rc = sync_all(h))) return rc;
    const HostDst dst(dst_host, bytes);
    HOSTBUF_OK(dst);
    HIPCHK(hipMemcpyAsync(dst.as<void>(), src_dev, bytes, hipMemcpyDeviceToHost, h->sc[h->cur].s));
    HIPCHK(hipStreamSynchronize(h->sc[h->cur].s));
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

int msiren_trunk_timeline(msiren_handle h, const float* mods_dev, int64_t B, float* out_dev, uint64_t* stamps_host) {
    int rc = check(h);
    if (rc) return rc;
    if (h->HP != 256 || h->cfg.activation != MSIREN_ACT_SINE || h->cfg.residual || h->H != 256)
        return fail(MSIREN_E_INVALID, "the timeline diagnostic is built for H=256, sine, non-residual only");
    if (B <= 0 || !mods_dev || !out_dev || !stamps_host) return fail(MSIREN_E_INVALID, "bad arguments");
    const int chunks = (h->P + 63) / 64;
    const int grid = (int)(B * chunks);
    DevBuf st;
    if ((rc = ensure(h, st, (size_t)grid * 32 * sizeof(uint64_t)))) return rc;
    HIPCHK(hipMemsetAsync(st.p, 0, (size_t)grid * 32 * sizeof(uint64_t), h->sc[h->cur].s));
    msiren::TrunkParams p = make_trunk_params(h, mods_dev, h->H, B, out_dev);
    p.stamps = (unsigned long long*)st.p;
    hipLaunchKernelGGL((msiren::siren_trunk_f32_kernel<256, 0, 0, 1>), dim3(grid), dim3(256), 256 * 256 + 256 * 16, h->sc[h->cur].s, p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(stamps_host, st.p, (size_t)grid * 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->sc[h->cur].s));
    HIPCHK(hipStreamSynchronize(h->sc[h->cur].s));
    HIPCHK(hipFree(st.p));
    return 0;
}

int msiren_f16x3_timeline(msiren_handle h, const float* mods_dev, int64_t B, float* out_dev, uint64_t* stamps_host) {
    int rc = check(h);
    if (rc) return rc;
    if (!h->f16x3_ready || h->cfg.activation != MSIREN_ACT_SINE) return fail(MSIREN_E_INVALID, "f16x3 timeline: H=256 sine model required");
    msiren::TrunkF16Params p{};
    p.grid = h->d_grid; p.l0 = h->d_l0; p.s0t = h->d_s0t; p.wp = (const _Float16*)h->d_wp16n; p.bias = h->d_bias16;
    p.wout = h->d_wout16; p.mods = mods_dev; p.out = out_dev;
    for (int i = 0; i < 16; ++i) p.winv[i] = h->mscale16[i];
    p.bout = h->bout; p.cg0 = h->cg0; p.cg = h->cg; p.B = (int)B; p.P = h->P; p.L = h->L;
    p.units_per_patch = (h->P + 31) / 32;
    p.total_units = (int)(B * p.units_per_patch);
    const int grid = (int)std::min<int64_t>(h->num_cus, (p.total_units + 3) / 4);
    DevBuf st, q;
    if ((rc = ensure(h, st, (size_t)grid * 4 * 48 * sizeof(uint64_t))) || (rc = ensure(h, q, 256))) return rc;
    hipStream_t s = h->sc[h->cur].s;
    HIPCHK(hipMemsetAsync(st.p, 0, (size_t)grid * 4 * 48 * sizeof(uint64_t), s));
    p.pass_counter = (int*)q.p;
    HIPCHK(hipMemsetAsync(p.pass_counter, 0, 4, s));
    p.pass_base = 0;
    p.stamps = (unsigned long long*)st.p;
    const int lds = msiren::F16Lds<4>::total(h->L);
    {
        if (h->L != 5) return fail(MSIREN_E_INVALID, "f16x3 timeline: the stamped build is the num_layers = 5 instance");
        auto k = msiren::siren_trunk_f16x3n_kernel<0, 4, 5, 1>;
        HIPCHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, p);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(stamps_host, st.p, (size_t)grid * 4 * 48 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(st.p));
    HIPCHK(hipFree(q.p));
    return 0;
}

int msiren_f16x3w_timeline(msiren_handle h, const float* mods_dev, int64_t B, float* out_dev, uint64_t* stamps_host) {
    int rc = check(h);
    if (rc) return rc;
    if (!h->f16x3_ready || h->cfg.activation != MSIREN_ACT_SINE || h->L < msiren::WS_MIN_L || h->L > msiren::WS_MAX_L)
        return fail(MSIREN_E_INVALID, "f16x3w timeline: H=256 sine model with %d <= num_layers <= %d required", msiren::WS_MIN_L, msiren::WS_MAX_L);
    if (B <= 0 || !mods_dev || !out_dev || !stamps_host) return fail(MSIREN_E_INVALID, "bad arguments");
    msiren::TrunkWsParams p{};
    if (!h->d_dump) HIPCHK(hipMalloc((void**)&h->d_dump, 256 * sizeof(float)));
    p.dump = h->d_dump;
    p.s0t = h->d_s0t; p.wp = (const _Float16*)h->d_wp16n; p.bias = h->d_bias16; p.wout = h->d_wout16; p.mods = mods_dev; p.out = out_dev;
    for (int i = 0; i < 16; ++i) p.mscale[i] = h->mscale16[i];
    p.bout = h->bout; p.cg0 = h->cg0; p.cg = h->cg; p.B = (int)B; p.P = h->P; p.L = h->L;
    const int upp = (h->P + 31) / 32;
    p.units_per_patch = upp;
    p.total_units = (int)(B * upp);
    int lg = 0;
    while ((1 << lg) < upp) ++lg;
    p.div_k = 30 + lg;
    p.div_m = (unsigned)(((1ULL << p.div_k) + (unsigned)upp - 1) / (unsigned)upp);
    const int grid = (int)std::min<int64_t>(h->num_cus, ((int64_t)p.total_units + 1) / 2);
    DevBuf st, q;
    struct Free {  // whichever way the function is left
        DevBuf &a, &b;
        ~Free() { if (a.p) (void)hipFree(a.p); if (b.p) (void)hipFree(b.p); }
    } free_on_exit{st, q};
    const size_t nst = (size_t)grid * 96 * 8 * sizeof(uint64_t);
    if ((rc = ensure(h, st, nst)) || (rc = ensure(h, q, 256))) return rc;
    hipStream_t s = h->sc[h->cur].s;
    HIPCHK(hipMemsetAsync(st.p, 0, nst, s));
    p.pass_counter = (int*)q.p;
    HIPCHK(hipMemsetAsync(p.pass_counter, 0, 4, s));
    p.pass_base = 0;
    p.stamps = (unsigned long long*)st.p;
    const int lds = msiren::WsLds<4>::total(h->L);
    auto k = msiren::siren_trunk_f16x3w_kernel<0, 4, 1>;
    HIPCHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(stamps_host, st.p, nst, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

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

int msiren_mfma_sustained_probe(msiren_handle h, double* tflops, double* mhz_equivalent) {
    int rc = check(h, false);
    if (rc) return rc;
    if (!tflops) return fail(MSIREN_E_INVALID, "null argument");
    // operands with the trunk's magnitudes: weights 0.1 rms (hi) and 2^-11 of that (lo); activations in [-1.5, 1.5] and 2^-11 of that
    const size_t n = (size_t)8 * 12 * 512;
    std::vector<_Float16> host(n);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xffff) / 32768.f - 1.f; };  // [-1, 1)
    for (size_t g = 0; g < 8; ++g)
        for (int kind = 0; kind < 12; ++kind)
            for (int e = 0; e < 512; ++e) {
                const float u = rnd();
                float v;
                if (kind < 4) v = 0.17f * u;                       // W hi
                else if (kind < 8) v = 0.17f * u * (1.f / 2048.f);  // W lo
                else if (kind < 10) v = 1.5f * u;                   // x hi
                else v = 1.5f * u * (1.f / 2048.f);                 // x lo
                host[(g * 12 + kind) * 512 + e] = (_Float16)v;
            }
    DevBuf src, sink;
    if ((rc = ensure(h, src, n * sizeof(_Float16))) || (rc = ensure(h, sink, 1024))) return rc;
    hipStream_t s = h->sc[h->cur].s;
    HIPCHK(hipMemcpyAsync(src.p, host.data(), n * sizeof(_Float16), hipMemcpyHostToDevice, s));
    const int iters = 40000;  // x 24 MFMAs x 16 cycles = 15.4 M cycles: ~8 ms, long enough for the clock to settle
    const int grid = h->num_cus;
    hipLaunchKernelGGL(msiren::mfma_sustained_probe_kernel, dim3(grid), dim3(256), 0, s, (const _Float16*)src.p, (float*)sink.p, iters / 8);  // warm
    HIPCHK(hipEventRecord(h->ev0, s));
    hipLaunchKernelGGL(msiren::mfma_sustained_probe_kernel, dim3(grid), dim3(256), 0, s, (const _Float16*)src.p, (float*)sink.p, iters);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, s));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    const double flops = (double)grid * 4 * (double)iters * 24 * (16.0 * 16 * 32 * 2);
    *tflops = flops / (ms * 1e-3) * 1e-12;
    if (mhz_equivalent) *mhz_equivalent = (double)iters * 24 * 16.0 / (ms * 1e-3) * 1e-6;  // the clock at which one MFMA per 16 cycles gives this rate
    (void)hipFree(src.p);
    (void)hipFree(sink.p);
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
