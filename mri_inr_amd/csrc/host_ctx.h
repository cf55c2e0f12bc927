// Shared by the host translation units of libmsiren.so (round 6: the former 3 000-line msiren.hip, cut by responsibility):
//     msiren.hip           C ABI: lifecycle, weights, the forward / slice entry points on host pointers, memory, timing, info
//     launch_dispatch.hip  which kernel runs for which shape, and its launch: trunks, prologue, tiling / fold, the *_dev slice pipeline
//     weights_pack.hip     state_dict -> the kernels' weight layouts (host arithmetic + uploads)
//     host_buffers.hip     what a caller's host range is (pageable / page-locked / page-locked in part), bounce buffers
//     comm_rccl.hip        RCCL through dlopen: communicator, the one weight broadcast, barrier / MAX
//     diagnostics.hip      stamped timeline builds, the sustained-MFMA probe
// Everything in namespace mh is internal (the library is built with -fvisibility=hidden; only include/msiren.h is exported).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/msiren.h"
#include "encoder_params.h"
#include "pass_queue.h"

typedef struct ncclComm* ncclComm_t;  // (<rccl/rccl.h>'s own typedef: the handle only stores the pointer)

namespace mh {

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const char* last_error();

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return ::mh::fail(MSIREN_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace mh

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
        mh::DevBuf mods, modpad, latent, patches, keep, rec, queue, feat, plan;
        mh::DevBuf cscratch;  // split-fp16 Modulator: the latent part of layers 1.., lane-private (encoder_modulator_f16x3.hip.h)
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
    mh::DevBuf ws_out, ws_tiles, ws_in, ws_img;  // staging of the host-pointer entry points
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
    mh::DevBuf ws_comm;
};

namespace mh {

// msiren.hip
int use_device(msiren_ctx* h);
int ensure(msiren_ctx* h, DevBuf& b, size_t bytes);                 // grow-only device workspace
int upload(float** dst, const std::vector<float>& src);            // (re)allocate + blocking H2D copy
int check(msiren_ctx* h, bool need_commit = true);
int sync_all(msiren_ctx* h);
bool take_range_flag(msiren_ctx* h);
void next_stream(msiren_ctx* h);   // asynchronous forward entry points rotate over the configured streams

// weights_pack.hip: state_dict -> kernel layouts.  pack_modulator / pack_encoder return 1 when their keys are absent.
void declare_expected(msiren_ctx* h);
int pack_trunk(msiren_ctx* h);
int pack_trunk_f16x3(msiren_ctx* h);
int pack_trunk_x1(msiren_ctx* h);
int pack_modulator(msiren_ctx* h);
int pack_encoder(msiren_ctx* h);
int pack_prologue_f16x3(msiren_ctx* h);
int pack_fold_weights(msiren_ctx* h);

// launch_dispatch.hip: everything below enqueues on h->sc[h->cur].s
bool use_f16x3(msiren_ctx* h);
bool ws_capable(msiren_ctx* h, int64_t B);
int ensure_queue(msiren_ctx* h);
int launch_trunk(msiren_ctx* h, const float* mods_dev, int64_t B, float* out_dev);
int launch_trunk_f32_cond(msiren_ctx* h, const float* mods_dev, int64_t B, float* out_dev, const int* flag_word = nullptr, unsigned flag_val = 0);
int launch_modulator(msiren_ctx* h, const float* z_dev, int64_t B, float* mods_dev);
int launch_encoder(msiren_ctx* h, const float* tiles_dev, int64_t B, float* z_dev);
int forward_latent_dev(msiren_ctx* h, const float* z_dev, int64_t B, float* out_dev, float* mods_out_dev);
int forward_tiles_dev(msiren_ctx* h, const float* tiles_dev, int64_t B, float* out_dev);
int reconstruct_on_current_stream(msiren_ctx* h, const float* images_dev, int64_t n, int32_t height, int32_t width, float* recon_dev);

// comm_rccl.hip
int comm_destroy(msiren_ctx* h);

}  // namespace mh
