// libmsiren.so, host side: which kernel runs for which shape, and its launch (DESIGN.md section 4 is this file as a table).
//     [tiling] -> encoder -> modulator -> fused SIREN trunk -> [weighted fold]
// Nothing here falls back to the CPU: every function launches HIP kernels on the handle's current stream or fails.
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "host_ctx.h"
#include "encoder_modulator.hip.h"
#include "encoder_modulator_f16x3.hip.h"
#include "siren_trunk_f16x3n.hip.h"
#include "siren_trunk_f16x3h.hip.h"
#include "siren_trunk_f16x3w.hip.h"
#include "siren_trunk_f32.hip.h"
#include "siren_trunk_x1n.hip.h"
#include "siren_trunk_x1w.hip.h"
#include "tiling.hip.h"
#include "trunk_instances.h"  // the trunk / prologue kernels are compiled in their own translation units (k_*.hip)
#include "launch_dispatch.h"

namespace mh {

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
int launch_trunk_f32_cond(msiren_ctx* h, const float* mods_dev, int64_t B, float* out_dev, const int* flag_word, unsigned flag_val) {
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
    // Shapes the MFMA Linear kernels do not take (latent_dim not a multiple of 16): one fused per-tile kernel.  (Until round 6 batches
    // below 48 tiles took it as well, for two launches less -- but its VALU sums run in another order than the MFMA kernels', so an
    // fp32 handle's latent depended in the last bits on the size of the batch a tile came in; the split-fp16 prologue never had that seam.)
    if (h->Z % 16 != 0) {
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

}  // namespace mh

using namespace mh;

// ---- tiling steps and the device-resident slice pipeline (include/msiren.h) -------------------------------------------------
extern "C" {

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

}  // extern "C"

namespace mh {

// filter -> model -> reintegrate -> weighted fold on tiles that are already on the device (CURRENT stream)
// `images_dev` given: `patches` is scratch that image_to_patches fills; null: `patches` are the caller's tiles
int reconstruct_tiles_on_current_stream(msiren_handle h, const float* images_dev, int32_t height, int32_t width, float* patches_rw, const float* patches_ro,
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
int reconstruct_on_current_stream(msiren_handle h, const float* images_dev, int64_t n, int32_t height, int32_t width, float* recon_dev) {
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

}  // namespace mh

extern "C" {

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

}  // extern "C"

