// libmsiren.so, host side: diagnostics that never run on the forward path -- stamped builds of the trunks (per-phase timelines) and
// the sustained-MFMA probe bench.py reports beside the roofline.
#include <algorithm>
#include <vector>

#include "host_ctx.h"
#include "mfma_probe.hip.h"
#include "siren_trunk_f16x3n.hip.h"
#include "siren_trunk_f16x3w.hip.h"
#include "siren_trunk_f32.hip.h"
#include "trunk_instances.h"
#include "launch_dispatch.h"

using namespace mh;

extern "C" {

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

}  // extern "C"
