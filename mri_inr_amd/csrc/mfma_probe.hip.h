// Diagnostic (msiren_mfma_sustained_probe): what the chip sustains on NOTHING but the split-fp16 trunk's MFMA stream --
// v_mfma_f32_16x16x32_f16, one wave per SIMD on every CU, eight accumulators, the trunk's three products per k-step
// (W_lo x_hi, W_hi x_lo, W_hi x_hi) on operands with the trunk's magnitudes (weights ~0.1 rms, activations x modulation ~1,
// the lo halves 2^-11 of them).  The nominal 2.5 PFLOP/s is reached only by operands that toggle few bits
// (tools/mfma_operand_probe.hip: 2.39-2.46); on real data the board's power limit sets the clock.  bench.py reports the
// figure beside the roofline (never as `peak`).
#pragma once
#include <hip/hip_runtime.h>

namespace msiren {

typedef _Float16 probe_h8 __attribute__((ext_vector_type(8)));
typedef float probe_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void mfma_sustained_probe_kernel(const _Float16* __restrict__ src, float* __restrict__ sink, int iters) {
    // 4 weight fragments (hi, lo) and 2 activation fragments (hi, lo) per lane, from a host-filled buffer of 8 x 64 lanes x 8 halves per kind
    probe_h8 wh[4], wl[4], xh[2], xl[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const _Float16* base = src + (size_t)((blockIdx.x * 4 + wave) % 8) * 64 * 8 * 12 + lane * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wh[q] = *reinterpret_cast<const probe_h8*>(base + (size_t)q * 512);
        wl[q] = *reinterpret_cast<const probe_h8*>(base + (size_t)(4 + q) * 512);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        xh[q] = *reinterpret_cast<const probe_h8*>(base + (size_t)(8 + q) * 512);
        xl[q] = *reinterpret_cast<const probe_h8*>(base + (size_t)(10 + q) * 512);
    }
    probe_f4 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = probe_f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[a >> 1], xh[a & 1], acc[a], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[a >> 1], xl[a & 1], acc[a], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[a >> 1], xh[a & 1], acc[a], 0, 0, 0);
        // keep the sums bounded (and the operands honest) without touching the MFMA stream's density: once per 64 k-steps
        if ((i & 63) == 63) {
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] *= 0.5f;
        }
    }
    float r = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) r += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    if (r == 12345.678f) sink[threadIdx.x] = r;  // never true: keeps the chains alive
}

}  // namespace msiren
