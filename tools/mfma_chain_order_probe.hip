// Probe (round 3): does the ORDER in which a k-step's MFMAs visit the accumulators change what the chip sustains under its power
// limit?  v_mfma_f32_16x16x32_f16, one wave per SIMD on every CU, 8 accumulators, 192 MFMAs per "slot", operands of the
// trunk's magnitudes (pseudo-random; weights hi/lo, activations hi/lo).
//   order 0: round-robin -- consecutive MFMAs write different accumulators (the trunk's order: 8 accumulators x 3 products per k-step)
//   order 1: chains -- the 24 MFMAs of one accumulator back to back, then the next accumulator
//   order 2: pairs of chains -- two accumulators alternate through their 24 MFMAs each
// hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_order_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int ORDER>
__global__ __launch_bounds__(256, 1) void k(const _Float16* __restrict__ src, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    h8 wh[8], wl[8], xh[2], xl[2];   // 8 k-steps' worth of one tile's weights would be 16 fragments; 8 + 8 keeps the operand variety
    const _Float16* base = src + lane * 8;
    for (int q = 0; q < 8; ++q) { wh[q] = *(const h8*)(base + q * 512); wl[q] = *(const h8*)(base + (8 + q) * 512); }
    for (int q = 0; q < 2; ++q) { xh[q] = *(const h8*)(base + (16 + q) * 512); xl[q] = *(const h8*)(base + (18 + q) * 512); }
    f4 acc[8];
    for (int a = 0; a < 8; ++a) acc[a] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        if (ORDER == 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[(s + a) & 7], xh[a & 1], acc[a], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + a) & 7], xl[a & 1], acc[a], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + a) & 7], xh[a & 1], acc[a], 0, 0, 0);
            }
        } else if (ORDER == 1) {
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[(s + a) & 7], xh[a & 1], acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + a) & 7], xl[a & 1], acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + a) & 7], xh[a & 1], acc[a], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int a = 0; a < 8; a += 2)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[(s + a) & 7], xh[0], acc[a], 0, 0, 0);
                    acc[a + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[(s + a + 1) & 7], xh[1], acc[a + 1], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + a) & 7], xl[0], acc[a], 0, 0, 0);
                    acc[a + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + a + 1) & 7], xl[1], acc[a + 1], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + a) & 7], xh[0], acc[a], 0, 0, 0);
                    acc[a + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + a + 1) & 7], xh[1], acc[a + 1], 0, 0, 0);
                }
        }
        if ((i & 7) == 7)
            for (int a = 0; a < 8; ++a) acc[a] *= 0.5f;
    }
    float r = 0.f;
    for (int a = 0; a < 8; ++a) r += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    if (r == 12345.678f) sink[threadIdx.x] = r;
}

int main() {
    const size_t n = 20 * 512;
    std::vector<_Float16> host(n);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xffff) / 32768.f - 1.f; };
    for (int kind = 0; kind < 20; ++kind)
        for (int e = 0; e < 512; ++e) {
            const float u = rnd();
            host[kind * 512 + e] = (_Float16)(kind < 8 ? 0.17f * u : kind < 16 ? 0.17f * u / 2048.f : kind < 18 ? 1.5f * u : 1.5f * u / 2048.f);
        }
    _Float16* d; float* sink;
    (void)hipMalloc(&d, n * 2); (void)hipMalloc(&sink, 1024);
    (void)hipMemcpy(d, host.data(), n * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 5000;
    const char* names[3] = {"round-robin over 8 accumulators (trunk)", "chains of 24 on one accumulator", "two accumulators alternating"};
    for (int rep = 0; rep < 3; ++rep)
        for (int o = 0; o < 3; ++o) {
            auto launch = [&](int it) {
                if (o == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, sink, it);
                else if (o == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, sink, it);
                else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, sink, it);
            };
            launch(iters / 8);
            (void)hipEventRecord(e0); launch(iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double tf = 256.0 * 4 * 192.0 * iters * 16384.0 / (ms * 1e-3) * 1e-12;
            printf("%-42s %.3f ms  %.0f TFLOP/s fp16 chip-wide  (= %.0f MHz at 16 cycles per MFMA)\n", names[o], ms, tf, 192.0 * iters * 16 / (ms * 1e-3) * 1e-6);
        }
    return 0;
}
