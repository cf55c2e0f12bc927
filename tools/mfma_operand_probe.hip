// Probe (round 3): cycles per v_mfma_f32_16x16x32_f16 in the weight-stationary trunk's operand pattern -- one wave per SIMD,
// 8 accumulators v[192:223], 24 MFMAs per k-step (8 accumulators x 3 products) -- with the A operand taken from
//   0: AGPRs, 64 different fragments over 8 k-steps (the trunk's form)      1: AGPRs, the same 2 fragments every time
//   2: VGPRs (8 fragments v[128:159])                                        3: as 0 with the accumulators in AGPRs a[192:223], A in VGPRs
// hipcc --offload-arch=gfx950 -O3 tools/mfma_operand_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(VD, A, B) asm volatile("v_mfma_f32_16x16x32_f16 v[%0:%1], a[%2:%3], %4, v[%0:%1]" ::"n"(VD), "n"((VD) + 3), "n"(A), "n"((A) + 3), "v"(B))
#define MFV(VD, A, B) asm volatile("v_mfma_f32_16x16x32_f16 v[%0:%1], v[%2:%3], %4, v[%0:%1]" ::"n"(VD), "n"((VD) + 3), "n"(A), "n"((A) + 3), "v"(B))
#define MFA(VD, A, B) asm volatile("v_mfma_f32_16x16x32_f16 a[%0:%1], v[%2:%3], %4, a[%0:%1]" ::"n"(VD), "n"((VD) + 3), "n"(A), "n"((A) + 3), "v"(B))

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int S>
__device__ __forceinline__ void kstep(f32x4 bh0, f32x4 bl0, f32x4 bh1, f32x4 bl1) {
#define ONE(T, G, BH, BL)                                                                              \
    do {                                                                                               \
        constexpr int VD = 192 + 8 * (T) + 4 * (G);                                                    \
        constexpr int AL = MODE == 0 ? 32 * S + 8 * (T) + 4 : (MODE == 1 ? 4 : 128 + 8 * (T) + 4);    \
        constexpr int AH = MODE == 0 ? 32 * S + 8 * (T) : (MODE == 1 ? 0 : 128 + 8 * (T));            \
        if constexpr (MODE <= 1) { MF(VD, AL, BH); MF(VD, AH, BL); MF(VD, AH, BH); }                   \
        else if constexpr (MODE == 2) { MFV(VD, AL, BH); MFV(VD, AH, BL); MFV(VD, AH, BH); }           \
        else { MFA(VD, AL, BH); MFA(VD, AH, BL); MFA(VD, AH, BH); }                                    \
    } while (0)
    ONE(0, 0, bh0, bl0); ONE(1, 0, bh0, bl0); ONE(2, 0, bh0, bl0); ONE(3, 0, bh0, bl0);
    ONE(0, 1, bh1, bl1); ONE(1, 1, bh1, bl1); ONE(2, 1, bh1, bl1); ONE(3, 1, bh1, bl1);
#undef ONE
}

template <int MODE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(120))) void k(float* out, int iters) {
    asm volatile("" ::: "v255", "a255");
    f32x4 bh0, bl0, bh1, bl1;
    for (int i = 0; i < 4; ++i) { bh0[i] = 1e-3f * threadIdx.x; bl0[i] = 2e-3f; bh1[i] = 3e-3f; bl1[i] = 1e-3f * i; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        kstep<MODE, 0>(bh0, bl0, bh1, bl1); kstep<MODE, 1>(bh0, bl0, bh1, bl1); kstep<MODE, 2>(bh0, bl0, bh1, bl1); kstep<MODE, 3>(bh0, bl0, bh1, bl1);
        kstep<MODE, 4>(bh0, bl0, bh1, bl1); kstep<MODE, 5>(bh0, bl0, bh1, bl1); kstep<MODE, 6>(bh0, bl0, bh1, bl1); kstep<MODE, 7>(bh0, bl0, bh1, bl1);
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE] = (float)(t1 - t0) / (192.0f * iters);
}

int main() {
    float* d; (void)hipMalloc(&d, 64); (void)hipMemset(d, 0, 64); float h[4];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        float ms[4];
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[0], e0, e1);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[1], e0, e1);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[2], e0, e1);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[3], e0, e1);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const char* names[4] = {"A in AGPRs, 64 fragments (trunk form)", "A in AGPRs, 2 fragments", "A in VGPRs", "accumulators in AGPRs, A in VGPRs"};
        for (int m = 0; m < 4; ++m) {
            const double tf = 256.0 * 4 * 192.0 * iters * 16 * 16 * 32 * 2 / (ms[m] * 1e-3) * 1e-12;
            printf("%-40s %.2f cycles per MFMA; %.3f ms; %.0f TFLOP/s fp16 issued, chip-wide\n", names[m], h[m], ms[m], tf);
        }
    }
    return 0;
}
