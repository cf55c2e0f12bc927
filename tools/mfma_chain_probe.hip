// Probe: cycles per v_mfma_f32_32x32x16_f16 for one dependent accumulation chain vs two interleaved
// chains, VGPR-dst form with the B operand in AGPRs (the f16x3 trunk's exact instruction form).
// hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_chain_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ h8 to_a(h8 v) { h8 r; asm("" : "=a"(r) : "0"(v)); return r; }
template <int CHAINS>
__global__ void k(float* out, int iters) {
    h8 a, b0;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b0[i] = (_Float16)(0.02f * i); }
    h8 b = to_a(b0);
    f16v c0 = {0}, c1 = {0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (CHAINS == 1 || (j & 1) == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            else c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = (float)(t1 - t0) / (8.0f * iters); out[1] = c0[0] + c1[1]; }
}
int main() {
    float* d; (void)hipMalloc(&d, 64); float h[4];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, 20000);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("1 chain : %.2f cycles per MFMA (256 WGs x 4 waves)\n", h[0]);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, 20000);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("2 chains: %.2f cycles per MFMA\n", h[0]);
    }
    return 0;
}
