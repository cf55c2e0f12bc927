// Probe (gfx950): does v_mfma_f32_32x32x16_f16 keep f16 subnormal INPUTS, and what do the f32->f16
// conversions produce for subnormal results?  hipcc --offload-arch=gfx950 -O2 tools/f16_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val, float b_val) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0];
        typedef __fp16 fp2 __attribute__((ext_vector_type(2)));
        fp2 p = __builtin_amdgcn_cvt_pkrtz(a_val, a_val * 3.0f);
        out[1] = (float)p[0]; out[2] = (float)p[1];
        out[3] = (float)(_Float16)a_val;
    }
}
__global__ void timing(float* out, int iters) {
    h8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + threadIdx.x * 1e-3f); b[i] = (_Float16)(0.25f + i * 1e-2f); }
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = (float)(t1 - t0) / (4.0f * iters); out[1] = c0[0] + c1[1] + c2[2] + c3[3]; }
}
int main() {
    float* d; hipMalloc(&d, 64); float h[8];
    const float vals[3] = {9.5367431640625e-07f /*2^-20*/, 3.0517578125e-05f /*2^-15*/, 1.0f};
    for (float v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v, 1.0f);
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("a=%g: mfma sum=%g (expected %g)  cvt_pkrtz=(%g,%g) cvt_rne=%g\n", v, h[0], 16.0 * v, h[1], h[2], h[3]);
    }
    hipLaunchKernelGGL(timing, dim3(1), dim3(64), 0, 0, d, 10000);
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("v_mfma_f32_32x32x16_f16: %.1f memtime ticks per MFMA (one wave, 4 accumulators)\n", h[0]);
    return 0;
}
