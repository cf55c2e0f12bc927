// Probe: sustained full-chip rate of v_mfma_f32_32x32x16_f16 (the f16x3 trunk's instruction) with
// pseudo-random operands, alone and mixed with the trunk's other per-MFMA work (LDS fragment reads,
// VALU+v_sin).  Shows what the 2.5 PFLOP/s nominal peak becomes under the board's power limit.
// hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_peak_probe.hip -o /tmp/peak && /tmp/peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// MODE 0: MFMA only (operands in registers).  MODE 1: + one 16-byte LDS fragment read per lane per
// 1.5 MFMAs (2 reads per 3 MFMAs, the trunk's ratio).  MODE 2: MODE 1 + ~1/3 element of epilogue
// VALU per MFMA (sin, fma, mul, 2 cvt).
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const _Float16* src, int iters, int wgs_active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if ((int)blockIdx.x >= wgs_active) return;
    h8 a[4], b[4];
    for (int q = 0; q < 4; ++q) {
        a[q] = *reinterpret_cast<const h8*>(src + ((threadIdx.x * 4 + q) * 8) % 8192);
        b[q] = *reinterpret_cast<const h8*>(src + ((threadIdx.x * 4 + q) * 8 + 4096) % 8192);
    }
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<_Float16*>(lds)[i] = src[i];
    __syncthreads();
    f16v c0 = {0}, c1 = {0};
    float v = 0.001f * threadIdx.x, acc = 0.f;
    const unsigned char* lp = lds + (threadIdx.x & 63) * 16;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (MODE >= 1 && (j % 3) != 2) a[j & 3] = *reinterpret_cast<const h8*>(lp + ((i * 12 + j) & 15) * 1024);
            if (j & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[j & 3], c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[j & 3], c0, 0, 0, 0);
            if (MODE >= 2 && (j % 3) == 0) {
                float s = __builtin_amdgcn_sinf(__builtin_fmaf(v, 0.37f, acc));
                s *= 1.01f;
                _Float16 hi = (_Float16)s;
                _Float16 lo = (_Float16)(s - (float)hi);
                acc = __builtin_fmaf(s, (float)lo, acc);
                v += (float)hi;
            }
        }
    }
    float r = acc + v;
    for (int e = 0; e < 16; ++e) r += c0[e] + c1[e];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, float* d_out, _Float16* d_src, int wgs, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 16384, 0, d_out, d_src, iters / 10, wgs);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 16384, 0, d_out, d_src, iters, wgs);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * 12 * 32768.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    // one wave per SIMD: cycles per MFMA = clock * time / count; at 32 cycles each the clock follows
    printf("%-34s %3d CUs: %8.1f TFLOP/s  (%.2f of 2500 x CUs/256; implied clock %.2f GHz if MFMA-bound)\n", name, wgs, tf,
           tf / (2500.0 * wgs / 256.0), (double)iters * 12 * 32 / (ms * 1e-3) / 1e9);
}

int main() {
    float* d_out; _Float16* d_src;
    (void)hipMalloc(&d_out, 4096); (void)hipMalloc(&d_src, 8192 * 2);
    _Float16 hsrc[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) hsrc[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
    (void)hipMemcpy(d_src, hsrc, sizeof hsrc, hipMemcpyHostToDevice);
    const int iters = 40000;  // ~8 ms per launch at full rate
    for (int wgs : {256, 128, 64}) {
        run<0>("MFMA only", d_out, d_src, wgs, iters);
        run<1>("MFMA + LDS fragment reads", d_out, d_src, wgs, iters);
        run<2>("MFMA + LDS reads + epilogue VALU", d_out, d_src, wgs, iters);
    }
    return 0;
}
