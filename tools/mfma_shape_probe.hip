// Probe: does the MFMA SHAPE change what the chip sustains under its power limit, in the f16x3 trunk's
// operand pattern?  (MI355X_MICROARCH.md, "DVFS give-back" (7): 16x16x32 delivered 1.12-1.15x the FLOP/s of
// 32x32x16 in power-limited loops at equal cycles per FLOP.)
//
// Both loops do the trunk's per-k-step work on pseudo-random data, one wave per SIMD, software-pipelined
// (the LDS fragment reads of step i+1 are issued before the MFMAs of step i):
//   SHAPE 0: v_mfma_f32_32x32x16_f16 -- per 32-feature x 32-coordinate x 16-k step: 2 fragment reads
//            (W_hi, W_lo: 2 x 1 KB) and 3 MFMAs (W_lo*x_hi, W_hi*x_lo, W_hi*x_hi), 96 cycles.
//   SHAPE 1: v_mfma_f32_16x16x32_f16 -- per 16-feature x 32-coordinate (two 16-column groups) x 32-k step:
//            2 fragment reads and 6 MFMAs (3 products x 2 column groups), 96 cycles.  Same bytes per FLOP.
// MODE 0: operands in registers only; MODE 1: with the LDS fragment reads; MODE 2: MODE 1 + one element of the
// trunk's epilogue per step (fma, v_sin, mul, cvt_pkrtz, residual, cvt_pkrtz, a move: the trunk's ~2 VALU per 32 MFMA cycles).
// hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_shape_probe.hip -o /tmp/shape && /tmp/shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int MODE, int BAGPR = 0>
__global__ void __launch_bounds__(256, 1) k(float* out, const _Float16* src, int iters, int wgs_active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if ((int)blockIdx.x >= wgs_active) return;
    // activation fragments: 16 hi + 16 lo (k = 256), as in the trunk
    h8 xh[16], xl[16];
    for (int q = 0; q < 16; ++q) {
        xh[q] = *reinterpret_cast<const h8*>(src + ((threadIdx.x * 16 + q) * 8) % 32768);
        xl[q] = *reinterpret_cast<const h8*>(src + ((threadIdx.x * 16 + q) * 8 + 16384) % 32768);
    }
    if (BAGPR) {  // B operands in the accumulator half of the register file, as in the trunk
        for (int q = 0; q < 16; ++q) {
            asm("; B -> AGPR" : "=a"(xh[q]) : "0"(xh[q]));
            asm("; B -> AGPR" : "=a"(xl[q]) : "0"(xl[q]));
        }
    }
    for (int i = threadIdx.x; i < 32768; i += 256) reinterpret_cast<_Float16*>(lds)[i] = src[i];  // 64 KB of "weights"
    __syncthreads();
    const unsigned char* lp = lds + (threadIdx.x & 63) * 16;
    h8 wf[2][2];  // [buffer][hi|lo]: the step in flight and the next one
    wf[0][0] = *reinterpret_cast<const h8*>(lp);
    wf[0][1] = *reinterpret_cast<const h8*>(lp + 1024);
    wf[1][0] = wf[0][0];
    wf[1][1] = wf[0][1];
    f32x16 c32 = {0};
    f32x4 c16[2][2] = {};
    float ev = 0.001f * threadIdx.x, eacc = 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1) {  // next step's fragments (64 KB ring, 2 KB per step), a whole step ahead of their use
                const int o = ((i * 16 + s + 1) & 31) * 2048;
                wf[(s + 1) & 1][0] = *reinterpret_cast<const h8*>(lp + o);
                wf[(s + 1) & 1][1] = *reinterpret_cast<const h8*>(lp + o + 1024);
            }
            const h8 wh = wf[s & 1][0], wl = wf[s & 1][1];
            if (SHAPE == 0) {
                c32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[s], c32, 0, 0, 0);
                c32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[s], c32, 0, 0, 0);
                c32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[s], c32, 0, 0, 0);
            } else {
                const int a = s & ~1, b = s | 1;  // the two column groups' fragments of this k-step
                c16[s & 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[a], c16[s & 1][0], 0, 0, 0);
                c16[s & 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[b], c16[s & 1][1], 0, 0, 0);
                c16[s & 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[a], c16[s & 1][0], 0, 0, 0);
                c16[s & 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[b], c16[s & 1][1], 0, 0, 0);
                c16[s & 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[a], c16[s & 1][0], 0, 0, 0);
                c16[s & 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[b], c16[s & 1][1], 0, 0, 0);
            }
            if (MODE >= 2) {
                float sv = __builtin_amdgcn_sinf(__builtin_fmaf(ev, 0.37f, eacc)) * 1.01f;
                const auto hi = __builtin_amdgcn_cvt_pkrtz(sv, eacc);
                const float lo = sv - (float)hi[0];
                const auto lo2 = __builtin_amdgcn_cvt_pkrtz(lo, ev);
                asm volatile("; keep" : "+v"(sv));
                eacc = __builtin_fmaf(sv, 0.5f, (float)lo2[0]);
                ev = sv + (float)hi[1];
            }
        }
    }
    float r = ev + eacc;
    for (int e = 0; e < 16; ++e) r += c32[e];
    for (int e = 0; e < 4; ++e) r += c16[0][0][e] + c16[0][1][e] + c16[1][0][e] + c16[1][1][e];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int SHAPE, int MODE, int BAGPR = 0>
void run(const char* name, float* d_out, _Float16* d_src, int wgs, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto kern = k<SHAPE, MODE, BAGPR>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 65536, 0, d_out, d_src, iters / 4, wgs);  // warm / clock settle
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 65536, 0, d_out, d_src, iters, wgs);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // FLOPs per k-step of the loop: 3 products x (32 x 32 x 16) x 2 in both shapes
    const double flops = (double)wgs * 4 * iters * 16 * 3 * 32768.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-44s %3d CUs: %8.1f TFLOP/s  (%.3f of 2500 x CUs/256; %.2f GHz if MFMA-bound)  %.2f ms\n", name, wgs, tf,
           tf / (2500.0 * wgs / 256.0), (double)iters * 16 * 96 / (ms * 1e-3) / 1e9, ms);
}

int main() {
    float* d_out; _Float16* d_src;
    (void)hipMalloc(&d_out, 4096); (void)hipMalloc(&d_src, 32768 * 2);
    static _Float16 hsrc[32768];
    srand(1);
    for (int i = 0; i < 32768; ++i) hsrc[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
    (void)hipMemcpy(d_src, hsrc, sizeof hsrc, hipMemcpyHostToDevice);
    const int iters = 12000;  // ~10 ms per launch at full rate
    for (int rep = 0; rep < 2; ++rep)
        for (int wgs : {256, 128}) {
            run<0, 0>("32x32x16, registers only", d_out, d_src, wgs, iters);
            run<1, 0>("16x16x32, registers only", d_out, d_src, wgs, iters);
            run<0, 1>("32x32x16 + LDS fragment reads (pipelined)", d_out, d_src, wgs, iters);
            run<1, 1>("16x16x32 + LDS fragment reads (pipelined)", d_out, d_src, wgs, iters);
            run<0, 2>("32x32x16 + LDS reads + epilogue VALU", d_out, d_src, wgs, iters);
            run<1, 2>("16x16x32 + LDS reads + epilogue VALU", d_out, d_src, wgs, iters);
            run<1, 0, 1>("16x16x32, registers only, B in AGPRs", d_out, d_src, wgs, iters);
            run<1, 1, 1>("16x16x32 + LDS reads, B in AGPRs", d_out, d_src, wgs, iters);
        }
    return 0;
}
