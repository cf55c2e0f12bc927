#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cmath>
__global__ void k(float* o, const float* a, const float* b) {
    float s0 = a[threadIdx.x], s1 = a[threadIdx.x + 64], m0 = b[threadIdx.x], m1 = b[threadIdx.x + 64];
    unsigned hi, lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(hi) : "v"(s0), "v"(m0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hi) : "v"(s1), "v"(m1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(s0), "v"(m0), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(s1), "v"(m1), "v"(hi));
    o[threadIdx.x] = __builtin_bit_cast(float, hi);
    o[threadIdx.x + 64] = __builtin_bit_cast(float, lo);
}
int main() {
    float *a, *b, *o; hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&o, 512);
    float ha[128], hb[128], ho[128];
    for (int i = 0; i < 128; ++i) { ha[i] = 0.37f * i - 11.3f; hb[i] = 1.0f + 0.0123f * i; }
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, a, b);
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 64; ++i) {
        unsigned hi, lo; memcpy(&hi, &ho[i], 4); memcpy(&lo, &ho[i + 64], 4);
        _Float16 h0, h1, l0, l1; unsigned short t;
        t = hi & 0xffff; memcpy(&h0, &t, 2); t = hi >> 16; memcpy(&h1, &t, 2);
        t = lo & 0xffff; memcpy(&l0, &t, 2); t = lo >> 16; memcpy(&l1, &t, 2);
        double p0 = (double)ha[i] * hb[i], p1 = (double)ha[i + 64] * hb[i + 64];
        double e0 = fabs((double)h0 + (double)l0 - p0) / fabs(p0 + 1e-30), e1 = fabs((double)h1 + (double)l1 - p1) / fabs(p1 + 1e-30);
        if (e0 > worst) worst = e0; if (e1 > worst) worst = e1;
    }
    printf("worst relative error of hi+lo vs exact product: %.3g (2^-22 = %.3g)\n", worst, 1.0 / (1 << 22));
    return 0;
}
