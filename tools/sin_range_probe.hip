#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* x, float* y, int n) { int i = threadIdx.x; if (i < n) y[i] = __builtin_amdgcn_sinf(x[i]); }
int main() {
    float hx[12] = {0.25f, 255.25f, 256.25f, 300.25f, 511.25f, 512.25f, 1000.25f, 4096.25f, 65536.25f, 1048576.25f, -300.25f, 3.0e9f};
    float *dx, *dy, hy[12];
    (void)hipMalloc(&dx, 48); (void)hipMalloc(&dy, 48);
    (void)hipMemcpy(dx, hx, 48, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, 12);
    (void)hipMemcpy(hy, dy, 48, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("v_sin_f32(%.2f rev) = %.7f   (exact %.7f)\n", hx[i], hy[i], std::sin(2.0 * M_PI * std::fmod((double)hx[i], 1.0)));
    return 0;
}
