// Probe: accuracy of v_sin_f32 (input in revolutions) and v_exp_f32 on gfx950, against fp64.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/sin_accuracy.hip -o /tmp/sinacc && /tmp/sinacc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const float* x, float* s, float* e, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float r = x[i];
        float f = r - __builtin_rintf(r);
        s[i] = __builtin_amdgcn_sinf(f);
        e[i] = __builtin_amdgcn_sinf(r);
    }
}

int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), e(n);
    for (int i = 0; i < n; ++i) {
        double t = (double)i / n;
        x[i] = (float)((i & 1) ? (t - 0.5) : (t - 0.5) * 64.0);  // [-0.5,0.5] and [-32,32] revolutions
    }
    float *dx, *ds, *de;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&de, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, de, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(e.data(), de, n * 4, hipMemcpyDeviceToHost);
    double m_small = 0, m_big = 0, m_big_direct = 0, m_small_rel = 0;
    for (int i = 0; i < n; ++i) {
        double ref = std::sin(2.0 * M_PI * (double)x[i]);
        double d = std::fabs((double)s[i] - ref), dd = std::fabs((double)e[i] - ref);
        if (i & 1) { m_small = std::fmax(m_small, d); if (std::fabs(ref) > 1e-3) m_small_rel = std::fmax(m_small_rel, d / std::fabs(ref)); }
        else { m_big = std::fmax(m_big, d); m_big_direct = std::fmax(m_big_direct, dd); }
    }
    printf("v_sin_f32 max abs err: |r|<=0.5 reduced %.3e (max rel %.3e) ; |r|<=32 reduced %.3e ; |r|<=32 direct %.3e\n",
           m_small, m_small_rel, m_big, m_big_direct);
    return 0;
}
