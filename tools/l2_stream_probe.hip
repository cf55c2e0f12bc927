// Probe (round 5): how fast ONE workgroup can stream a few MB of read-only data (the packed weights of the encoder tail +
// Modulator, 2.9 MB) from L2 / the Infinity Cache into registers, as a function of waves per workgroup, loads in flight per wave,
// workgroups launched (25 = one 320x320 slice's row blocks; 256 = every CU) and the footprint (does it stay in the 4 MB L2?).
// Every wave reads its own contiguous quarter (or 1/waves) of the buffer with global_load_dwordx4, 1 KB per instruction.
// hipcc --offload-arch=gfx950 -O3 tools/l2_stream_probe.hip -o probes/l2_stream_probe && probes/l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const u4* __restrict__ src, unsigned* __restrict__ sink, int lines_per_wave, int evict_lines) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u4* p = src + (size_t)wave * lines_per_wave * 64 + lane;
    u4 ring[DEPTH];
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) ring[u] = p[(size_t)u * 64];
    for (int i = 0; i < lines_per_wave; i += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            acc ^= ring[u][0] ^ ring[u][1] ^ ring[u][2] ^ ring[u][3];
            ring[u] = p[(size_t)(i + u + DEPTH) * 64];  // (buffer padded)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
    (void)evict_lines;
}

__global__ void evict_kernel(const u4* __restrict__ src, unsigned* __restrict__ sink, size_t n) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i][0];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH>
float run(const u4* src, unsigned* sink, int grid, int waves, size_t bytes, const u4* junk, size_t junk_n, bool evict, int reps = 20) {
    const int lines_per_wave = (int)(bytes / 1024 / waves) / DEPTH * DEPTH;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float total = 0.f;
    for (int r = 0; r < reps + 2; ++r) {
        if (evict) hipLaunchKernelGGL(evict_kernel, dim3(2048), dim3(256), 0, 0, junk, sink, junk_n);
        hipEventRecord(a);
        hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(grid), dim3(64 * waves), 0, 0, src, sink, lines_per_wave, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (r >= 2) total += ms;
    }
    return total / reps * 1000.f;  // us
}

int main() {
    const size_t bytes = 2949120, pad = 1 << 20;  // the prologue's 2.9 MB
    u4 *src, *junk;
    unsigned* sink;
    const size_t junk_bytes = 64u << 20;
    hipMalloc(&src, bytes + pad);
    hipMalloc(&junk, junk_bytes);
    hipMalloc(&sink, 4096 * 4);
    hipMemset(src, 1, bytes + pad);
    hipMemset(junk, 2, junk_bytes);
    printf("one workgroup streams %.2f MB; us per launch (GB/s per workgroup)\n", bytes / 1e6);
    printf("%-28s %10s %10s %10s %10s\n", "grid x waves, L2 state", "depth 2", "depth 4", "depth 8", "depth 16");
    for (int evict = 0; evict < 2; ++evict)
        for (int grid : {1, 25, 256})
            for (int waves : {4, 8, 16}) {
                float t2 = run<2>(src, sink, grid, waves, bytes, junk, junk_bytes / 16, evict);
                float t4 = run<4>(src, sink, grid, waves, bytes, junk, junk_bytes / 16, evict);
                float t8 = run<8>(src, sink, grid, waves, bytes, junk, junk_bytes / 16, evict);
                float t16 = run<16>(src, sink, grid, waves, bytes, junk, junk_bytes / 16, evict);
                char name[64];
                snprintf(name, sizeof name, "%3d x %2d waves, %s", grid, waves, evict ? "evicted" : "warm");
                printf("%-28s %6.1f(%3.0f) %6.1f(%3.0f) %6.1f(%3.0f) %6.1f(%3.0f)\n", name, t2, bytes / t2 / 1e3, t4, bytes / t4 / 1e3, t8, bytes / t8 / 1e3, t16,
                       bytes / t16 / 1e3);
            }
    return 0;
}
