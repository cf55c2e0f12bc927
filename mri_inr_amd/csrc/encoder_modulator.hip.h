// Producers of the modulation vectors: FixedEncoder (tile -> latent) and Modulator (latent -> mods).
//
// Reference: src/networks/encoding/siren_encoder.py:503-512,565-577 (three convolutions with
// LeakyReLU(0.2) + Linear(64, Z)) and src/networks/modulated_siren.py:325-343 (L x Linear+ReLU with
// the latent re-concatenated, hidden first).  Together they are 2.1 MFLOP per patch against the
// trunk's 303 MFLOP.  The two small convolutions run per tile on the VALU with LDS-resident feature
// maps; everything that is a Linear layer over the batch (conv3 == Linear(2048, 64), the encoder's
// Linear(64, Z), the modulator layers) runs as an exact-fp32 MFMA GEMM so that weights are read once
// per 16 patches instead of once per patch.
#pragma once
#include <hip/hip_runtime.h>

#ifndef MSIREN_ENC_C3_UNROLL
#define MSIREN_ENC_C3_UNROLL 32  // weight loads in flight per thread in the fused small-batch kernel's conv3 (8 -> 32: a single tile 67 -> 61.5 us; same summation order, same bits)
#endif

#include "encoder_params.h"

namespace msiren {

// One workgroup (256 threads) per 32x32 tile; all intermediate feature maps live in LDS.
// launch_bounds(256, 5) caps the kernel at 96 VGPRs: with two streams its waves run BESIDE the persistent
// trunk workgroup of the previous call, which leaves 104 registers per SIMD lane free.
__global__ __launch_bounds__(256, 5) void encoder_kernel(EncoderParams p, const float* __restrict__ tiles, float* __restrict__ latent) {
    __shared__ float t0[33 * 33];       // input with a zero row/column in front (padding = 1)
    __shared__ float a1[16 * 17 * 17];  // conv1 output, same front padding for conv2
    __shared__ float a2[2048];          // conv2 output, flattened (c, y, x)
    __shared__ float red[4 * 64];
    __shared__ float a3[64];
    const int tid = threadIdx.x;
    if (p.plan && (int)blockIdx.x >= p.plan[0]) return;  // workgroup-uniform
    const float* tile = enc_tile(p, tiles, p.plan ? p.plan[2 + blockIdx.x] : (int)blockIdx.x);

    for (int i = tid; i < 33 * 33; i += 256) {
        const int y = i / 33, x = i - y * 33;
        t0[i] = (y == 0 || x == 0) ? 0.f : tile[(y - 1) * 32 + (x - 1)];
    }
    for (int i = tid; i < 16 * 17 * 17; i += 256) {
        const int r = i % (17 * 17);
        if (r < 17 || r % 17 == 0) a1[i] = 0.f;
    }
    __syncthreads();

    // conv1: 16 x 16 x 16 outputs, 9 MACs each
    for (int i = tid; i < 16 * 256; i += 256) {
        const int c = i >> 8, y = (i >> 4) & 15, x = i & 15;
        float s = p.c1b[c];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) s = __builtin_fmaf(t0[(2 * y + ky) * 33 + 2 * x + kx], p.c1w[c * 9 + ky * 3 + kx], s);
        a1[c * 289 + (y + 1) * 17 + (x + 1)] = leaky02(s);
    }
    __syncthreads();

    // conv2: 32 x 8 x 8 outputs, 144 MACs each; thread = (output channel, output row)
    {
        const int o = tid & 31, y = tid >> 5;
        float s[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) s[x] = p.c2b[o];
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float w = p.c2w[(c * 9 + ky * 3 + kx) * 32 + o];
                    const float* row = &a1[c * 289 + (2 * y + ky) * 17 + kx];
#pragma unroll
                    for (int x = 0; x < 8; ++x) s[x] = __builtin_fmaf(row[2 * x], w, s[x]);
                }
#pragma unroll
        for (int x = 0; x < 8; ++x) a2[o * 64 + y * 8 + x] = leaky02(s[x]);
    }
    __syncthreads();

    // conv3 == Linear(2048, 64): 4 k-slices x 64 outputs, reduced through LDS
    {
        const int o = tid & 63, ks = tid >> 6;
        float s = 0.f;
        const float* w = p.c3w + (size_t)(ks * 512) * 64 + o;
        const float* a = a2 + ks * 512;
#pragma unroll MSIREN_ENC_C3_UNROLL
        for (int k = 0; k < 512; ++k) s = __builtin_fmaf(a[k], w[(size_t)k * 64], s);
        red[ks * 64 + o] = s;
    }
    __syncthreads();
    if (tid < 64) a3[tid] = leaky02(red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid] + p.c3b[tid]);
    __syncthreads();

    // Linear(64, Z)
    for (int z = tid; z < p.Z; z += 256) {
        float s = p.fcb[z];
#pragma unroll 8
        for (int k = 0; k < 64; ++k) s = __builtin_fmaf(a3[k], p.fcw[k * p.Z + z], s);
        latent[(size_t)blockIdx.x * p.Z + z] = s;
    }
}

// conv1 + conv2 of the encoder only: one workgroup per tile, writes the flattened conv2 output
// (c, y, x) -> feat[tile][2048]; conv3 and the Linear follow as batched MFMA GEMMs (linear_mfma_kernel).
__global__ __launch_bounds__(256, 5) void encoder_conv_kernel(EncoderParams p, const float* __restrict__ tiles, float* __restrict__ feat) {
    // conv1 output with conv2's front padding (17 x 17 per channel), columns de-interleaved into an even and an odd
    // plane of row stride 12: the stride-2 convolution then reads unit-stride along x, and a wave's 64 output
    // positions (8 x 8) fall on 32 different banks twice (bank = x - 8y mod 32) -- no conflict beyond the 2 passes a
    // 64-lane ds_read_b32 takes anyway.
    constexpr int RSP = 12, PLANE = 17 * RSP;
    __shared__ float t0[33 * 33];            // input with a zero row/column in front (padding = 1)
    __shared__ float a1[16 * 2 * PLANE];     // [channel][column parity][row][column >> 1]
    const int tid = threadIdx.x;
    if (p.plan && (int)blockIdx.x >= p.plan[0]) return;  // workgroup-uniform
    const float* tile = enc_tile(p, tiles, p.plan ? p.plan[2 + blockIdx.x] : (int)blockIdx.x);

    for (int i = tid; i < 33 * 33; i += 256) {
        const int y = i / 33, x = i - y * 33;
        t0[i] = (y == 0 || x == 0) ? 0.f : tile[(y - 1) * 32 + (x - 1)];
    }
    for (int i = tid; i < 16 * 2 * PLANE; i += 256) a1[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < 16 * 256; i += 256) {
        const int c = i >> 8, y = (i >> 4) & 15, x = i & 15;
        float s = p.c1b[c];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) s = __builtin_fmaf(t0[(2 * y + ky) * 33 + 2 * x + kx], p.c1w[c * 9 + ky * 3 + kx], s);
        const int col = x + 1;  // padded column
        a1[(c * 2 + (col & 1)) * PLANE + (y + 1) * RSP + (col >> 1)] = leaky02(s);
    }
    __syncthreads();
    // conv2: lane = output position (8 x 8), wave = 8 of the 32 output channels.  The weights of a tap are the same
    // for the whole wave: 8 consecutive floats at a wave-uniform address (scalar loads), one LDS read of the lane's
    // input feeds 8 FMAs.  Accumulation order (c, ky, kx) as in the fused kernel: same bits.
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, y = lane >> 3, x = lane & 7;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = p.c2b[8 * wv + j];
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float* even = &a1[(c * 2 + 0) * PLANE + (2 * y + ky) * RSP + x];  // padded columns 2x, 2x+2
            const float* odd = &a1[(c * 2 + 1) * PLANE + (2 * y + ky) * RSP + x];   // padded column 2x+1
            const float in[3] = {even[0], odd[0], even[1]};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* w = p.c2w + (c * 9 + ky * 3 + kx) * 32 + 8 * wv;
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] = __builtin_fmaf(in[kx], w[j], s[j]);
            }
        }
    float* dst = feat + (size_t)blockIdx.x * 2048 + (8 * wv) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j * 64] = leaky02(s[j]);
}

// ---- modulator ----------------------------------------------------------------------------------
constexpr int MOD_ROWS = 8;  // patches per workgroup

struct ModulatorLayerParams {
    const float* wt;     // (Kh + Z, H): transposed nn.Linear weight, rows ordered [hidden ; latent]
    const float* bias;   // (H)
    const float* hprev;  // (B, H) previous layer's output, or nullptr for layer 0
    const float* z;      // (B, Z)
    float* out;          // (B, H)
    int B, H, Z, Kh;
    const int* count;    // optional: number of rows to process, on the device (<= B)
};

// out[b, f] = relu(bias[f] + sum_k in[b, k] * wt[k, f]),  in = [hprev[b] ; z[b]].
// grid = (ceil(B / MOD_ROWS), ceil(H / 64)); 256 threads = 64 features x 4 k-slices.
__global__ __launch_bounds__(256) void modulator_layer_kernel(ModulatorLayerParams p) {
    extern __shared__ __attribute__((aligned(16))) float in[];  // [MOD_ROWS][K]
    __shared__ float red[4][MOD_ROWS][64];
    const int tid = threadIdx.x;
    const int K = p.Kh + p.Z;
    const int b0 = blockIdx.x * MOD_ROWS;
    const int nrows = p.count ? *p.count : p.B;
    if (b0 >= nrows) return;  // workgroup-uniform
    for (int i = tid; i < MOD_ROWS * K; i += 256) {
        const int r = i / K, k = i - r * K;
        const int b = b0 + r;
        float v = 0.f;
        if (b < nrows) v = k < p.Kh ? p.hprev[(size_t)b * p.H + k] : p.z[(size_t)b * p.Z + (k - p.Kh)];
        in[i] = v;
    }
    __syncthreads();
    const int fi = tid & 63, ks = tid >> 6;
    const int f = blockIdx.y * 64 + fi;
    const int kper = (K + 3) / 4;
    const int k0 = ks * kper, k1 = min(K, k0 + kper);
    float acc[MOD_ROWS];
#pragma unroll
    for (int r = 0; r < MOD_ROWS; ++r) acc[r] = 0.f;
    if (f < p.H) {
        for (int k = k0; k < k1; ++k) {
            const float w = p.wt[(size_t)k * p.H + f];
#pragma unroll
            for (int r = 0; r < MOD_ROWS; ++r) acc[r] = __builtin_fmaf(in[r * K + k], w, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < MOD_ROWS; ++r) red[ks][r][fi] = acc[r];
    __syncthreads();
    for (int i = tid; i < MOD_ROWS * 64; i += 256) {
        const int r = i >> 6, ff = i & 63;
        const int b = b0 + r, fo = blockIdx.y * 64 + ff;
        if (b < nrows && fo < p.H) {
            const float s = red[0][r][ff] + red[1][r][ff] + red[2][r][ff] + red[3][r][ff] + p.bias[fo];
            p.out[(size_t)b * p.H + fo] = s <= 0.f ? 0.f : s;  // NaN stays NaN, as torch.relu
        }
    }
}


// ---- Linear layer over the batch on the matrix cores ---------------------------------------------
// out[b, f] = act(bias[f] + sum_k in[b, k] * w[f, k]),  in = [hprev[b] ; z[b]]  (hprev may be absent):
// the modulator layers (act = ReLU), and with Kh = 0 the encoder's conv3 (in = conv2 features, H = 64,
// Z = 2048, LeakyReLU) and Linear(64, latent) (identity).  H, Z, Kh multiples of 16.  One workgroup = one
// 16 x 16 output tile (16 patches x 16 features); its 4 waves split K four ways and each runs a chain
// of v_mfma_f32_16x16x4_f32 (exact fp32) on float4 fragments read straight from the row-major
// operands: lane l holds in[row l&15][k0 + 4(l>>4) + j] and W[feature l&15][k0 + 4(l>>4) + j], j = 0..3,
// so one pair of 16-byte loads feeds 4 MFMAs.  `w` is the nn.Linear weight as stored, (H, Kh+Z).
typedef float mod_f32x4 __attribute__((ext_vector_type(4)));

struct ModulatorMfmaParams {
    const float* w;      // (H, Kh + Z) row-major
    const float* bias;   // (H)
    const float* hprev;  // (B, H) or nullptr
    const float* z;      // (B, Z)
    float* out;          // (B, H)
    int B, H, Z, Kh;
    int act;             // LIN_ACT_*
    const int* count;    // optional: number of rows to process, on the device (<= B)
};
enum { LIN_ACT_RELU = 0, LIN_ACT_LEAKY02 = 1, LIN_ACT_NONE = 2 };

__global__ __launch_bounds__(256) void modulator_layer_mfma_kernel(ModulatorMfmaParams p) {
    __shared__ float red[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * 16, f0 = blockIdx.y * 16;
    const int nrows = p.count ? *p.count : p.B;
    if (r0 >= nrows) return;  // workgroup-uniform
    const int row = min(r0 + (lane & 15), nrows - 1);
    const int f = f0 + (lane & 15);
    const int kq = lane >> 4;
    const int K = p.Kh + p.Z;
    const int nb = K / 16;
    const int b_lo = (nb * wave) / 4, b_hi = (nb * (wave + 1)) / 4;
    const float* wrow = p.w + (size_t)f * K + 4 * kq;
    const float* hrow = p.hprev ? p.hprev + (size_t)row * p.H + 4 * kq : nullptr;
    const float* zrow = p.z + (size_t)row * p.Z + 4 * kq;
    mod_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // batches of 8 k-blocks: all 16 fragment loads of a batch are issued before its 32 MFMAs, so the wave
    // pays one memory round trip per 128 k instead of one per 16
    int blk = b_lo;
    for (; blk + 8 <= b_hi; blk += 8) {
        mod_f32x4 a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k0 = (blk + j) * 16;
            a[j] = *reinterpret_cast<const mod_f32x4*>(k0 < p.Kh ? hrow + k0 : zrow + (k0 - p.Kh));
            b[j] = *reinterpret_cast<const mod_f32x4*>(wrow + k0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][0], b[j][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][1], b[j][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][2], b[j][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][3], b[j][3], acc1, 0, 0, 0);
        }
    }
    for (; blk < b_hi; ++blk) {
        const int k0 = blk * 16;
        const mod_f32x4 a = *reinterpret_cast<const mod_f32x4*>(k0 < p.Kh ? hrow + k0 : zrow + (k0 - p.Kh));
        const mod_f32x4 b = *reinterpret_cast<const mod_f32x4*>(wrow + k0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
    }
    // D layout: col = lane & 15 (feature), row = 4 * (lane >> 4) + reg (patch)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][lane & 15] = acc0[r] + acc1[r];
    __syncthreads();
    const int rr = tid >> 4, cc = tid & 15;
    if (r0 + rr < nrows) {
        const float s = red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc] + p.bias[f0 + cc];
        const float neg = p.act == LIN_ACT_RELU ? 0.f : (p.act == LIN_ACT_LEAKY02 ? 0.2f * s : s);
        p.out[(size_t)(r0 + rr) * p.H + f0 + cc] = s <= 0.f ? neg : s;  // NaN stays NaN, as torch's activations
    }
}

// ---- the same Linear layer on (16 RT) x (16 FT) output tiles: throughput sizes ---------------------------------------
// At a few thousand rows the 16 x 16 kernel above is bound by what it fetches, not by its MFMAs: every tile reads its own
// 16 rows of the input and 16 rows of W in full (32 KB + 32 KB at K = 512 -- 1.6 GB per Modulator layer at 25 600 rows,
// i.e. the L2s' whole bandwidth for 190 us; rocprofv3, profiles/r4/).  Here a workgroup owns RT x FT sub-tiles and every
// fragment it loads feeds RT (weights) or FT (inputs) MFMAs: half the bytes per FLOP at 2 x 2.
//
// ARITHMETIC IS THAT OF THE 16 x 16 KERNEL, element for element: the same v_mfma_f32_16x16x4_f32 on the same k groups,
// the same two accumulation chains per sub-tile (columns 0, 2 / 1, 3 of each 16-k block), K split over the four waves at
// the same block boundaries, partial sums added in the same order ((r0 + r1) + r2) + r3 + bias -- so an output does not
// depend on which of the two kernels (i.e. on how large a batch) produced it: bit-identical, tests/test_gpu_host_calls.py.
// Registers: 32 accumulators + two stages of 4 fragments at 2 x 2 -- under the 96 a kernel may use to run BESIDE the
// register-resident trunk (tests/test_register_budget.py); LDS 16.5 KB.
template <int RT, int FT>
__global__ __launch_bounds__(256, 5) void linear_mfma_tile_kernel(ModulatorMfmaParams p) {
    __shared__ float red[4][16 * RT][16 * FT + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * (16 * RT), f0 = blockIdx.y * (16 * FT);
    const int nrows = p.count ? *p.count : p.B;
    if (r0 >= nrows) return;  // workgroup-uniform
    const int kq = lane >> 4;
    const int K = p.Kh + p.Z;
    const int nb = K / 16;
    const int b_lo = (nb * wave) / 4, b_hi = (nb * (wave + 1)) / 4;
    // per-lane row bases as 32-bit element offsets from the (uniform) operand pointers; rows / features past the end are
    // clamped (computed, not stored)
    unsigned wo[FT], ho[RT], zo[RT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) wo[ft] = (unsigned)min(f0 + 16 * ft + (lane & 15), p.H - 1) * (unsigned)K + 4u * kq;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const unsigned row = (unsigned)min(r0 + 16 * rt + (lane & 15), nrows - 1);
        ho[rt] = row * (unsigned)p.H + 4u * kq;
        zo[rt] = row * (unsigned)p.Z + 4u * kq;
    }
    mod_f32x4 acc0[RT][FT], acc1[RT][FT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) acc0[rt][ft] = acc1[rt][ft] = mod_f32x4{0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int blk, mod_f32x4 (&a)[RT], mod_f32x4 (&b)[FT]) {
        const int k0 = blk * 16;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            a[rt] = *reinterpret_cast<const mod_f32x4*>(k0 < p.Kh ? p.hprev + ho[rt] + k0 : p.z + zo[rt] + (k0 - p.Kh));
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) b[ft] = *reinterpret_cast<const mod_f32x4*>(p.w + wo[ft] + k0);
    };
    auto mac = [&](const mod_f32x4 (&a)[RT], const mod_f32x4 (&b)[FT]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                acc0[rt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][0], b[ft][0], acc0[rt][ft], 0, 0, 0);
                acc1[rt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][1], b[ft][1], acc1[rt][ft], 0, 0, 0);
                acc0[rt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][2], b[ft][2], acc0[rt][ft], 0, 0, 0);
                acc1[rt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][3], b[ft][3], acc1[rt][ft], 0, 0, 0);
            }
    };
    // two register stages: the fragments of block k + 1 are in flight while block k's 4 RT FT MFMAs run
    mod_f32x4 a0[RT], b0[FT], a1[RT], b1[FT];
    int blk = b_lo;
    if (blk < b_hi) fetch(blk, a0, b0);
    for (; blk + 2 <= b_hi; blk += 2) {
        fetch(blk + 1, a1, b1);
        mac(a0, b0);
        if (blk + 2 < b_hi) fetch(blk + 2, a0, b0);
        mac(a1, b1);
    }
    if (blk < b_hi) mac(a0, b0);
    // D layout: col = lane & 15 (feature), row = 4 * (lane >> 4) + reg (patch)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][16 * rt + 4 * kq + r][16 * ft + (lane & 15)] = acc0[rt][ft][r] + acc1[rt][ft][r];
    __syncthreads();
    for (int i = tid; i < 16 * RT * 16 * FT; i += 256) {
        const int rr = i / (16 * FT), cc = i - rr * (16 * FT);
        if (r0 + rr < nrows && f0 + cc < p.H) {
            const float s = red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc] + p.bias[f0 + cc];
            const float neg = p.act == LIN_ACT_RELU ? 0.f : (p.act == LIN_ACT_LEAKY02 ? 0.2f * s : s);
            p.out[(size_t)(r0 + rr) * p.H + f0 + cc] = s <= 0.f ? neg : s;  // NaN stays NaN, as torch's activations
        }
    }
}

}  // namespace msiren
