// The Linear layers between the encoder's convolutions and the trunk as ONE launch: conv3 == Linear(2048, 64),
// Linear(64, Z) and the L Modulator layers (reference: src/networks/encoding/siren_encoder.py:503-512,565-577 and
// src/networks/modulated_siren.py:325-343,446 -- `self.modulator(self.encoder(tiles))` is one call there).
//
// Each stage is the same 16 x 16 output tile as modulator_layer_mfma_kernel (encoder_modulator.hip.h): the same fragment
// loads, the same v_mfma_f32_16x16x4_f32 chains (4 k-quarters x 2 accumulators), the same reduction order -- the results
// are bit-identical to the per-layer launches.  What changes is who runs the tiles and how a layer learns that its
// inputs are there:
//
//   * the grid is `clusters` x 16 workgroups, at most one per CU (the whole grid is resident: nothing here waits for a
//     workgroup that has not started).  A cluster owns `gpc` groups of 16 patches through ALL stages; member m of the
//     cluster computes feature tile m (m + 16, ...) of every stage for the cluster's groups, two groups at a time (they
//     share the weight fragments).
//   * hand-off between stages inside the cluster (MI355X_MICROARCH.md, "Valid forms", first row of the table): outputs
//     are stored write-through (`sc1`, 16 bytes per lane), the storing waves drain (`s_waitcnt vmcnt(0)`), the
//     workgroup's barrier, ONE lane adds 1 to the cluster's counter of that stage (agent scope); a consumer polls that
//     counter with `sc1` loads (one lane, `s_sleep` between polls), joins the workgroup's barrier, and then EVERY load of
//     handed-off bytes is an `sc1` buffer load (L1 bypassed: no acquire fence).  Weights (never written in the launch)
//     are plain loads, issued BEFORE the poll: their latency hides behind the hand-off.
//   * the counters are never reset: the host passes the value each stage's counter had before the launch (every cluster
//     of every launch adds the same amounts), comparisons are wrap-safe.  The spin is bounded: a workgroup that gives up
//     raises a word in host memory and leaves; the host reports it, re-zeroes the counters and goes back to the
//     per-layer launches (msiren.hip: chain_failed).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "encoder_modulator.hip.h"

namespace msiren {

constexpr int CHAIN_STAGES = 16;   // conv3 + Linear + up to 14 modulator layers
constexpr int CHAIN_MEMBERS = 16;  // workgroups per cluster

struct ChainStage {
    const float* w;     // (H, Ka + Kb) row-major: the nn.Linear weight as stored
    const float* bias;  // (H)
    const float* a;     // (B, Ka): the first Ka inputs of a row (previous layer's output), or nullptr (Ka = 0)
    const float* b;     // (B, Kb): the remaining inputs (latent / conv features)
    float* out;         // (B, H)
    int H, Ka, Kb, act;
};

struct ChainParams {
    ChainStage st[CHAIN_STAGES];
    unsigned base[CHAIN_STAGES];  // value of every cluster's counter of stage s before this launch
    unsigned* ctr;                // [clusters][CHAIN_STAGES]
    int nstages, B, gpc;          // gpc = groups of 16 patches per cluster
    unsigned spin_limit;
    const int* count;             // optional: number of rows to process, on the device (<= B)
    int* gave_up;                 // host-mapped word
};

typedef unsigned chain_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ mod_f32x4 chain_load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    const chain_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);  // aux 16 = sc1
    return __builtin_bit_cast(mod_f32x4, v);
}

// A stage's description in scalar registers (uniform for the compiler: descriptors built from it need no waterfall loop).
struct ChainStageS {
    const float *w, *bias, *a, *b;
    float* out;
    int H, Ka, Kb, act;
};

__device__ __forceinline__ ChainStageS chain_stage_scalars(const ChainStage& s) {
    ChainStageS r;
    r.w = s.w;  // kernel arguments: scalar loads
    r.bias = s.bias;
    r.a = s.a;
    r.b = s.b;
    r.out = s.out;
    r.H = __builtin_amdgcn_readfirstlane(s.H);
    r.Ka = __builtin_amdgcn_readfirstlane(s.Ka);
    r.Kb = __builtin_amdgcn_readfirstlane(s.Kb);
    r.act = __builtin_amdgcn_readfirstlane(s.act);
    return r;
}

// One output tile (16 features f0..) of one stage for NG groups of 16 patches (rows r0[g]..); this wave's k range is
// blocks [b_lo, b_hi) of 16, taken in batches of 8, then 4, then single blocks -- block order, hence the order of the
// MFMAs on each accumulator, is that of modulator_layer_mfma_kernel.  A batch lies on one side of the [a ; b] seam
// (Ka is a multiple of 128 or 0: msiren.hip, use_chain).  `wpre` = the wave's first batch of weights, loaded by the
// caller before the hand-off wait.
template <int NG>
__device__ __forceinline__ void chain_tile(const ChainStageS& s, int f0, const int (&r0)[NG], int nrows, int lane, int wave,
                                           float (*red)[4][16][17], const mod_f32x4 (&wpre)[8], int wpre_blocks) {
    const int K = s.Ka + s.Kb;
    const int nb = K >> 4;
    const int b_lo = (nb * wave) / 4, b_hi = (nb * (wave + 1)) / 4;
    const int kq = lane >> 4;
    const float* wrow = s.w + (size_t)(f0 + (lane & 15)) * K + 4 * kq;
    unsigned oa[NG], ob[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int row = min(r0[g] + (lane & 15), nrows - 1);
        oa[g] = (unsigned)(((size_t)row * s.Ka + 4 * kq) * 4);
        ob[g] = (unsigned)(((size_t)row * s.Kb + 4 * kq) * 4);
    }
    mod_f32x4 acc0[NG], acc1[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc0[g] = acc1[g] = mod_f32x4{0.f, 0.f, 0.f, 0.f};

    // N consecutive blocks from `blk`: every input load of the batch goes out before its MFMAs (one round trip per batch)
    auto batch = [&](int blk, auto n_tag, auto pre_tag) {
        constexpr int N = decltype(n_tag)::value;
        constexpr bool PRE = decltype(pre_tag)::value;  // the weights are the caller's `wpre`
        const int k0 = blk * 16;
        const bool in_a = k0 < s.Ka;  // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(in_a ? s.a : s.b), 0, (unsigned)((size_t)nrows * (in_a ? s.Ka : s.Kb) * 4), 0x00020000);
        const unsigned kk = 4u * (unsigned)(in_a ? k0 : k0 - s.Ka);
        mod_f32x4 w[N], in[NG][N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if constexpr (PRE) w[j] = wpre[j < 8 ? j : 0];
            else w[j] = *reinterpret_cast<const mod_f32x4*>(wrow + k0 + j * 16);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < N; ++j) in[g][j] = chain_load_sc1(rs, (in_a ? oa[g] : ob[g]) + kk + 64u * j);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                acc0[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[g][j][0], w[j][0], acc0[g], 0, 0, 0);
                acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[g][j][1], w[j][1], acc1[g], 0, 0, 0);
                acc0[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[g][j][2], w[j][2], acc0[g], 0, 0, 0);
                acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[g][j][3], w[j][3], acc1[g], 0, 0, 0);
            }
    };
    int blk = b_lo;
    if (wpre_blocks == 8) {
        batch(blk, std::integral_constant<int, 8>{}, std::true_type{});
        blk += 8;
    } else if (wpre_blocks == 4) {
        batch(blk, std::integral_constant<int, 4>{}, std::true_type{});
        blk += 4;
    }
    for (; blk + 8 <= b_hi; blk += 8) batch(blk, std::integral_constant<int, 8>{}, std::false_type{});
    if (blk + 4 <= b_hi) {
        batch(blk, std::integral_constant<int, 4>{}, std::false_type{});
        blk += 4;
    }
    for (; blk < b_hi; ++blk) batch(blk, std::integral_constant<int, 1>{}, std::false_type{});

    // D layout: col = lane & 15 (feature), row = 4 * (lane >> 4) + reg (patch)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[g][wave][4 * kq + r][lane & 15] = acc0[g][r] + acc1[g][r];
    __syncthreads();
    // wave g finishes group g: lane = (patch row, 4 features), one 16-byte write-through store
    if (wave < NG) {
        const int g = wave, rr = lane >> 2, c4 = (lane & 3) * 4;
        if (r0[g] + rr < nrows) {
            mod_f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cc = c4 + c;
                const float v = red[g][0][rr][cc] + red[g][1][rr][cc] + red[g][2][rr][cc] + red[g][3][rr][cc] + s.bias[f0 + cc];
                const float neg = s.act == LIN_ACT_RELU ? 0.f : (s.act == LIN_ACT_LEAKY02 ? 0.2f * v : v);
                o[c] = v <= 0.f ? neg : v;  // NaN stays NaN, as torch's activations
            }
            const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)s.out, 0, (unsigned)((size_t)nrows * s.H * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(chain_u32x4, o), rs_o,
                                                   (int)(((size_t)(r0[g] + rr) * s.H + f0 + c4) * 4), 0, 16);  // sc1
        }
    }
    __syncthreads();  // red is free again
}

__global__ __launch_bounds__(256) void modulator_chain_kernel(ChainParams p) {
    __shared__ float red[2][4][16][17];
    __shared__ int gave_up;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cluster = blockIdx.x / CHAIN_MEMBERS, member = blockIdx.x % CHAIN_MEMBERS;
    const int nrows = __builtin_amdgcn_readfirstlane(p.count ? *p.count : p.B);
    const int ngroups = (nrows + 15) >> 4;
    const int g_lo = cluster * p.gpc, g_hi = min(g_lo + p.gpc, ngroups);
    if (tid == 0) gave_up = 0;
    __syncthreads();
    unsigned* my_ctr = p.ctr + (size_t)cluster * CHAIN_STAGES;

    for (int si = 0; si < p.nstages; ++si) {
        const ChainStageS s = chain_stage_scalars(p.st[si]);
        const int ntiles = s.H >> 4;
        const bool active = member < ntiles;
        const bool work = active && g_lo < g_hi;
        // this wave's first weight batch: needs nothing from the launch, so it goes out before the wait
        mod_f32x4 wpre[8];
        const int K = s.Ka + s.Kb, nb = K >> 4;
        const int b_lo = (nb * wave) / 4, b_hi = (nb * (wave + 1)) / 4;
        const int wpre_blocks = !work ? 0 : (b_hi - b_lo >= 8 ? 8 : (b_hi - b_lo >= 4 ? 4 : 0));
        {
            const float* wrow = s.w + (size_t)(member * 16 + (lane & 15)) * K + 4 * (lane >> 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < wpre_blocks) wpre[j] = *reinterpret_cast<const mod_f32x4*>(wrow + (b_lo + j) * 16);
        }
        if (si > 0 && work) {
            // the previous stage of this cluster: every member that had tiles there has added 1 behind its drained stores
            if (tid == 0) {
                const int prev_tiles = p.st[si - 1].H >> 4;
                const unsigned target = p.base[si - 1] + (unsigned)min(prev_tiles, CHAIN_MEMBERS);
                unsigned spins = 0;
                while ((int)(__hip_atomic_load(my_ctr + (si - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > p.spin_limit) {
                        gave_up = 1;
                        __hip_atomic_store(p.gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(gave_up)) return;  // workgroup-uniform
        }
        if (work) {
            bool first = true;
            for (int t = member; t < ntiles; t += CHAIN_MEMBERS) {
                int g = g_lo;
                for (; g + 2 <= g_hi; g += 2) {
                    const int r0[2] = {g * 16, g * 16 + 16};
                    chain_tile<2>(s, t * 16, r0, nrows, lane, wave, red, wpre, first ? wpre_blocks : 0);
                    first = false;
                }
                if (g < g_hi) {
                    const int r0[1] = {g * 16};
                    chain_tile<1>(s, t * 16, r0, nrows, lane, wave, red, wpre, first ? wpre_blocks : 0);
                    first = false;
                }
            }
        }
        if (active) {
            // every storing wave drains its write-through stores, then the barrier, then ONE lane signals for the workgroup
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(my_ctr + si, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace msiren
