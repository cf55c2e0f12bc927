// The Linear layers between the encoder's convolutions and the trunk as ONE launch: conv3 == Linear(2048, 64),
// Linear(64, Z) and the L Modulator layers (reference: src/networks/encoding/siren_encoder.py:503-512,565-577 and
// src/networks/modulated_siren.py:325-343,446 -- `self.modulator(self.encoder(tiles))` is one call there).
//
// Each stage is the same 16 x 16 output tile as modulator_layer_mfma_kernel (encoder_modulator.hip.h): the same fragment
// values, the same v_mfma_f32_16x16x4_f32 chains (4 k-quarters x 2 accumulators, blocks in order), the same reduction
// order -- the results are bit-identical to the per-layer launches.  What changes is who runs the tiles and how a layer
// learns that its inputs are there:
//
//   * the grid is `clusters` x 16 workgroups, at most one per CU (the whole grid is resident: nothing here waits for a
//     workgroup that has not started).  A cluster owns `gpc` groups of 16 patches through ALL stages; member m of the
//     cluster computes feature tile m (m + 16, ...) of every stage for the cluster's groups, two groups at a time (they
//     share the weight fragments).
//   * a stage's output that a later stage of the launch reads travels as data-tagged granules (MI355X_MICROARCH.md,
//     "Valid forms", R2: the data IS the flag): 8 bytes {value, epoch}, written by ONE write-through (`sc1`) store (two
//     granules per 16-byte store), read by `sc1` loads (L1 bypassed) that the reading wave repeats until every tag it
//     needs equals the launch's epoch.  No counter, no flag, no fence, no drain, no barrier on the hand-off: the price is
//     one store latency plus one load round trip.  (The first form handed over plain tiles behind a per-stage counter --
//     drained sc1 stores, barrier, atomic add / sc1 poll, barrier, sc1 loads: ~8 us per stage, no faster than a launch
//     per layer: profiles/r3/11_chain_counter_handoff.)  The epoch is unique per launch and process (msiren.hip);
//     exchange buffers are zeroed when allocated, so a stale granule never carries the current epoch.
//   * the latent rows of a cluster's groups (<= CHAIN_GPC x 16 rows) are swept ONCE into the workgroup's LDS; the Modulator
//     layers take their latent half from there.  (Measured with every layer sweeping them again: 26 MB of fabric reads
//     per layer at 400 patches -- 16 members x 2 inputs x 8 bytes per value -- bounded the stage at 6.5 us.)
//   * weights (never written in the launch) are plain loads issued BEFORE the sweep: their latency hides behind it.
//     The plain copy of every output (what the trunk, a later launch, reads) is written beside the granules.
//   * every sweep is bounded: a wave that gives up raises a word in host memory; the workgroups leave at the end of the
//     stage; the host reports it and goes back to one launch per layer (msiren.hip: take_chain_flag).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "encoder_modulator.hip.h"

namespace msiren {

constexpr int CHAIN_STAGES = 16;   // conv3 + Linear + up to 14 modulator layers
constexpr int CHAIN_MEMBERS = 16;  // workgroups per cluster
constexpr int CHAIN_GPC = 2;       // groups of 16 patches a cluster owns at most (their latent rows live in LDS)
constexpr int CHAIN_ZPAD = 4;      // floats of padding per latent row in LDS: 16 rows x ds_read_b128 without bank conflicts

struct ChainStage {
    const float* w;     // (H, Ka + Kb) row-major: the nn.Linear weight as stored
    const float* bias;  // (H)
    const void* a;      // (B, Ka): the first Ka inputs of a row (previous layer's output), or nullptr (Ka = 0)
    const void* b;      // (B, Kb): the remaining inputs (latent / conv features)
    float* out;         // (B, H) plain
    void* gout;         // (B, H) granules, or nullptr: nothing in this launch reads the output
    int H, Ka, Kb, act;
    int a_gran, b_gran;  // the input is a granule array written in this launch (else plain floats written before it)
    int b_lds;           // 1: `b` is the latent: the workgroup holds its groups' rows in LDS (filled before the first such stage)
};

struct ChainParams {
    ChainStage st[CHAIN_STAGES];
    int nstages, B, gpc;  // gpc = groups of 16 patches per cluster
    unsigned epoch;       // tag of this launch's granules (never 0)
    unsigned spin_limit;
    const int* count;     // optional: number of rows to process, on the device (<= B)
    int* gave_up;         // host-mapped word
    unsigned long long* stamps;  // diagnostic (msiren_chain_timeline): per workgroup, s_memrealtime at start and after each stage
};

typedef unsigned chain_u32x4 __attribute__((ext_vector_type(4)));
typedef volatile int __attribute__((address_space(3))) * chain_lds_flag;

// One output tile (16 features f0..) of one stage for NG groups of 16 patches (rows r0[g]..); this wave's k range is
// blocks [b_lo, b_hi) of 16, taken in batches of 8, then 4, then single blocks -- block order, hence the order of the
// MFMAs on each accumulator, is that of modulator_layer_mfma_kernel.  A batch lies on one side of the [a ; b] seam
// (Ka is a multiple of 128 or 0: msiren.hip, use_chain).  `wpre` = the wave's first batch of weights, loaded by the
// caller before anything that waits.
template <int NG>
__device__ __forceinline__ void chain_tile(const ChainStage& s, const ChainParams& p, int f0, const int (&r0)[NG], int nrows, int lane,
                                           int wave, float (*red)[4][16][17], chain_lds_flag gave_up, const mod_f32x4 (&wpre)[8],
                                           int wpre_blocks, const float* zl, const int (&zslot)[NG]) {
    // (wpre_blocks is consumed below)
    const int K = s.Ka + s.Kb;
    const int nb = K >> 4;
    const int b_lo = (nb * wave) / 4, b_hi = (nb * (wave + 1)) / 4;
    const int kq = lane >> 4;
    const float* wrow = s.w + (size_t)(f0 + (lane & 15)) * K + 4 * kq;
    unsigned ea[NG], eb[NG];  // element index of the lane's first input of block 0, per side
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int row = min(r0[g] + (lane & 15), nrows - 1);
        ea[g] = (unsigned)(row * s.Ka + 4 * kq);
        eb[g] = (unsigned)(row * s.Kb + 4 * kq);
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
        const bool gran = in_a ? s.a_gran != 0 : s.b_gran != 0;
        const int Kx = in_a ? s.Ka : s.Kb;
        const unsigned kk = (unsigned)(in_a ? k0 : k0 - s.Ka);
        mod_f32x4 w[N], in[NG][N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if constexpr (PRE) w[j] = wpre[j < 8 ? j : 0];
            else w[j] = *reinterpret_cast<const mod_f32x4*>(wrow + k0 + j * 16);
        }
        if (!in_a && s.b_lds) {
            const int zs = s.Kb + CHAIN_ZPAD;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int j = 0; j < N; ++j)
                    in[g][j] = *reinterpret_cast<const mod_f32x4*>(zl + (size_t)(zslot[g] * 16 + (lane & 15)) * zs + 4 * kq + kk + 16 * j);
        } else if (!gran) {
            const float* base = (const float*)(in_a ? s.a : s.b);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int j = 0; j < N; ++j)
                    in[g][j] = *reinterpret_cast<const mod_f32x4*>(base + (size_t)(in_a ? ea[g] : eb[g]) + kk + 16 * j);
        } else {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(in_a ? s.a : s.b), 0,
                                                                                (unsigned)((size_t)nrows * Kx * 8), 0x00020000);
            chain_u32x4 q[NG][N][2];
            for (unsigned spins = 0;;) {
                bool ok = true;
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const unsigned off = ((in_a ? ea[g] : eb[g]) + kk + 16u * j) * 8u;
                        q[g][j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);  // aux 16 = sc1
                        q[g][j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16u), 0, 16);
                    }
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        ok &= q[g][j][0][1] == p.epoch && q[g][j][0][3] == p.epoch && q[g][j][1][1] == p.epoch && q[g][j][1][3] == p.epoch;
                if (__all(ok)) break;
                if (++spins > p.spin_limit || __builtin_amdgcn_readfirstlane(*gave_up)) {  // wave-uniform
                    if (lane == 0) {
                        *gave_up = 1;
                        __hip_atomic_store(p.gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    // (whole-vector casts: __builtin_bit_cast of a vector ELEMENT read element 0 every time, hipcc 7.2)
                    const mod_f32x4 f0 = __builtin_bit_cast(mod_f32x4, q[g][j][0]), f1 = __builtin_bit_cast(mod_f32x4, q[g][j][1]);
                    in[g][j] = mod_f32x4{f0[0], f0[2], f1[0], f1[2]};
                }
        }
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
    // a long k range of plain inputs (conv3: 32 blocks per wave): batches of 8, the loads of batch i+1 in flight during
    // the MFMAs of batch i (same block order on every accumulator)
    if (s.Ka == 0 && !s.b_gran && !s.b_lds && wpre_blocks == 8 && b_hi - b_lo >= 16) {
        const float* base = (const float*)s.b;
        auto load8 = [&](int bk, mod_f32x4 (&w)[8], mod_f32x4 (&in)[NG][8], bool pre) {
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = pre ? wpre[j] : *reinterpret_cast<const mod_f32x4*>(wrow + (bk + j) * 16);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int j = 0; j < 8; ++j) in[g][j] = *reinterpret_cast<const mod_f32x4*>(base + (size_t)eb[g] + (bk + j) * 16);
        };
        auto mfma8 = [&](const mod_f32x4 (&w)[8], const mod_f32x4 (&in)[NG][8]) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc0[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[g][j][0], w[j][0], acc0[g], 0, 0, 0);
                    acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[g][j][1], w[j][1], acc1[g], 0, 0, 0);
                    acc0[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[g][j][2], w[j][2], acc0[g], 0, 0, 0);
                    acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[g][j][3], w[j][3], acc1[g], 0, 0, 0);
                }
        };
        mod_f32x4 w0[8], w1[8], i0[NG][8], i1[NG][8];
        load8(blk, w0, i0, true);
        for (;;) {
            const bool more1 = blk + 16 <= b_hi;
            if (more1) load8(blk + 8, w1, i1, false);
            mfma8(w0, i0);
            blk += 8;
            if (!more1) break;
            const bool more0 = blk + 16 <= b_hi;
            if (more0) load8(blk + 8, w0, i0, false);
            mfma8(w1, i1);
            blk += 8;
            if (!more0) break;
        }
        wpre_blocks = 0;
    }
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
    // wave g finishes group g: lane = (patch row, 4 features)
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
            const size_t e = (size_t)(r0[g] + rr) * s.H + f0 + c4;
            if (s.gout) {  // granules first: someone is waiting for them
                const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(s.gout, 0, (unsigned)((size_t)nrows * s.H * 8), 0x00020000);
                const chain_u32x4 ou = __builtin_bit_cast(chain_u32x4, o);
                const chain_u32x4 g0 = {ou[0], p.epoch, ou[1], p.epoch};
                const chain_u32x4 g1 = {ou[2], p.epoch, ou[3], p.epoch};
                __builtin_amdgcn_raw_buffer_store_b128(g0, rs_o, (int)(e * 8), 0, 16);  // sc1: write-through
                __builtin_amdgcn_raw_buffer_store_b128(g1, rs_o, (int)(e * 8 + 16), 0, 16);
            }
            *reinterpret_cast<mod_f32x4*>(s.out + e) = o;
        }
    }
    __syncthreads();  // red is free again
}

__global__ __launch_bounds__(256) void modulator_chain_kernel(ChainParams p) {
    __shared__ float red[2][4][16][17];
    __shared__ int gave_up;
    extern __shared__ __attribute__((aligned(16))) float zl[];  // [CHAIN_GPC][16][Z + CHAIN_ZPAD]: the cluster's latent rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cluster = blockIdx.x / CHAIN_MEMBERS, member = blockIdx.x % CHAIN_MEMBERS;
    const int nrows = __builtin_amdgcn_readfirstlane(p.count ? *p.count : p.B);
    const int ngroups = (nrows + 15) >> 4;
    const int g_lo = cluster * p.gpc, g_hi = min(g_lo + p.gpc, ngroups);
    if (g_lo >= g_hi) return;  // nothing waits for a cluster without rows
    if (tid == 0) gave_up = 0;
    __syncthreads();
    unsigned long long* stamp = p.stamps ? p.stamps + (size_t)blockIdx.x * (CHAIN_STAGES + 1) : nullptr;
    if (stamp && tid == 0) stamp[0] = __builtin_amdgcn_s_memrealtime();

    bool z_filled = false;
    for (int si = 0; si < p.nstages; ++si) {
        const ChainStage& s = p.st[si];
        const int ntiles = s.H >> 4;
        if (member >= ntiles) continue;
        if (s.b_lds && !z_filled) {
            // the latent rows of the cluster's groups -> LDS, once: every Modulator layer reads them (16 members x L layers
            // of sc1 sweeps of the same rows otherwise: fabric traffic, not latency, bounded the stage)
            const int Z = s.Kb, zs = Z + CHAIN_ZPAD;
            const int nrow = (g_hi - g_lo) * 16, quads = Z >> 2;  // one thread-step = 4 consecutive floats of a row
            if (!s.b_gran) {
                const float* zsrc = (const float*)s.b;
                for (int i = tid; i < nrow * quads; i += 256) {
                    const int r = i / quads, c = (i - r * quads) * 4;
                    const int row = min(g_lo * 16 + r, nrows - 1);
                    *reinterpret_cast<mod_f32x4*>(zl + (size_t)r * zs + c) = *reinterpret_cast<const mod_f32x4*>(zsrc + (size_t)row * Z + c);
                }
            } else {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)s.b, 0, (unsigned)((size_t)nrows * Z * 8), 0x00020000);
                for (int i0 = 0; i0 < nrow * quads; i0 += 256 * 8) {  // 8 thread-steps (16 loads) in flight per sweep
                    chain_u32x4 q[8][2];
                    for (unsigned spins = 0;;) {
                        bool ok = true;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = min(i0 + u * 256 + tid, nrow * quads - 1);
                            const int r = i / quads, c = (i - r * quads) * 4;
                            const unsigned off = (unsigned)(min(g_lo * 16 + r, nrows - 1) * Z + c) * 8u;
                            q[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);  // sc1
                            q[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16u), 0, 16);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            ok &= q[u][0][1] == p.epoch && q[u][0][3] == p.epoch && q[u][1][1] == p.epoch && q[u][1][3] == p.epoch;
                        if (__all(ok)) break;
                        if (++spins > p.spin_limit || __builtin_amdgcn_readfirstlane(gave_up)) {  // wave-uniform
                            if (lane == 0) {
                                gave_up = 1;
                                __hip_atomic_store(p.gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * 256 + tid;
                        if (i < nrow * quads) {
                            const int r = i / quads, c = (i - r * quads) * 4;
                            const mod_f32x4 f0 = __builtin_bit_cast(mod_f32x4, q[u][0]), f1 = __builtin_bit_cast(mod_f32x4, q[u][1]);
                            *reinterpret_cast<mod_f32x4*>(zl + (size_t)r * zs + c) = mod_f32x4{f0[0], f0[2], f1[0], f1[2]};
                        }
                    }
                }
            }
            z_filled = true;
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(gave_up)) return;
        }
        // this wave's first weight batch: needs nothing from the launch, so it goes out before anything that waits
        mod_f32x4 wpre[8];
        const int K = s.Ka + s.Kb, nb = K >> 4;
        const int b_lo = (nb * wave) / 4, b_hi = (nb * (wave + 1)) / 4;
        const int wpre_blocks = b_hi - b_lo >= 8 ? 8 : (b_hi - b_lo >= 4 ? 4 : 0);
        {
            const float* wrow = s.w + (size_t)(member * 16 + (lane & 15)) * K + 4 * (lane >> 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < wpre_blocks) wpre[j] = *reinterpret_cast<const mod_f32x4*>(wrow + (b_lo + j) * 16);
        }
        bool first = true;
        for (int t = member; t < ntiles; t += CHAIN_MEMBERS) {
            int g = g_lo;
            for (; g + 2 <= g_hi; g += 2) {
                const int r0[2] = {g * 16, g * 16 + 16}, zslot[2] = {g - g_lo, g - g_lo + 1};
                chain_tile<2>(s, p, t * 16, r0, nrows, lane, wave, red, (chain_lds_flag)&gave_up, wpre, first ? wpre_blocks : 0, zl, zslot);
                first = false;
            }
            if (g < g_hi) {
                const int r0[1] = {g * 16}, zslot[1] = {g - g_lo};
                chain_tile<1>(s, p, t * 16, r0, nrows, lane, wave, red, (chain_lds_flag)&gave_up, wpre, first ? wpre_blocks : 0, zl, zslot);
                first = false;
            }
        }
        // a wave that gave up said so before the tile's barriers: every wave of the workgroup sees it here
        if (__builtin_amdgcn_readfirstlane(gave_up)) return;
        if (stamp && tid == 0) stamp[1 + si] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace msiren
