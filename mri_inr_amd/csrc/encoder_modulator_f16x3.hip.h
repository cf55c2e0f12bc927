// Producers of the modulation vectors in the trunk's split-fp16 arithmetic: FixedEncoder's tail (conv3 == Linear(2048, 64),
// Linear(64, Z)) and the whole Modulator as ONE launch, a row block of 16 patches per workgroup through every layer.
//
// Reference: src/networks/encoding/siren_encoder.py:503-512,565-577 and src/networks/modulated_siren.py:325-343
//     a3 = leaky(W3 a2 + b3);  z = Wfc a3 + bfc;  h_0 = relu(M_0 z + c_0);  h_l = relu(M_l [h_{l-1} ; z] + c_l)
// (hidden first, latent second, :341).  Rows (patches) are independent through all of it, so a workgroup that owns 16 rows
// needs no hand-off to any other workgroup: the seven dependent launches of encoder_modulator.hip.h (68 us in front of a 261 us
// trunk at one slice per call; 0.6 ms per 25 600 tiles at ~64 TFLOP/s of fp32 MFMA) become one launch whose cost is streaming
// the 2.9 MB of weights through the workgroup once.
//
// Arithmetic (the trunk's, siren_trunk_f16_common.hip.h): every operand is v = hi + lo with hi = f16(v), lo = f16(v - hi)
// (22 significant bits), a product is W_lo x_hi + W_hi x_lo + W_hi x_hi on v_mfma_f32_16x16x32_f16, fp32 accumulation.
// Unlike the trunk's activations (|sin| <= 1) these operands have no natural range -- fastMRI intensities are ~1e-5, conv
// features and modulations follow -- so both sides are scaled by exact powers of two first:
//   * weights: per layer, max|W| -> [2^13, 2^14) (host, once);
//   * inputs: PER ROW, max_k |x[row][k]| -> [2^13, 2^14), computed where the row is produced (a row never sees another row's
//     scale: an output does not depend on the batch it came in);
//   * the accumulator is multiplied back by 2^-(a + s) (exact) before bias and activation.
// Elements more than 2^17 below their row's (layer's) maximum keep fewer than 22 bits (fp16 subnormals: absolute error 2^-39
// of that maximum) -- far below what fp32 rounding of the dominant terms leaves.  Non-finite inputs give non-finite outputs.
//
// The latent part of every Modulator layer is hoisted: [h ; z] has two row scales, so  M_l [h ; z] = Mh_l h + Mz_l z  is
// evaluated as a z stage (all layers' Mz_l z + c_l at once, K = Z, kept per lane in a scratch buffer) followed by the
// hidden chain with K = H -- half the dependent work per layer.
//
// Data flow.  A wave owns two 16-feature output tiles of the current pass (8 tiles = 128 features per pass and workgroup)
// and runs the K loop over them: A = weight fragments straight from the packed stream (global_load_dwordx4, a register ring
// DEPTH k-steps ahead, running on across pass and layer boundaries -- weights do not depend on data), B = the input image in
// LDS, [k-step][hi|lo][64 lanes][8 x f16], which all four waves read.  The D layout of the MFMA (lane (n, q): features
// 4q..4q+3 of patch n) is, after the epilogue, this lane's 16-byte piece of the NEXT stage's B image: element j of k-step s
// is input 32 s + 16 (j >> 2) + 4 q + (j & 3), and the host packs the weights in that k order.
// Bound: the 64 B/clk a CU's vector memory path delivers (4 KB of weight fragments per wave and k-step against 6 MFMAs).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "encoder_params.h"  // EncoderParams, leaky02
#include "host_plan.h"       // em_ring_depth_ok: the rule the packer, the kernel and the CPU test share

namespace msiren {

typedef _Float16 em_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 em_h2 __attribute__((ext_vector_type(2)));
typedef float em_f4 __attribute__((ext_vector_type(4)));
typedef unsigned em_u4 __attribute__((ext_vector_type(4)));

constexpr int EM_ROWS = 16;          // patches per workgroup (one MFMA N tile)
constexpr int EM_C3_KSTEPS = 64;     // conv3: K = 2048
constexpr int EM_MAX_DEPTH = 16;     // deepest weight ring: streams and feature buffers are padded by this many k-steps
constexpr int EM_FC_KSTEPS = 4;      // Linear(64, Z): K = 64, padded to 128 with zero weights / a zero image (two passes = one round of a ring of 8)

struct EmTailParams {
    const em_u4* wstream;    // [4 waves][k-steps in consumption order][tile 0 hi | tile 0 lo | tile 1 hi | tile 1 lo][64 lanes][8 x f16]
    const float* bias;       // [conv3: 64][fc: Z][modulator: L x H]
    const em_u4* feat;       // conv2 features as B images, [row block][64 k-steps][hi|lo][64 lanes][8 x f16]; null: the latent is given
    const float* feat_inv;   // [row]: 2^-s of the row's image
    const float* z_in;       // (B, Z) when feat == null
    float* z_out;            // optional (B, Z)
    float* mods;             // (L, B, H); null: encoder only
    em_f4* cscratch;         // [row block][(L-1) NPH][4 waves][2 tiles][64 lanes]: Mz_l z + c_l of layers 1.., lane-private
    float winv_c3, winv_fc;  // exact inverses of the layers' power-of-two weight scales
    float winv_z[64], winv_h[64];  // latent / hidden part of Modulator layer l
    int B, L;
    int wave_stride;         // em_u4 per wave stream
    int zp_start;            // k-step at which the z stage begins in a wave's stream (entry point when the latent is given)
    const int* count;        // optional: number of rows to process, on the device (<= B)
    // Latency sizes only: the grid carries `pf_blocks` workgroups behind the row blocks that do nothing but pull the weight stream into
    // their XCD's L2 -- between two calls the trunk's traffic evicts it (rocprofv3: every XCD fetches the 2.9 MB again, 29 MB per
    // launch), and ONE workgroup streams from the Infinity Cache at 96 GB/s where it gets 127 GB/s from L2 (tools/l2_stream_probe.hip).
    // Workgroup e of them takes slice e / 8 of eight (consecutive workgroup ids go to consecutive XCDs), in consumption order.
    int row_blocks, pf_blocks;
    unsigned pf_lines;       // 1 KB lines (64 lanes x 16 B) per slice
};

// max -> power-of-two scale of a row: m * 2^s in [2^13, 2^14); s clamped so that 2^s and 2^-s are normal numbers
__device__ __forceinline__ void em_row_scale(float m, float& sc, float& inv) {
    const unsigned e = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu;  // m >= 0
    int s = m > 0.f ? 14 - ((int)e - 126) : 0;
    s = s > 100 ? 100 : (s < -100 ? -100 : s);
    sc = __builtin_bit_cast(float, (unsigned)(127 + s) << 23);
    inv = __builtin_bit_cast(float, (unsigned)(127 - s) << 23);
}

// eight fp32 values (already scaled) -> their hi and lo fragments
__device__ __forceinline__ void em_split8(const em_f4 a, const em_f4 b, em_u4& hi, em_u4& lo) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 h0 = (_Float16)v[2 * i], h1 = (_Float16)v[2 * i + 1];
        const _Float16 l0 = (_Float16)(v[2 * i] - (float)h0), l1 = (_Float16)(v[2 * i + 1] - (float)h1);
        hi[i] = __builtin_bit_cast(unsigned, em_h2{h0, h1});
        lo[i] = __builtin_bit_cast(unsigned, em_h2{l0, l1});
    }
}

__device__ __forceinline__ em_f4 em_mfma(const em_u4 a, const em_u4 b, const em_f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(em_h8, a), __builtin_bit_cast(em_h8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float em_absmax8(const em_f4 a, const em_f4 b) {
    float m = __builtin_fabsf(a[0]);
#pragma unroll
    for (int i = 1; i < 4; ++i) m = __builtin_fmaxf(m, __builtin_fabsf(a[i]));
#pragma unroll
    for (int i = 0; i < 4; ++i) m = __builtin_fmaxf(m, __builtin_fabsf(b[i]));
    return m;
}

// NPH = H / 128, NPZ = Z / 128 (passes of 8 tiles per layer); DEPTH = k-steps of weight fragments in flight per wave
// (2: <= 96 registers, runs BESIDE the register-resident trunk of the other stream; 4 / 8: workgroups with CUs of their own --
// one CU streams 2.9 MB in 23 us from a warm L2 and in 30 us from the Infinity Cache once >= 8 k-steps per wave are in flight
// (tools/l2_stream_probe.hip); the ring of 16 = 256 registers holds 1.5 us of stream, so that the fragments keep arriving
// through a layer's epilogue -- row maxima, two barriers, the split -- during which no wave issues a load).
// MODE: 1 = the encoder's tail only (features -> latent), 2 = the Modulator only (latent -> modulations), 3 = both, the latent
// staying in the workgroup (compile-time: as a run-time choice the compiler hoisted the other branch's loads over the first stage
// and spilled them).
template <int NPH, int NPZ, int DEPTH, int MODE>
__global__ __launch_bounds__(256, DEPTH == 2 ? 5 : (DEPTH == 4 ? 2 : 1)) void latent_mods_f16x3_kernel(EmTailParams p) {
    constexpr bool ENC = (MODE & 1) != 0, MOD = (MODE & 2) != 0;
    constexpr int H = 128 * NPH, Z = 128 * NPZ;
    constexpr int KH = H / 32, KZ = Z / 32;           // k-steps of a hidden / latent image
    constexpr int ZIMG = 0, HIMG = Z * 4;             // em_u4 offsets: an image is K/32 k-steps x 128 em_u4
    constexpr int RMAX = (HIMG + H * 4) * 16;         // byte offset of the 4 x 16 row maxima
    // ring slot of a k-step = its position in the wave's stream modulo DEPTH, and every slot index is a compile-time constant:
    // the stages whose count is a run-time value (L layers) must advance the position by a multiple of DEPTH per iteration
    static_assert(em_ring_depth_ok(NPH, NPZ, DEPTH, EM_C3_KSTEPS / 2), "ring depth must divide every layer");
    constexpr int PHZ = (NPZ * EM_FC_KSTEPS) % DEPTH;  // ring phase at which the z stage (and every layer behind it) starts
    constexpr bool EARLY = DEPTH > 2;  // fetch what an epilogue reads from global memory in front of its K loop (costs 8 registers)
    constexpr int BD = DEPTH < 8 ? DEPTH : 8;           // conv3's B images come from HBM / L2 as well: their own, shallower ring
    static_assert(H * 4 >= EM_FC_KSTEPS * 128 && Z * 4 >= 256, "the conv3 image and partial sums alias the H / Z images");
    extern __shared__ __attribute__((aligned(16))) em_u4 em_smem[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nrows = p.count ? *p.count : p.B;
    const int rb = blockIdx.x, r0 = rb * EM_ROWS;
    if (p.pf_blocks && rb >= p.row_blocks) {  // workgroup-uniform
        const em_u4* src = p.wstream + (size_t)((rb - p.row_blocks) >> 3) * p.pf_lines * 64 + lane;
        unsigned acc_ = 0;
        for (unsigned i = wave; i < p.pf_lines; i += 32) {  // 8 lines per wave in flight; plain loads: the compiler counts them
            em_u4 v[8];
#pragma unroll
            for (unsigned u = 0; u < 8; ++u) {
                const unsigned j = i + 4 * u < p.pf_lines ? i + 4 * u : i;
                v[u] = src[(size_t)j * 64];
            }
#pragma unroll
            for (unsigned u = 0; u < 8; ++u) acc_ ^= v[u][0];
        }
        asm volatile("; the prefetched lines have arrived" ::"v"(acc_));
        return;
    }
    if (r0 >= nrows) return;  // workgroup-uniform
    // (beside a trunk wave on its SIMD this wave issues little -- a vector load here, six MFMAs there -- but each of its
    //  instructions is on the critical path of a 25-workgroup launch: let it win the arbitration)
    if constexpr (DEPTH == 2) __builtin_amdgcn_s_setprio(3);
    const bool live = r0 + n < nrows;
    const int row = live ? r0 + n : nrows - 1;  // rows past the end: computed on a clamped row, never stored
    const int L = p.L;
    em_u4* const zimg = em_smem + ZIMG + lane;
    em_u4* const himg = em_smem + HIMG + lane;
    float* const rmax = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(em_smem) + RMAX);

    // ---- the weight stream of this wave: ring[u] = fragments of k-step g + u, g = the k-step about to be consumed ----
    const em_u4* const wp = p.wstream + (size_t)wave * p.wave_stride + lane;
    int g = ENC ? 0 : p.zp_start;
    em_u4 ring[DEPTH][4];
    if constexpr (ENC) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[u][i] = wp[(size_t)(g + u) * 256 + i * 64];
    } else {  // (entry at the z stage: the same slots as when the stream is consumed from its start)
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[(PHZ + u) % DEPTH][i] = wp[(size_t)(g + u) * 256 + i * 64];
    }

    em_f4 acc[2];
    // one k-step: 2 tiles x 3 products; then the slot is refilled DEPTH k-steps ahead
#define EM_KSTEP(U, BH, BL)                                                                  \
    do {                                                                                     \
        acc[0] = em_mfma(ring[U][1], BH, acc[0]);                                            \
        acc[1] = em_mfma(ring[U][3], BH, acc[1]);                                            \
        acc[0] = em_mfma(ring[U][0], BL, acc[0]);                                            \
        acc[1] = em_mfma(ring[U][2], BL, acc[1]);                                            \
        acc[0] = em_mfma(ring[U][0], BH, acc[0]);                                            \
        acc[1] = em_mfma(ring[U][2], BH, acc[1]);                                            \
        /* (unconditional: the stream is padded by EM_MAX_DEPTH k-steps, so that no branch splits the loop body and */ \
        /*  the compiler counts the loads in flight exactly) */                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) ring[U][i_] = wp[(size_t)(g + DEPTH) * 256 + i_ * 64]; \
        ++g;                                                                                 \
        /* (left alone the scheduler sinks the refills towards their use, DEPTH k-steps later: nothing would be in flight) */ \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)

    // K loop of one pass over an image in LDS: NK k-steps, fully unrolled; the ring slot of k-step k is (PH0 + k) % DEPTH
    auto pass_lds = [&](const em_u4* img, auto nk_c, auto ph0_c) {
        constexpr int NK = decltype(nk_c)::value, PH0 = decltype(ph0_c)::value;
        acc[0] = acc[1] = em_f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (DEPTH == 2) {  // (the instance that lives on 96 registers: no second B buffer)
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const em_u4 bh = img[k * 128], bl = img[k * 128 + 64];
                EM_KSTEP((PH0 + k) % DEPTH, bh, bl);
            }
        } else {
            em_u4 bh = img[0], bl = img[64];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                em_u4 nh = bh, nl = bl;
                if (k + 1 < NK) {
                    nh = img[(k + 1) * 128];
                    nl = img[(k + 1) * 128 + 64];
                }
                EM_KSTEP((PH0 + k) % DEPTH, bh, bl);
                bh = nh;
                bl = nl;
            }
        }
    };
    using em_zero = std::integral_constant<int, 0>;

    // row maxima of the 16 patches over the features all waves hold -> this lane's row scale; `mine` = max over this lane's values
    auto row_scale = [&](float mine, int nwaves, float& sc, float& inv) {
        mine = __builtin_fmaxf(mine, __shfl_xor(mine, 16));
        mine = __builtin_fmaxf(mine, __shfl_xor(mine, 32));
        if (q == 0 && wave < nwaves) rmax[wave * 16 + n] = mine;
        __syncthreads();  // (also: every wave has finished reading the image that is about to be rewritten)
        float m = rmax[n];
        for (int w = 1; w < nwaves; ++w) m = __builtin_fmaxf(m, rmax[w * 16 + n]);
        em_row_scale(m, sc, inv);
    };

    float inv_z = 1.f;  // 2^-s of this lane's row in the Z image
    em_f4 held[NPH > NPZ ? NPH : NPZ][2];

    if constexpr (ENC) {
        // ---- conv3 == Linear(2048, 64): 4 tiles; wave = (tile pair w & 1, K half w >> 1); B = the conv kernel's images ----
        // (everything an epilogue reads from global memory is fetched BEFORE its K loop: behind it the load's whole latency would be exposed)
        // (the 96-register instance fetches them behind the loop: it runs beside a trunk, where latency is not what it is short of)
        float finv;
        em_f4 b3_0, b3_1;
        auto fetch_c3 = [&]() {
            finv = p.feat_inv[row];
            b3_0 = *reinterpret_cast<const em_f4*>(p.bias + 32 * (wave & 1) + 4 * q);
            b3_1 = *reinterpret_cast<const em_f4*>(p.bias + 32 * (wave & 1) + 16 + 4 * q);
        };
        if constexpr (EARLY) fetch_c3();
        {
            const em_u4* fb = p.feat + ((size_t)rb * EM_C3_KSTEPS + (size_t)(wave >> 1) * (EM_C3_KSTEPS / 2)) * 128 + lane;
            em_u4 bring[BD][2];
#pragma unroll
            for (int u = 0; u < BD; ++u) {
                bring[u][0] = fb[u * 128];
                bring[u][1] = fb[u * 128 + 64];
            }
            acc[0] = acc[1] = em_f4{0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < EM_C3_KSTEPS / 2; k += DEPTH) {
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    const em_u4 bh = bring[u % BD][0], bl = bring[u % BD][1];
                    bring[u % BD][0] = fb[(k + u + BD) * 128];  // (runs up to EM_MAX_DEPTH k-steps past the K half: the buffer is padded)
                    bring[u % BD][1] = fb[(k + u + BD) * 128 + 64];
                    EM_KSTEP(u, bh, bl);
                }
            }
        }
        if constexpr (!EARLY) fetch_c3();
        // zero k-steps 2, 3 of the conv3 image (the Linear's K = 64 is padded to the ring depth); partial sums of K half 1 -> LDS
        himg[2 * 128] = himg[2 * 128 + 64] = himg[3 * 128] = himg[3 * 128 + 64] = em_u4{0u, 0u, 0u, 0u};
        em_f4* const part = reinterpret_cast<em_f4*>(em_smem + ZIMG) + lane;  // [tile pair][tile][64 lanes]
        if (wave >= 2) {
            part[((wave & 1) * 2 + 0) * 64] = acc[0];
            part[((wave & 1) * 2 + 1) * 64] = acc[1];
        }
        __syncthreads();
        em_f4 a3[2] = {em_f4{0.f, 0.f, 0.f, 0.f}, em_f4{0.f, 0.f, 0.f, 0.f}};
        float mine = 0.f;
        if (wave < 2) {
            const float u = finv * p.winv_c3;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const em_f4 s = acc[t] + part[(wave * 2 + t) * 64];  // (K half 0) + (K half 1)
                const em_f4 b = t ? b3_1 : b3_0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = __builtin_fmaf(s[r], u, b[r]);
                    a3[t][r] = v <= 0.f ? 0.2f * v : v;  // LeakyReLU(0.2); NaN stays NaN
                }
            }
            mine = em_absmax8(a3[0], a3[1]);
        }
        float sc, inv_a3;
        row_scale(mine, 2, sc, inv_a3);
        if (wave < 2) {
            em_u4 hi, lo;
            em_split8(a3[0] * sc, a3[1] * sc, hi, lo);
            himg[wave * 128] = hi;
            himg[wave * 128 + 64] = lo;
        }
        __syncthreads();
        // ---- Linear(64, Z): NPZ passes ----
#pragma unroll
        for (int pz = 0; pz < NPZ; ++pz) {
            static_assert(NPZ <= 2, "ring phases of the Linear's passes");
            em_f4 bf[2];
            auto fetch_fc = [&]() {
                bf[0] = *reinterpret_cast<const em_f4*>(p.bias + 64 + 128 * pz + 32 * wave + 4 * q);
                bf[1] = *reinterpret_cast<const em_f4*>(p.bias + 64 + 128 * pz + 32 * wave + 16 + 4 * q);
            };
            if constexpr (EARLY) fetch_fc();
            if (pz == 0) pass_lds(himg, std::integral_constant<int, EM_FC_KSTEPS>{}, em_zero{});
            else pass_lds(himg, std::integral_constant<int, EM_FC_KSTEPS>{}, std::integral_constant<int, EM_FC_KSTEPS % DEPTH>{});
            if constexpr (!EARLY) fetch_fc();
            const float u = inv_a3 * p.winv_fc;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) held[pz][t][r] = __builtin_fmaf(acc[t][r], u, bf[t][r]);
            }
        }
    } else {
#pragma unroll
        for (int pz = 0; pz < NPZ; ++pz)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                held[pz][t] = *reinterpret_cast<const em_f4*>(p.z_in + (size_t)row * Z + 128 * pz + 32 * wave + 16 * t + 4 * q);
    }
    // ---- the latent: out to HBM if asked for, row scale, Z image ----
    {
        float mine = 0.f;
#pragma unroll
        for (int pz = 0; pz < NPZ; ++pz) {
            mine = __builtin_fmaxf(mine, em_absmax8(held[pz][0], held[pz][1]));
            if (p.z_out && live)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    *reinterpret_cast<em_f4*>(p.z_out + (size_t)row * Z + 128 * pz + 32 * wave + 16 * t + 4 * q) = held[pz][t];
        }
        if constexpr (!MOD) return;  // encoder only
        float sc;
        row_scale(mine, 4, sc, inv_z);
#pragma unroll
        for (int pz = 0; pz < NPZ; ++pz) {
            em_u4 hi, lo;
            em_split8(held[pz][0] * sc, held[pz][1] * sc, hi, lo);
            zimg[(4 * pz + wave) * 128] = hi;
            zimg[(4 * pz + wave) * 128 + 64] = lo;
        }
        __syncthreads();
    }
    // ---- z stage: Mz_l z + c_l for every layer; layer 0 is h_0 itself ----
    const float* const mbias = p.bias + 64 + Z;
    em_f4* const cs = p.cscratch + ((size_t)rb * (L - 1) * NPH * 4 + wave) * 128 + lane;  // + ((l - 1) NPH + ph) * 512 + t * 64
    float inv_h = 1.f;
    for (int l = 0; l < L; ++l) {
        const float u = inv_z * p.winv_z[l];
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            static_assert(NPH == 2 || NPH == 4, "ring phases of a layer's passes");
            em_f4 bz[2];
            auto fetch_z = [&]() {
                bz[0] = *reinterpret_cast<const em_f4*>(mbias + (size_t)l * H + 128 * ph + 32 * wave + 4 * q);
                bz[1] = *reinterpret_cast<const em_f4*>(mbias + (size_t)l * H + 128 * ph + 32 * wave + 16 + 4 * q);
            };
            if constexpr (EARLY) fetch_z();
            if (ph == 0) pass_lds(zimg, std::integral_constant<int, KZ>{}, std::integral_constant<int, PHZ>{});
            else if (ph == 1) pass_lds(zimg, std::integral_constant<int, KZ>{}, std::integral_constant<int, (PHZ + KZ) % DEPTH>{});
            else if (ph == 2) pass_lds(zimg, std::integral_constant<int, KZ>{}, std::integral_constant<int, (PHZ + 2 * KZ) % DEPTH>{});
            else pass_lds(zimg, std::integral_constant<int, KZ>{}, std::integral_constant<int, (PHZ + 3 * KZ) % DEPTH>{});
            if constexpr (!EARLY) fetch_z();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                em_f4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[t][r], u, bz[t][r]);
                if (l == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] <= 0.f ? 0.f : v[r];  // ReLU; NaN stays NaN, as torch.relu
                    held[ph][t] = v;
                } else {
                    cs[(size_t)((l - 1) * NPH + ph) * 512 + t * 64] = v;
                }
            }
        }
        if (l == 0) {
            float mine = 0.f;
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                mine = __builtin_fmaxf(mine, em_absmax8(held[ph][0], held[ph][1]));
                if (live)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        *reinterpret_cast<em_f4*>(p.mods + (size_t)row * H + 128 * ph + 32 * wave + 16 * t + 4 * q) = held[ph][t];
            }
            if (L > 1) {
                float sc;
                row_scale(mine, 4, sc, inv_h);
#pragma unroll
                for (int ph = 0; ph < NPH; ++ph) {
                    em_u4 hi, lo;
                    em_split8(held[ph][0] * sc, held[ph][1] * sc, hi, lo);
                    himg[(4 * ph + wave) * 128] = hi;
                    himg[(4 * ph + wave) * 128 + 64] = lo;
                }
                __syncthreads();
            }
        }
    }
    // ---- hidden chain: h_l = relu(Mh_l h_{l-1} + (Mz_l z + c_l)) ----
    for (int l = 1; l < L; ++l) {
        const float u = inv_h * p.winv_h[l];
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            em_f4 c0, c1;
            auto fetch_c = [&]() {
                c0 = cs[(size_t)((l - 1) * NPH + ph) * 512];
                c1 = cs[(size_t)((l - 1) * NPH + ph) * 512 + 64];
            };
            if constexpr (EARLY) fetch_c();
            if (ph == 0) pass_lds(himg, std::integral_constant<int, KH>{}, std::integral_constant<int, PHZ>{});
            else if (ph == 1) pass_lds(himg, std::integral_constant<int, KH>{}, std::integral_constant<int, (PHZ + KH) % DEPTH>{});
            else if (ph == 2) pass_lds(himg, std::integral_constant<int, KH>{}, std::integral_constant<int, (PHZ + 2 * KH) % DEPTH>{});
            else pass_lds(himg, std::integral_constant<int, KH>{}, std::integral_constant<int, (PHZ + 3 * KH) % DEPTH>{});
            if constexpr (!EARLY) fetch_c();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v0 = __builtin_fmaf(acc[0][r], u, c0[r]), v1 = __builtin_fmaf(acc[1][r], u, c1[r]);
                held[ph][0][r] = v0 <= 0.f ? 0.f : v0;
                held[ph][1][r] = v1 <= 0.f ? 0.f : v1;
            }
        }
        float mine = 0.f;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            mine = __builtin_fmaxf(mine, em_absmax8(held[ph][0], held[ph][1]));
            if (live)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    *reinterpret_cast<em_f4*>(p.mods + ((size_t)l * p.B + row) * H + 128 * ph + 32 * wave + 16 * t + 4 * q) = held[ph][t];
        }
        if (l + 1 < L) {
            float sc;
            row_scale(mine, 4, sc, inv_h);
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                em_u4 hi, lo;
                em_split8(held[ph][0] * sc, held[ph][1] * sc, hi, lo);
                himg[(4 * ph + wave) * 128] = hi;
                himg[(4 * ph + wave) * 128 + 64] = lo;
            }
            __syncthreads();
        }
    }
#undef EM_KSTEP
}

template <int NPH, int NPZ>
constexpr int em_tail_lds_bytes() { return (128 * NPZ * 4 + 128 * NPH * 4) * 16 + 256; }
static_assert(em_tail_lds_bytes<2, 2>() <= 34 * 1024, "the H = Z = 256 instance runs beside the register-resident trunk: what its ring of 3 leaves free");

// ---- conv1 + conv2 of the encoder, writing conv3's B images --------------------------------------------------------------
// One workgroup per 32 x 32 tile.  The tile's 2048 conv2 features are scaled by the row's power of two, split, and stored as
// this row's column of the row block's 64 k-step images; thread (wave w, lane) stores the 16-byte piece (k-step 16 w + (lane >> 2),
// q = lane & 3), so conv3's k order is whatever values the thread holds (the host packs conv3's weights to match, per variant).
//
// VARIANT 1 (the only one left; the name stays what the round-5 profiles call it): conv2 as an implicit GEMM on the matrix cores in the split-fp16 arithmetic.  conv1 (K = 9: VALU, one thread per
//   output position, all 16 channels, fp32 FMAs in the reference order) leaves its output in LDS as conv2's B operand:
//   [padded position 17 x 17][hi: 16 channels | lo: 16 channels] x f16, scaled by the tile's own power of two.  A k-step is two
//   taps x 16 channels (9 taps -> 5 k-steps, the tenth tap has zero weights), so a lane's eight B elements are eight consecutive
//   channels of one position: one ds_read_b128.  Wave w = (channel tile w & 1, position tiles 2 (w >> 1) + {0, 1}); the
//   weights of its channel tile sit in 40 registers, fetched while conv1 runs.  Thread (w, lane (n, q)) ends up with channels
//   16 (w & 1) + 4 q + r at positions 16 (2 (w >> 1) + t) + n  (j = 4 t + r).
template <int VARIANT>
__global__ __launch_bounds__(256, 5) void encoder_conv_f16x3_kernel(EncoderParams p, const float* __restrict__ tiles, em_u4* __restrict__ feat,
                                                                      float* __restrict__ feat_inv) {
    __shared__ float t0[33 * 33];
    __shared__ float wmax[8];
    const int tid = threadIdx.x;
    if (p.plan && (int)blockIdx.x >= p.plan[0]) return;  // workgroup-uniform
    const float* tile = enc_tile(p, tiles, p.plan ? p.plan[2 + blockIdx.x] : (int)blockIdx.x);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    em_f4 o0, o1;  // the thread's eight features (before scale and split)

    static_assert(VARIANT == 1, "VARIANT 0 (conv2 on the VALU: 17.9 us against 10.3 per 400 tiles, profiles/r5/02_*) left the library in round 6");
    {
        __shared__ __attribute__((aligned(16))) em_u4 a1img[17 * 17 * 4];  // [position][hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]
        const int n = lane & 15, q = lane >> 4, mt = wv & 1;
        // conv2's weights of this wave's channel tile: in flight while conv1 runs
        em_u4 wa[5][2];
        {
            const em_u4* wsrc = reinterpret_cast<const em_u4*>(p.c2f16) + (size_t)mt * 5 * 2 * 64 + lane;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                wa[s][0] = wsrc[(s * 2 + 0) * 64];
                wa[s][1] = wsrc[(s * 2 + 1) * 64];
            }
        }
        const em_f4 b2 = *reinterpret_cast<const em_f4*>(p.c2b + 16 * mt + 4 * q);
        for (int i = tid; i < 33 * 33; i += 256) {
            const int y = i / 33, x = i - y * 33;
            t0[i] = (y == 0 || x == 0) ? 0.f : tile[(y - 1) * 32 + (x - 1)];
        }
        if (tid < 33) {  // the front padding of conv1's output: row 0 and column 0
            const int pos = tid < 17 ? tid : (tid - 16) * 17;
#pragma unroll
            for (int i = 0; i < 4; ++i) a1img[pos * 4 + i] = em_u4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        // conv1: thread = output position (y, x) of the 16 x 16 map, all 16 channels
        float a[16];
        {
            const int y = tid >> 4, x = tid & 15;
            float in[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) in[ky * 3 + kx] = t0[(2 * y + ky) * 33 + 2 * x + kx];
            float m = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float s = p.c1b[c];
#pragma unroll
                for (int k = 0; k < 9; ++k) s = __builtin_fmaf(in[k], p.c1w[c * 9 + k], s);
                a[c] = leaky02(s);
                m = __builtin_fmaxf(m, __builtin_fabsf(a[c]));
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) m = __builtin_fmaxf(m, __shfl_xor(m, d));
            if (lane == 0) wmax[wv] = m;
        }
        __syncthreads();
        float inv1;
        {
            const float m = __builtin_fmaxf(__builtin_fmaxf(wmax[0], wmax[1]), __builtin_fmaxf(wmax[2], wmax[3]));
            float sc1;
            em_row_scale(m, sc1, inv1);
            const int y = tid >> 4, x = tid & 15;
            em_u4 h0, l0, h1, l1;
            em_split8(em_f4{a[0] * sc1, a[1] * sc1, a[2] * sc1, a[3] * sc1}, em_f4{a[4] * sc1, a[5] * sc1, a[6] * sc1, a[7] * sc1}, h0, l0);
            em_split8(em_f4{a[8] * sc1, a[9] * sc1, a[10] * sc1, a[11] * sc1}, em_f4{a[12] * sc1, a[13] * sc1, a[14] * sc1, a[15] * sc1}, h1, l1);
            em_u4* dst = a1img + ((y + 1) * 17 + (x + 1)) * 4;
            dst[0] = h0;
            dst[1] = h1;
            dst[2] = l0;
            dst[3] = l1;
        }
        __syncthreads();
        // conv2: [16 channels of tile mt] x [2 x 16 positions], K = 5 k-steps of (2 taps x 16 channels)
        em_f4 acc[2] = {em_f4{0.f, 0.f, 0.f, 0.f}, em_f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            int tap = 2 * s + (q >> 1);
            tap = tap > 8 ? 8 : tap;  // (the tenth tap: zero weights, any finite operand)
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int pos = 16 * (2 * (wv >> 1) + t) + n, y = pos >> 3, x = pos & 7;
                const em_u4* src = a1img + ((2 * y + ky) * 17 + 2 * x + kx) * 4 + (q & 1);
                const em_u4 bh = src[0], bl = src[2];
                acc[t] = em_mfma(wa[s][1], bh, acc[t]);
                acc[t] = em_mfma(wa[s][0], bl, acc[t]);
                acc[t] = em_mfma(wa[s][0], bh, acc[t]);
            }
        }
        const float u = inv1 * p.c2_winv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[r] = leaky02(__builtin_fmaf(acc[0][r], u, b2[r]));
            o1[r] = leaky02(__builtin_fmaf(acc[1][r], u, b2[r]));
        }
    }
    float m = em_absmax8(o0, o1);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) m = __builtin_fmaxf(m, __shfl_xor(m, d));
    if (lane == 0) wmax[4 + wv] = m;
    __syncthreads();
    m = __builtin_fmaxf(__builtin_fmaxf(wmax[4], wmax[5]), __builtin_fmaxf(wmax[6], wmax[7]));
    float sc, inv;
    em_row_scale(m, sc, inv);
    em_u4 hi, lo;
    em_split8(o0 * sc, o1 * sc, hi, lo);
    const int row = blockIdx.x, ks = 16 * wv + (lane >> 2), qq = lane & 3;
    em_u4* dst = feat + ((size_t)(row >> 4) * EM_C3_KSTEPS + ks) * 128 + qq * 16 + (row & 15);
    dst[0] = hi;
    dst[64] = lo;
    if (tid == 0) feat_inv[row] = inv;
}

}  // namespace msiren
