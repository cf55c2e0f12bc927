// Fused modulated-SIREN trunk for gfx950, split-fp16 path ("f16x3"): fp32-equivalent accuracy at
// 3/16 of the fp32-MFMA cost.
//
// Arithmetic.  Every hidden-layer operand is split into two fp16 numbers, v = hi + lo exactly-ish
// (hi = f16(v), lo = f16(v - hi); 22 significant bits), and the product is evaluated as
//     W*x  ~=  W_lo*x_hi + W_hi*x_lo + W_hi*x_hi        (the lo*lo term, 2^-22 relative, is dropped)
// with three v_mfma_f32_32x32x16_f16 accumulating in fp32.  Weights are split once on the host after
// scaling by a power of two (so that W_lo stays clear of the fp16 subnormal range; the exact inverse
// scale is folded into the epilogue FMA); activations are split in the epilogue.  gfx950's MFMA keeps
// fp16 subnormal inputs (tools/f16_probe.hip), so x_lo needs no scaling.  Measured against the fp64
// oracle this path is indistinguishable from true fp32 (tests/test_gpu_parity.py).
//
// Data flow (reference: src/networks/modulated_siren.py:215-233).
//   workgroup = 4 waves, ONE per SIMD (the kernel owns the whole 512-register file); persistent grid,
//   one workgroup per CU.  A wave evaluates one UNIT = 32 coordinates of one patch through all layers
//   with its activations in REGISTERS: the fp32 accumulator tile of layer l (features on registers,
//   coordinate on the lane) is, after the epilogue, bit for bit the B operand of layer l+1 -- no LDS
//   round trip, no barrier for activations (the k order inside a step is permuted accordingly and the
//   host packs the weights in that order).
//   The weights are the operand every wave shares: the stream of 32 KB chunks
//   [layer][feature tile][k-step][hi|lo][lane][8 x f16] is DMA'd (global_load_lds_dwordx4) into a ring
//   in LDS, two to three chunks ahead, and read back with one conflict-free ds_read_b128 per MFMA.
//   One s_barrier per chunk (48 MFMAs) orders ring reuse.  The epilogue of tile t (VALU: FMA, v_sin,
//   modulation, fp16 split) is issued in the same scheduling region as the MFMAs of tile t+1.
#pragma once
#include <hip/hip_runtime.h>

#include "siren_trunk_f32.hip.h"

namespace msiren {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct TrunkF16Params {
    const float* grid;        // (P,2)
    const float* l0;          // (256,4) {w_row, w_col, b, 0} * w0_initial/2pi
    const _Float16* wp;       // [(L-1)*8 chunks][16 k-steps][2: hi,lo][64 lanes][8]
    const float* bias;        // (L-1, 256) in revolutions
    const float* wout;        // (256) * w0/2pi
    const float* mods;        // (L, B, 256)
    float* out;               // (B, P)
    float winv[16];           // per hidden layer: exact inverse of the power-of-two weight scale
    float bout, cg0, cg;
    int B, P, L, units_per_patch, total_units;
};

constexpr int F16_CHUNK_BYTES = 32768;

template <int R>
struct F16Lds {  // byte offsets into dynamic LDS
    static constexpr int ring = 0;
    static constexpr int l0 = R * F16_CHUNK_BYTES;  // 256 x float4
    static constexpr int wout = l0 + 4096;          // 256 floats
    static constexpr int zero = wout + 1024;        // 256 floats of 0 (stand-in for wout on non-final layers)
    static constexpr int bias = zero + 1024;        // (L-1) x 256 floats
    static __host__ __device__ constexpr int mods(int L) { return bias + (L - 1) * 1024; }  // 4 waves x L x 256 floats
    static __host__ __device__ constexpr int total(int L) { return mods(L) + 4 * L * 1024; }
};

__device__ __forceinline__ h8 pack_h8(fp16x2 a, fp16x2 b, fp16x2 c, fp16x2 d) {
    u32x4 u;
    u[0] = __builtin_bit_cast(unsigned, a);
    u[1] = __builtin_bit_cast(unsigned, b);
    u[2] = __builtin_bit_cast(unsigned, c);
    u[3] = __builtin_bit_cast(unsigned, d);
    return __builtin_bit_cast(h8, u);
}

// v (fp32) -> hi, lo (fp16, round toward zero; lo absorbs hi's truncation error exactly)
__device__ __forceinline__ void split4(const f32x4 v, fp16x2& h01, fp16x2& h23, fp16x2& l01, fp16x2& l23) {
    h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
    h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
    l01 = __builtin_amdgcn_cvt_pkrtz(v[0] - (float)h01[0], v[1] - (float)h01[1]);
    l23 = __builtin_amdgcn_cvt_pkrtz(v[2] - (float)h23[0], v[3] - (float)h23[1]);
}

template <int ACT, int R>
__global__ __launch_bounds__(256, 1) void siren_trunk_f16x3_kernel(TrunkF16Params p) {
    using LY = F16Lds<R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int c32 = lane & 31;
    const int L = p.L;
    const int nchunks = (L - 1) * 8;

    const f32x4* l0T = reinterpret_cast<const f32x4*>(smem + LY::l0);
    const float* woutT = reinterpret_cast<const float*>(smem + LY::wout);
    const float* zeroT = reinterpret_cast<const float*>(smem + LY::zero);
    const float* biasT = reinterpret_cast<const float*>(smem + LY::bias);
    float* modT = reinterpret_cast<float*>(smem + LY::mods(L)) + wave * (L * 256);

    // ---- once per workgroup: constant tables ------------------------------------------------------
    {
        f32x4* l0w = reinterpret_cast<f32x4*>(smem + LY::l0);
        float* wow = reinterpret_cast<float*>(smem + LY::wout);
        float* zw = reinterpret_cast<float*>(smem + LY::zero);
        float* bw = reinterpret_cast<float*>(smem + LY::bias);
        l0w[tid] = reinterpret_cast<const f32x4*>(p.l0)[tid];
        wow[tid] = p.wout[tid];
        zw[tid] = 0.f;
        for (int i = tid; i < (L - 1) * 256; i += 256) bw[i] = p.bias[i];
    }

    // ---- weight ring ---------------------------------------------------------------------------------
    // chunk sequence number `cg` counts tiles consumed by this workgroup; chunk id = cg mod nchunks.
    const int G = gridDim.x;
    const int upp = 4 * G;  // units per pass over the whole grid
    const int npass = (p.total_units - 4 * (int)blockIdx.x + upp - 1) / upp;  // passes of THIS workgroup (>= 0)
    const int total_chunks = npass * nchunks;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp) + wave * 8192 + lane * 16;
    auto dma_chunk = [&](int seq) {  // every wave moves its 8 KB slice of chunk `seq`
        if (seq < total_chunks) {
            const int id = seq % nchunks;
            const int buf = seq % R;
            const unsigned char* src = wsrc + (size_t)id * F16_CHUNK_BYTES;
            unsigned char* dst = smem + LY::ring + buf * F16_CHUNK_BYTES + wave * 8192;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
    };
    if (npass <= 0) return;
#pragma unroll
    for (int s = 0; s < R - 1; ++s) dma_chunk(s);
    int cg = 0;

    h8 Xh[16], Xl[16], Yh[16], Yl[16];
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;  // the first "pending" epilogue reads acc[1]
    float part = 0.f;

    // epilogue of one 32-feature tile, in four parts (g = 0..3: features 8g..8g+7 of the tile, 4 per
    // half-wave): acc -> (revolutions) -> activation -> modulation -> fp16 split; parts 0,1 make up
    // k-step 2t and parts 2,3 k-step 2t+1 of the next layer's B operand.  Also accumulates last_layer's
    // dot product with `wo` (the zero table on all but the final hidden layer).
    fp16x2 eh[4][2], el[4][2];
    auto epi_part = [&](const f32x16& a, int l, int t, int g, const float* wo) {
        const float winv = p.winv[l - 1];
        const int fo = 32 * t + 8 * g + 4 * half;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(biasT + (l - 1) * 256 + fo);
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(modT + l * 256 + fo);
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wo + fo);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r = __builtin_fmaf(a[4 * g + e], winv, b4[e]);
            v[e] = activate<ACT>(r, p.cg) * m4[e];
            part = __builtin_fmaf(v[e], w4[e], part);
        }
        split4(v, eh[g][0], eh[g][1], el[g][0], el[g][1]);
    };
    auto epi_store = [&](bool write, h8& dh0, h8& dl0, h8& dh1, h8& dl1) {
        const h8 nh0 = pack_h8(eh[0][0], eh[0][1], eh[1][0], eh[1][1]);
        const h8 nl0 = pack_h8(el[0][0], el[0][1], el[1][0], el[1][1]);
        const h8 nh1 = pack_h8(eh[2][0], eh[2][1], eh[3][0], eh[3][1]);
        const h8 nl1 = pack_h8(el[2][0], el[2][1], el[3][0], el[3][1]);
        dh0 = write ? nh0 : dh0;
        dl0 = write ? nl0 : dl0;
        dh1 = write ? nh1 : dh1;
        dl1 = write ? nl1 : dl1;
    };

    // one hidden layer: IN -> OUT.  `pend` = the previous hidden layer's last tile still sits in acc[1]
    // and its epilogue (which produces IN[14], IN[15]) is issued inside this layer's first tile.
    // A tile = 4 groups of 4 k-steps (12 MFMAs); each group's scheduling region also holds the LDS
    // reads of the next group's weight fragments and one quarter of the previous tile's epilogue.
#define MSIREN_F16_LAYER(INh, INl, OUTh, OUTl, LIDX, PEND)                                                    \
    do {                                                                                                      \
        const int l_ = (LIDX);                                                                                \
        const float* wo_ = (l_ == L - 1) ? woutT : zeroT;                                                     \
        _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                                       \
            dma_chunk(cg + R - 1);                                                                            \
            const h8* ring_ = reinterpret_cast<const h8*>(smem + LY::ring + (cg % R) * F16_CHUNK_BYTES) + lane; \
            f32x16 a_;                                                                                        \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) a_[r] = 0.f;                                       \
            h8 wf_[2][8];                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) wf_[0][i] = ring_[i * 64];                          \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                            \
                if (q < 3) {                                                                                  \
                    _Pragma("unroll") for (int i = 0; i < 8; ++i) wf_[(q + 1) & 1][i] = ring_[(8 * (q + 1) + i) * 64]; \
                }                                                                                             \
                if (t == 0) {                                                                                 \
                    /* pending tile of the previous layer feeds k-steps 14, 15 of THIS tile (group 3): */    \
                    /* all four parts in groups 0-1, stored in group 2 */                                     \
                    if (q < 2) {                                                                              \
                        epi_part(acc[1], l_ > 1 ? l_ - 1 : 1, 7, 2 * q, zeroT);                               \
                        epi_part(acc[1], l_ > 1 ? l_ - 1 : 1, 7, 2 * q + 1, zeroT);                           \
                    }                                                                                         \
                    if (q == 2) epi_store((PEND), INh[14], INl[14], INh[15], INl[15]);                        \
                } else {                                                                                      \
                    epi_part(acc[(t - 1) & 1], l_, t - 1, q, wo_);                                            \
                }                                                                                             \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
                    const int s = 4 * q + j;                                                                  \
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf_[q & 1][2 * j + 1], INh[s], a_, 0, 0, 0);  \
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf_[q & 1][2 * j], INl[s], a_, 0, 0, 0);      \
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf_[q & 1][2 * j], INh[s], a_, 0, 0, 0);      \
                }                                                                                             \
            }                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            if (t > 0) epi_store(true, OUTh[2 * t - 2], OUTl[2 * t - 2], OUTh[2 * t - 1], OUTl[2 * t - 1]);   \
            acc[t & 1] = a_;                                                                                  \
            ++cg;                                                                                             \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 8) : "memory");                                \
            __builtin_amdgcn_s_barrier();                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                \
        }                                                                                                     \
    } while (0)

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 8) : "memory");
    __syncthreads();  // tables + first chunk visible

    for (int pass = 0; pass < npass; ++pass) {
        int unit = (pass * G + (int)blockIdx.x) * 4 + wave;
        const bool active = unit < p.total_units;
        unit = active ? unit : p.total_units - 1;
        const int b = unit / p.units_per_patch;
        const int cu = unit - b * p.units_per_patch;
        int pc = cu * 32 + c32;
        const bool pvalid = active && pc < p.P;
        pc = pc < p.P ? pc : p.P - 1;

        // this wave's modulation table: (L, 256) floats of patch b
        for (int l = 0; l < L; ++l) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(p.mods + ((size_t)l * p.B + b) * 256 + lane * 4);
            *reinterpret_cast<f32x4*>(modT + l * 256 + lane * 4) = m;
        }
        const float2 xy = reinterpret_cast<const float2*>(p.grid)[pc];

        // ---- layer 0 (K = 2) directly in B-operand order: element j of k-step s is feature
        //      32*(s>>1) + 16*(s&1) + 8*(j>>2) + 4*half + (j&3)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            fp16x2 hh[2][2], ll[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int f0 = 32 * (s >> 1) + 16 * (s & 1) + 8 * q + 4 * half;
                const f32x4 m4 = *reinterpret_cast<const f32x4*>(modT + f0);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 w = l0T[f0 + e];
                    const float r = __builtin_fmaf(xy.y, w[1], __builtin_fmaf(xy.x, w[0], w[2]));
                    v[e] = activate<ACT>(r, p.cg0) * m4[e];
                }
                split4(v, hh[q][0], hh[q][1], ll[q][0], ll[q][1]);
            }
            Xh[s] = pack_h8(hh[0][0], hh[0][1], hh[1][0], hh[1][1]);
            Xl[s] = pack_h8(ll[0][0], ll[0][1], ll[1][0], ll[1][1]);
        }

        part = 0.f;
        for (int l = 1; l < L; l += 2) {
            MSIREN_F16_LAYER(Xh, Xl, Yh, Yl, l, l > 1);
            if (l + 1 < L) MSIREN_F16_LAYER(Yh, Yl, Xh, Xl, l + 1, true);
        }
        // the final hidden layer's last tile is still pending: only its contribution to `part` matters
#pragma unroll
        for (int g = 0; g < 4; ++g) epi_part(acc[1], L - 1, 7, g, woutT);
        part += __shfl_xor(part, 32);
        if (pvalid && half == 0) p.out[(size_t)b * p.P + pc] = sin_rev(part + p.bout);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may be in flight when the LDS is released
#undef MSIREN_F16_LAYER
}

}  // namespace msiren
