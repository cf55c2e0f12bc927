// RECORD ONLY -- not part of the product build since round 4 (profiles/r4/04_config5_x1n_vs_x1_ab.txt has the same-box A/B
// against siren_trunk_x1n.hip.h, which replaced it).  Round 1's single-product trunk on 32x32x16 MFMA tiles; it built against
// mri_inr_amd/csrc at commit 6afbb61 (weight stream [(L-1)*16 chunks][32 k-steps][64 lanes][8], fp16 bias table, l0last rows).
#pragma once
#include "../../mri_inr_amd/csrc/siren_trunk_x1.hip.h"

namespace msiren {

template <int R>
struct X1Lds {
    static constexpr int ring = 0;
    static constexpr int l0 = R * X1_CHUNK_BYTES;  // 32 x float4
    static constexpr int wout = l0 + 512;          // 512 x fp16
    static constexpr int zero = wout + 1024;       // 512 x fp16 zeros
    static constexpr int bias = zero + 1024;       // (L-1) x 512 x fp16
    static __host__ __device__ constexpr int mods(int L) { return bias + (L - 1) * 1024; }  // 4 waves x L x 512 x fp16
    static __host__ __device__ constexpr int queue(int L) { return mods(L) + 4 * L * 1024; }
    static __host__ __device__ constexpr int winv(int L) { return queue(L) + 16; }  // per-layer inverse weight scales
    static __host__ __device__ constexpr int total(int L) { return winv(L) + 256; }
};

template <int BF>
__device__ __forceinline__ void x1_mfma(f32x16& d, const u32x4& a, const u32x4& b, bool first) {
    f32x16 c = d;
    if (first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
    }
    if constexpr (BF)
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    else
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

template <int BF, int ACT, int RES, int R>
__global__ __launch_bounds__(256, 1) void siren_trunk_x1_kernel(TrunkX1Params p) {
    using LY = X1Lds<R>;
    constexpr int KS = 32;  // k-steps per layer (512 / 16)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int c32 = lane & 31;
    const int L = p.L;
    const int nchunks = (L - 1) * 16;

    // per-lane byte bases (feature offset 4*half folded in); fp16 tables: 2 bytes per feature
    const unsigned char* l0B = smem + LY::l0 + half * 64;
    const unsigned char* woutB = smem + LY::wout + half * 8;
    const unsigned char* zeroB = smem + LY::zero + half * 8;
    const unsigned char* biasB = smem + LY::bias + half * 8;
    _Float16* modT = reinterpret_cast<_Float16*>(smem + LY::mods(L)) + wave * (L * 512);
    const unsigned char* modB = reinterpret_cast<const unsigned char*>(modT) + half * 8;

    {   // constant tables
        if (tid < 32) reinterpret_cast<f32x4*>(smem + LY::l0)[tid] = reinterpret_cast<const f32x4*>(p.l0last)[tid];
        _Float16* wow = reinterpret_cast<_Float16*>(smem + LY::wout);
        _Float16* zw = reinterpret_cast<_Float16*>(smem + LY::zero);
        _Float16* bw = reinterpret_cast<_Float16*>(smem + LY::bias);
        for (int i = tid; i < 512; i += 256) {
            wow[i] = p.wout[i];
            zw[i] = (_Float16)0.f;
        }
        for (int i = tid; i < (L - 1) * 512; i += 256) bw[i] = p.bias[i];
    }

    volatile int* qslot = reinterpret_cast<volatile int*>(smem + LY::queue(L));
    // The per-layer inverse scales are indexed at run time.  Read straight from the kernel-argument segment
    // (host-visible memory) every such s_load that misses the scalar cache costs microseconds -- measured:
    // ~13k cycles at every layer boundary.  They are copied to LDS once instead.
    float* winvT = reinterpret_cast<float*>(smem + LY::winv(L));
    if (tid < 64) winvT[tid] = p.winv[tid];
    int cur_pass = (int)blockIdx.x;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp) + wave * 8192 + lane * 16 + 4096;
    int dma_id = 0, dma_buf = 0, rd_buf = 0;
    auto dma_next = [&]() {
        const unsigned char* src = wsrc + (size_t)dma_id * X1_CHUNK_BYTES;
        unsigned char* dst = smem + LY::ring + dma_buf * X1_CHUNK_BYTES + wave * 8192 + 4096;
#define MSIREN_DMA(I)                                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,                  \
                                     (__attribute__((address_space(3))) void*)dst, 16, (I) * 1024 - 4096, 0)
        MSIREN_DMA(0);
        MSIREN_DMA(1);
        MSIREN_DMA(2);
        MSIREN_DMA(3);
        MSIREN_DMA(4);
        MSIREN_DMA(5);
        MSIREN_DMA(6);
        MSIREN_DMA(7);
#undef MSIREN_DMA
        dma_id = dma_id + 1 == nchunks ? 0 : dma_id + 1;
        dma_buf = dma_buf + 1 == R ? 0 : dma_buf + 1;
    };
    // the kept patches only: known on the device (scalar: a vector register here costs the co-residency of §4.3)
    const int total_units = __builtin_amdgcn_readfirstlane(p.plan ? p.plan[1] : p.total_units);
    // unsigned compare: a pass id that came out negative (host/device counter disagreement) ends the workgroup
    // instead of indexing out of bounds
    const unsigned npasses = (unsigned)(total_units + 3) >> 2;
    if ((unsigned)cur_pass >= npasses) return;
#pragma unroll
    for (int s = 0; s < R - 1; ++s) dma_next();

    u32x4 X[KS], Y[KS];
    {   // the first "pending" slot multiplies Y[30], Y[31] by a residual flag of 0: keep them finite
        u32x4 z = {0u, 0u, 0u, 0u};
        Y[30] = x1_to_acc_file(z);
        Y[31] = x1_to_acc_file(z);
    }
    f32x16 acc[2];
    float part = 0.f;

    unsigned ew[4][2];  // packed output pairs of the tile being finished: [part][half]
    hf4 tb_b[2], tb_m[2], tb_w[2];
    auto tbl_load = [&](int set, const unsigned char* bl, const unsigned char* ml, const unsigned char* wo, int t, int g,
                        bool withw) {
        const int fo = (32 * t + 8 * g) * 2;  // compile-time byte offset (fp16)
        tb_b[set] = *reinterpret_cast<const hf4*>(bl + fo);
        tb_m[set] = *reinterpret_cast<const hf4*>(ml + fo);
        if (withw) tb_w[set] = *reinterpret_cast<const hf4*>(wo + fo);
    };
    // half `hh` (elements 2hh, 2hh+1) of part g; `old` = the fragment that holds the same features of the
    // layer's INPUT (word 2(g&1)+hh), `rflag` = 1 where the residual applies (0 for the layer-0 tile)
    // lastl = final hidden layer (own code instances): accumulate last_layer's dot product, produce no fragment
    // `fresh`: the accumulator was written by the MFMAs just before ("pending" tile of the previous layer): no asm anchor
    // in between, so that hipcc sees the MFMA -> VALU read and pads the hazard (an asm statement hides the producer).
    auto epi_half = [&](const f32x16& a, float winv, float cgl, int set, int g, int hh, const u32x4& old, float rflag,
                        bool lastl, bool fresh = false) {
        float a0 = a[4 * g + 2 * hh], a1 = a[4 * g + 2 * hh + 1];
        if (!fresh) asm volatile("; epilogue slice anchored to its MFMA group" : "+v"(a0), "+v"(a1));
        const float ain[2] = {a0, a1};
        float xo[2] = {0.f, 0.f};
        if constexpr (RES) x1_unpack2<BF>(old[2 * (g & 1) + hh], xo[0], xo[1]);
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float r = __builtin_fmaf(ain[e], winv, (float)tb_b[set][2 * hh + e]);
            v[e] = activate<ACT>(r, cgl) * (float)tb_m[set][2 * hh + e];
            if constexpr (RES) v[e] = __builtin_fmaf(xo[e], rflag, v[e]);
            if (lastl) part = __builtin_fmaf(v[e], (float)tb_w[set][2 * hh + e], part);
        }
        if (!lastl) ew[g][hh] = x1_pack2<BF>(v[0], v[1]);
    };
    auto epi_store1 = [&](int ks, u32x4& d) {  // k-step ks (0/1) of the tile = parts 2ks, 2ks+1
        u32x4 u;
        u[0] = ew[2 * ks][0];
        u[1] = ew[2 * ks][1];
        u[2] = ew[2 * ks + 1][0];
        u[3] = ew[2 * ks + 1][1];
        d = x1_to_acc_file(u);
    };
    u32x4 wf_[2][4];

#define MSIREN_X1_GROUP(IN, OUT, T, Q, LASTF)                                                  \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        {                                                                                     \
            const u32x4* src_ = (Q) < 7 ? ring_ + (4 * (((Q) + 1) & 7)) * 64 : ringn_;        \
            wf_[((Q) + 1) & 1][0] = src_[0 * 64];                                             \
            wf_[((Q) + 1) & 1][1] = src_[1 * 64];                                             \
            wf_[((Q) + 1) & 1][2] = src_[2 * 64];                                             \
            wf_[((Q) + 1) & 1][3] = src_[3 * 64];                                             \
        }                                                                                     \
        if ((T) == 0) {                                                                       \
            /* pending tile 15 of the previous layer: its input fragments are OUT[30], OUT[31] */ \
            if ((Q) < 3) tbl_load(((Q) + 1) & 1, blp_, mlp_, zeroB, 15, ((Q) + 1) & 3, false); \
            if ((Q) < 4) {                                                                    \
                epi_half(acc[1], wip_, cgp_, (Q) & 1, (Q) & 3, 0, OUT[30 + (((Q) & 3) >> 1)], rfp_, false, true); \
                epi_half(acc[1], wip_, cgp_, (Q) & 1, (Q) & 3, 1, OUT[30 + (((Q) & 3) >> 1)], rfp_, false, true); \
            }                                                                                 \
            if ((Q) == 4) epi_store1(0, IN[30]);                                              \
            if ((Q) == 5) epi_store1(1, IN[31]);                                              \
        } else {                                                                              \
            if (((Q) & 1) == 1 && (Q) < 7) tbl_load((((Q) >> 1) + 1) & 1, bl_, ml_, wo_, ((T) + 15) & 15, (((Q) >> 1) + 1) & 3, LASTF); \
            epi_half(acc[((T) + 1) & 1], wi_, p.cg, ((Q) >> 1) & 1, (Q) >> 1, (Q) & 1,       \
                     IN[(2 * (T) + 30 + ((Q) >> 2)) & 31], 1.0f, LASTF);                      \
            if ((Q) == 5 && !(LASTF)) epi_store1(0, OUT[(2 * (T) + 30) & 31]);                \
        }                                                                                     \
        if ((Q) == 7) tbl_load(0, bl_, ml_, wo_, (T), 0, LASTF);                              \
        x1_mfma<BF>(acc[(T) & 1], wf_[(Q) & 1][0], IN[4 * (Q) + 0], (Q) == 0);                \
        x1_mfma<BF>(acc[(T) & 1], wf_[(Q) & 1][1], IN[4 * (Q) + 1], false);                   \
        x1_mfma<BF>(acc[(T) & 1], wf_[(Q) & 1][2], IN[4 * (Q) + 2], false);                   \
        x1_mfma<BF>(acc[(T) & 1], wf_[(Q) & 1][3], IN[4 * (Q) + 3], false);                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                    \
    } while (0)

#define MSIREN_X1_TILE(IN, OUT, T, LASTF)                                                      \
    do {                                                                                      \
        const u32x4* ring_ = reinterpret_cast<const u32x4*>(smem + LY::ring + rd_buf * X1_CHUNK_BYTES) + lane; \
        rd_buf = rd_buf + 1 == R ? 0 : rd_buf + 1;                                            \
        const u32x4* ringn_ = reinterpret_cast<const u32x4*>(smem + LY::ring + rd_buf * X1_CHUNK_BYTES) + lane; \
        MSIREN_X1_GROUP(IN, OUT, T, 0, LASTF);                                                  \
        MSIREN_X1_GROUP(IN, OUT, T, 1, LASTF);                                                  \
        MSIREN_X1_GROUP(IN, OUT, T, 2, LASTF);                                                  \
        MSIREN_X1_GROUP(IN, OUT, T, 3, LASTF);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 8) : "memory");                    \
        __builtin_amdgcn_s_barrier();                                                         \
        dma_next();                                                                           \
        MSIREN_X1_GROUP(IN, OUT, T, 4, LASTF);                                                  \
        MSIREN_X1_GROUP(IN, OUT, T, 5, LASTF);                                                  \
        MSIREN_X1_GROUP(IN, OUT, T, 6, LASTF);                                                  \
        MSIREN_X1_GROUP(IN, OUT, T, 7, LASTF);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if ((T) > 0 && !(LASTF)) epi_store1(1, OUT[(2 * (T) + 31) & 31]);                     \
    } while (0)

// final hidden layer's last tile: only `part` matters; its residual input sits in the array that layer read (IN)
#define MSIREN_X1_FINAL(IN)                                                                          \
    {                                                                                                \
        MSIREN_X1_FINAL_G(IN, 0);                                                                    \
        MSIREN_X1_FINAL_G(IN, 1);                                                                    \
        MSIREN_X1_FINAL_G(IN, 2);                                                                    \
        MSIREN_X1_FINAL_G(IN, 3);                                                                    \
    }
#define MSIREN_X1_FINAL_G(IN, G)                                                                     \
    {                                                                                                \
        if ((G) > 0) tbl_load((G) & 1, biasB + (L - 2) * 1024, modB + (L - 1) * 1024, woutB, 15, (G), true); \
        epi_half(acc[1], winvT[L - 2], p.cg, (G) & 1, (G), 0, IN[30 + ((G) >> 1)], 1.0f, true, true); \
        epi_half(acc[1], winvT[L - 2], p.cg, (G) & 1, (G), 1, IN[30 + ((G) >> 1)], 1.0f, true, true); \
    }

#define MSIREN_X1_LAYER(IN, OUT, LIDX, LASTF)                                                  \
    do {                                                                                      \
        const int l_ = (LIDX);                                                                \
        const unsigned char* wo_ = woutB; /* read by the final-layer instances only */        \
        const unsigned char* bl_ = biasB + (l_ - 1) * 1024;                                   \
        const unsigned char* ml_ = modB + l_ * 1024;                                          \
        const unsigned char* blp_ = l_ > 1 ? biasB + (l_ - 2) * 1024 : zeroB;                 \
        const unsigned char* mlp_ = modB + (l_ - 1) * 1024;                                   \
        const float wi_ = winvT[l_ - 1], wip_ = l_ > 1 ? winvT[l_ - 2] : 1.0f;              \
        const float cgp_ = l_ > 1 ? p.cg : p.cg0;                                             \
        const float rfp_ = l_ > 1 ? 1.0f : 0.0f; /* layer 0 has no skip connection */         \
        MSIREN_X1_TILE(IN, OUT, 0, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 1, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 2, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 3, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 4, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 5, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 6, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 7, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 8, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 9, LASTF);                                                           \
        MSIREN_X1_TILE(IN, OUT, 10, LASTF);                                                          \
        MSIREN_X1_TILE(IN, OUT, 11, LASTF);                                                          \
        MSIREN_X1_TILE(IN, OUT, 12, LASTF);                                                          \
        MSIREN_X1_TILE(IN, OUT, 13, LASTF);                                                          \
        MSIREN_X1_TILE(IN, OUT, 14, LASTF);                                                          \
        MSIREN_X1_TILE(IN, OUT, 15, LASTF);                                                          \
    } while (0)

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 8) : "memory");
    __syncthreads();
    {
        const u32x4* r0 = reinterpret_cast<const u32x4*>(smem + LY::ring) + lane;
        wf_[0][0] = r0[0 * 64];
        wf_[0][1] = r0[1 * 64];
        wf_[0][2] = r0[2 * 64];
        wf_[0][3] = r0[3 * 64];
    }

    for (int pass = 0; (unsigned)cur_pass < npasses; ++pass) {
        int unit = cur_pass * 4 + wave;
        const bool active = unit < total_units;
        unit = active ? unit : total_units - 1;
        const int b = unit / p.units_per_patch;
        const int cu = unit - b * p.units_per_patch;
        int pc = cu * 32 + c32;
        const bool pvalid = active && pc < p.P;
        pc = pc < p.P ? pc : p.P - 1;

        int nxt = 0;
        if (tid == 0) nxt = (int)((unsigned)atomicAdd(p.pass_counter, 1) - p.pass_base) + (int)gridDim.x;
        // this wave's modulation table: (L, 512) of patch b, narrowed to fp16
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(p.mods + ((size_t)l * p.B + b) * 512 + i * 256 + lane * 4);
                hf4 hm;
                hm[0] = (_Float16)m[0];
                hm[1] = (_Float16)m[1];
                hm[2] = (_Float16)m[2];
                hm[3] = (_Float16)m[3];
                *reinterpret_cast<hf4*>(modT + l * 512 + i * 256 + lane * 4) = hm;
            }
        if (tid == 0) qslot[(pass + 1) & 1] = nxt;
        const float2 xy = reinterpret_cast<const float2*>(p.grid)[pc];

        // layer 0 from the activation table (k-steps 0..29), in two batches of 15 k-steps
        const f32x4* s0 = reinterpret_cast<const f32x4*>(p.s0t) + (size_t)half * p.P + pc;
#pragma unroll
        for (int sb = 0; sb < 30; sb += 15) {
            f32x4 raw[15][2];
#pragma unroll
            for (int s = 0; s < 15; ++s)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f0 = 32 * ((sb + s) >> 1) + 16 * ((sb + s) & 1) + 8 * q;
                    raw[s][q] = s0[(size_t)(f0 / 4) * p.P];
                }
#pragma unroll
            for (int s = 0; s < 15; ++s) {
                u32x4 u;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f0 = 32 * ((sb + s) >> 1) + 16 * ((sb + s) & 1) + 8 * q;
                    const hf4 m4 = *reinterpret_cast<const hf4*>(modB + f0 * 2);
                    u[2 * q] = x1_pack2<BF>(raw[s][q][0] * (float)m4[0], raw[s][q][1] * (float)m4[1]);
                    u[2 * q + 1] = x1_pack2<BF>(raw[s][q][2] * (float)m4[2], raw[s][q][3] * (float)m4[3]);
                }
                X[sb + s] = x1_to_acc_file(u);
            }
        }
        {   // sine arguments of the last 32 features -> the first hidden layer's pending-epilogue slot
            f32x16 r7;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(l0B + (8 * g + e) * 16);
                    r7[4 * g + e] = __builtin_fmaf(xy.y, w[1], __builtin_fmaf(xy.x, w[0], w[2]));
                }
            acc[1] = r7;
        }
        tbl_load(0, zeroB, modB, zeroB, 15, 0, false);

        part = 0.f;
        // hidden layers alternate X->Y / Y->X; the final hidden layer has its own instances (one per input array)
        for (int l = 1;;) {
            if (l == L - 1) {
                MSIREN_X1_LAYER(X, Y, l, true);
                MSIREN_X1_FINAL(X);
                break;
            }
            MSIREN_X1_LAYER(X, Y, l, false);
            ++l;
            if (l == L - 1) {
                MSIREN_X1_LAYER(Y, X, l, true);
                MSIREN_X1_FINAL(Y);
                break;
            }
            MSIREN_X1_LAYER(Y, X, l, false);
            ++l;
        }
        part += __shfl_xor(part, 32);
        if (pvalid && half == 0) p.out[(size_t)b * p.P + pc] = sin_rev(part + p.bout);
        cur_pass = __builtin_amdgcn_readfirstlane(qslot[(pass + 1) & 1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef MSIREN_X1_LAYER
#undef MSIREN_X1_TILE
#undef MSIREN_X1_GROUP
}

}  // namespace msiren
