// Register-resident fused trunk, single-product 16-bit variant on 16x16x32 MFMA tiles ("x1n"): H = 512, bf16 (or fp16)
// operands, fp32 accumulation, optional residual connections -- BASELINE config 5.  Semantics as siren_trunk_x1.hip.h
// (this build's own residual definition, PARITY UNPINNED against the reference: no source of that branch in the container):
//     x_{l+1} = x_l + mod_l * act(W_l x_l + b_l)   for l >= 1      (layer 0 and last_layer unchanged)
//
// What siren_trunk_f16x3n.hip.h did for the split-fp16 trunk in round 2, done for the single-product one (round 4):
//   * v_mfma_f32_16x16x32_{bf16,f16} instead of 32x32x16: under the board power limit the chip sustains ~1.2x the FLOP/s on
//     this shape at the same bytes per FLOP (1.78-1.81 vs 1.48 PFLOP/s on the bare MFMA streams, DESIGN.md sections 4.2, 8);
//   * a thinner epilogue: the bias (fp32, in revolutions) enters as the C operand of a tile's first MFMAs, so the
//     accumulator IS the sine argument (bf16 weights carry no power-of-two scale; the fp16 instance keeps one multiply);
//     activation x modulation + residual is ONE v_fma_mix (fp16 modulation table read as the mixed operand).
// Data flow unchanged: a wave owns one UNIT = 32 coordinates of one patch (two column groups of 16) through all layers with
// its activations in AGPRs -- B fragment [2 s + g], element j = feature 32 s + 16 (j >> 2) + 4 q + (j & 3) at the lane's
// coordinate of group g: the four accumulator registers of the two 16-feature sub-tiles of a 32-feature tile, so a tile's
// epilogue output is the next layer's B operand and, for the residual, sits at the same place as the layer's input -- the
// weights stream as 32 KB chunks [16 k-steps][2 sub-tiles][64 lanes][8 x 16 bit] (one 32-feature tile) through an LDS ring
// by DMA, one fragment feeding the MFMAs of both column groups; persistent grid, pass queue.
#pragma once
#include <hip/hip_runtime.h>

#include "siren_trunk_f16x3n.hip.h"  // sum_over_q
#include "siren_trunk_x1.hip.h"      // TrunkX1Params, x1_pack2 / x1_unpack2 / x1_to_acc_file, vector types

namespace msiren {

template <int R>
struct X1nLds {
    static constexpr int ring = 0;
    static constexpr int wout = R * X1_CHUNK_BYTES;  // 512 x fp16
    static constexpr int bias = wout + 1024;         // (L-1) x 512 x fp32 (C operand)
    static __host__ __device__ constexpr int mods(int L) { return bias + (L - 1) * 2048; }  // 4 waves x L x 512 x fp16
    static __host__ __device__ constexpr int queue(int L) { return mods(L) + 4 * L * 1024; }
    static __host__ __device__ constexpr int winv(int L) { return queue(L) + 16; }  // per-layer inverse weight scales (fp16 instance)
    static __host__ __device__ constexpr int total(int L) { return winv(L) + 256; }
};

template <int BF>
__device__ __forceinline__ void x1n_mfma(f32x4& d, const u32x4& a, const u32x4& b, const f32x4& c) {
    if constexpr (BF)
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    else
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

template <int BF, int ACT, int RES, int R>
__global__ __launch_bounds__(256, 1) void siren_trunk_x1n_kernel(TrunkX1Params p) {
    using LY = X1nLds<R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;     // which four features of a 16-feature sub-tile this lane holds
    const int n16 = lane & 15;   // coordinate inside a 16-column group
    const int L = p.L;
    const int nchunks = (L - 1) * 16;

    // per-lane byte bases of the LDS tables (feature offset 4 q folded in)
    const unsigned char* woutB = smem + LY::wout + q * 8;   // fp16 per feature
    const unsigned char* biasB = smem + LY::bias + q * 16;  // fp32 per feature
    _Float16* modT = reinterpret_cast<_Float16*>(smem + LY::mods(L)) + wave * (L * 512);
    const unsigned char* modB = reinterpret_cast<const unsigned char*>(modT) + q * 8;

    {   // constant tables
        _Float16* wow = reinterpret_cast<_Float16*>(smem + LY::wout);
        float* bw = reinterpret_cast<float*>(smem + LY::bias);
        for (int i = tid; i < 512; i += 256) wow[i] = p.wout[i];
        for (int i = tid; i < (L - 1) * 512; i += 256) bw[i] = p.bias32[i];
    }
    volatile int* qslot = reinterpret_cast<volatile int*>(smem + LY::queue(L));
    float* winvT = reinterpret_cast<float*>(smem + LY::winv(L));  // (kernel-argument reads at run-time indices cost microseconds)
    if (tid < 64) winvT[tid] = p.winv[tid];

    int cur_pass = (int)blockIdx.x;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp) + wave * 8192 + lane * 16 + 4096;
    int dma_id = 0, dma_buf = 0, rd_buf = 0;
    const unsigned char* dsrc_ = wsrc;
    unsigned char* ddst_ = smem + LY::ring + wave * 8192 + 4096;
    auto dma_begin = [&]() {
        dsrc_ = wsrc + (size_t)dma_id * X1_CHUNK_BYTES;
        ddst_ = smem + LY::ring + dma_buf * X1_CHUNK_BYTES + wave * 8192 + 4096;
        dma_id = dma_id + 1 == nchunks ? 0 : dma_id + 1;
        dma_buf = dma_buf + 1 == R ? 0 : dma_buf + 1;
    };
#define MSIREN_X1N_DMA(I)                                                                                 \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)dsrc_,                \
                                     (__attribute__((address_space(3))) void*)ddst_, 16, (I) * 1024 - 4096, 0)
    auto dma_next = [&]() {
        dma_begin();
        MSIREN_X1N_DMA(0);
        MSIREN_X1N_DMA(1);
        MSIREN_X1N_DMA(2);
        MSIREN_X1N_DMA(3);
        MSIREN_X1N_DMA(4);
        MSIREN_X1N_DMA(5);
        MSIREN_X1N_DMA(6);
        MSIREN_X1N_DMA(7);
    };
    const int total_units = __builtin_amdgcn_readfirstlane(p.plan ? p.plan[1] : p.total_units);
    const unsigned npasses = (unsigned)(total_units + 3) >> 2;  // (unsigned compare: a negative pass id ends the workgroup)
    if ((unsigned)cur_pass >= npasses) return;
#pragma unroll
    for (int s = 0; s < R - 1; ++s) dma_next();

    u32x4 X[32], Y[32];  // B fragments [2 * k-step + column group], in the accumulator half of the register file
    {   // the first "pending" slot multiplies Y[30], Y[31] by a residual flag of 0: keep them finite
        const u32x4 z = {0u, 0u, 0u, 0u};
        Y[30] = x1_to_acc_file(z);
        Y[31] = x1_to_acc_file(z);
    }
    f32x4 acc[2][4];     // [tile parity][part], part = 2 * column group + sub-tile
    float part[2] = {0.f, 0.f};  // last_layer dot product per column group

    unsigned ew[4][2];   // packed output pairs of the tile being finished: [part][half]
    hf4 tb_m[2], tb_w[2];
    f32x4 bia[2];
    auto tbl_load = [&](int sub, const unsigned char* ml, const unsigned char* wo, int t, bool withw) {
        const int fo = (32 * t + 16 * sub) * 2;  // compile-time byte offset (fp16)
        tb_m[sub] = *reinterpret_cast<const hf4*>(ml + fo);
        if (withw) tb_w[sub] = *reinterpret_cast<const hf4*>(wo + fo);
    };
    auto bias_load = [&](int sub, const unsigned char* bl, int t) {
        bia[sub] = *reinterpret_cast<const f32x4*>(bl + (32 * t + 16 * sub) * 4);
        asm("; bias fragment stays in arch VGPRs" : "+v"(bia[sub]));
    };
    // half `hh` (elements 2hh, 2hh+1) of part pt = 2 g + sub of the tile whose accumulators are `a`.
    //   old    the fragment that holds the same features of the layer's INPUT (word 2 sub + hh): the residual
    //   rf     1 where the residual applies, 0 for the layer-0 tile (a run-time value only in the "pending" slot)
    //   steady the accumulator is a tile old and the residual applies: sine anchored through asm, one fused multiply-add
    //   ready  the values are layer 0's last 32 features straight from the activation table (already activated)
    auto epi_half = [&](const f32x4& a, float winv, int pt, int hh, const u32x4& old, float rf, bool lastl, bool steady, bool ready) {
        const int sub = pt & 1, g = pt >> 1;
        float r0 = a[2 * hh], r1 = a[2 * hh + 1];
        if constexpr (!BF) {  // fp16 weights carry a power-of-two scale
            if (!ready) {
                r0 *= winv;
                r1 *= winv;
            }
        }
        float s0, s1;
        if (ready) {
            s0 = r0;
            s1 = r1;
        } else if (steady && ACT == 0) {
            // anchored to its MFMA group: instruction selection orders pure VALU code only by data dependence and would emit
            // the whole tile's epilogue in one block
            asm volatile("v_sin_f32 %0, %1" : "=v"(s0) : "v"(r0));
            asm volatile("v_sin_f32 %0, %1" : "=v"(s1) : "v"(r1));
        } else {
            if (steady) asm volatile("; epilogue slice anchored to its MFMA group" : "+v"(r0), "+v"(r1));
            s0 = activate<ACT>(r0, p.cg);
            s1 = activate<ACT>(r1, p.cg);
        }
        const float m0 = (float)tb_m[sub][2 * hh], m1 = (float)tb_m[sub][2 * hh + 1];
        float v0, v1;
        if constexpr (RES) {
            float x0, x1;
            x1_unpack2<BF>(old[2 * sub + hh], x0, x1);
            if (steady) {
                v0 = __builtin_fmaf(s0, m0, x0);
                v1 = __builtin_fmaf(s1, m1, x1);
            } else {
                v0 = __builtin_fmaf(x0, rf, s0 * m0);
                v1 = __builtin_fmaf(x1, rf, s1 * m1);
            }
        } else {
            v0 = s0 * m0;
            v1 = s1 * m1;
        }
        if (lastl) {
            part[g] = __builtin_fmaf(v0, (float)tb_w[sub][2 * hh], part[g]);
            part[g] = __builtin_fmaf(v1, (float)tb_w[sub][2 * hh + 1], part[g]);
        } else {
            ew[pt][hh] = x1_pack2<BF>(v0, v1);
        }
    };
    auto epi_store2 = [&](int g, u32x4& d) {  // column group g of the tile = parts 2g (sub-tile 0), 2g+1 (sub-tile 1)
        u32x4 u;
        u[0] = ew[2 * g][0];
        u[1] = ew[2 * g][1];
        u[2] = ew[2 * g + 1][0];
        u[3] = ew[2 * g + 1][1];
        d = x1_to_acc_file(u);
    };
    u32x4 wf_[2][4];  // weight fragments of the group in flight / the next one: [k-step parity * 2 + sub-tile]

// k-step KS_ (compile-time) of tile T, sub-tile SUB: 2 MFMAs (the two column groups share the A operand)
#define MSIREN_X1N_KSTEP(IN, T, Q, KK, SUB)                                                                      \
    do {                                                                                                         \
        if ((Q) == 0 && (KK) == 0) { /* C = bias (revolutions) of the sub-tile's features */                      \
            x1n_mfma<BF>(acc[(T) & 1][0 + (SUB)], wf_[(Q) & 1][2 * (KK) + (SUB)], IN[2 * (2 * (Q) + (KK)) + 0], bia[SUB]); \
            x1n_mfma<BF>(acc[(T) & 1][2 + (SUB)], wf_[(Q) & 1][2 * (KK) + (SUB)], IN[2 * (2 * (Q) + (KK)) + 1], bia[SUB]); \
        } else {                                                                                                 \
            x1n_mfma<BF>(acc[(T) & 1][0 + (SUB)], wf_[(Q) & 1][2 * (KK) + (SUB)], IN[2 * (2 * (Q) + (KK)) + 0], acc[(T) & 1][0 + (SUB)]); \
            x1n_mfma<BF>(acc[(T) & 1][2 + (SUB)], wf_[(Q) & 1][2 * (KK) + (SUB)], IN[2 * (2 * (Q) + (KK)) + 1], acc[(T) & 1][2 + (SUB)]); \
        }                                                                                                        \
    } while (0)

// requested issue order inside a group (8 MFMAs of 16 cycles; an MFMA holds the vector issue port for 8 of them): the weight
// fragment / table reads behind the first MFMAs, two VALU of the epilogue slice in every gap
#define MSIREN_X1N_SGB()                                                                      \
    do {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                    \
    } while (0)

    // Group Q of tile T = one scheduling region: k-steps 2Q, 2Q+1 (8 MFMAs, 128 cycles), the LDS reads of the NEXT group's
    // four weight fragments (Q == 7: the next tile's first, from the next ring buffer, which the mid-tile barrier has already
    // published), one slice of the previous tile's epilogue and the table reads it needs later.
    // Epilogue schedule.  T > 0: tile T-1, half (Q & 1) of part Q >> 1 per group.  T == 0: the previous layer's tile 15
    // ("pending"), whose result feeds k-step 15 of THIS tile (group 7): parts 0..3 in groups 0..3, stores in groups 4 and 5.
#define MSIREN_X1N_GROUP(IN, OUT, T, Q, LASTF)                                                \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if ((Q) >= 4) { /* two of the eight DMA pieces of chunk c+R-1 per group */            \
            MSIREN_X1N_DMA(2 * ((Q) & 3));                                                    \
            MSIREN_X1N_DMA(2 * ((Q) & 3) + 1);                                                \
        }                                                                                     \
        {                                                                                     \
            const u32x4* src_ = (Q) < 7 ? ring_ + (4 * (((Q) + 1) & 7)) * 64 : ringn_;        \
            wf_[((Q) + 1) & 1][0] = src_[0 * 64];                                             \
            wf_[((Q) + 1) & 1][1] = src_[1 * 64];                                             \
            wf_[((Q) + 1) & 1][2] = src_[2 * 64];                                             \
            wf_[((Q) + 1) & 1][3] = src_[3 * 64];                                             \
        }                                                                                     \
        if ((T) == 0) {                                                                       \
            if ((Q) == 0) tbl_load(1, mlp_, woutB, 15, false);                                \
            if ((Q) < 4) {                                                                    \
                epi_half(acc[1][(Q) & 3], wip_, (Q) & 3, 0, OUT[30 + (((Q) & 3) >> 1)], rfp_, false, false, l_ == 1); \
                epi_half(acc[1][(Q) & 3], wip_, (Q) & 3, 1, OUT[30 + (((Q) & 3) >> 1)], rfp_, false, false, l_ == 1); \
            }                                                                                 \
            if ((Q) == 4) epi_store2(0, IN[30]);                                              \
            if ((Q) == 5) epi_store2(1, IN[31]);                                              \
        } else {                                                                              \
            if ((Q) == 0) tbl_load(1, ml_, wo_, (T) - 1, LASTF);                              \
            epi_half(acc[((T) + 1) & 1][(Q) >> 1], wi_, (Q) >> 1, (Q) & 1, IN[2 * ((T) - 1) + ((Q) >> 2)], 1.0f, LASTF, true, false); \
            if ((Q) == 5 && !(LASTF)) epi_store2(0, OUT[2 * ((T) - 1)]);                      \
        }                                                                                     \
        /* bias fragments of the NEXT tile (its first MFMAs are a group or two away; bia is free after group 0) */ \
        if ((Q) == 5) bias_load(0, (T) < 15 ? bl_ : bnx_, ((T) + 1) & 15);                    \
        if ((Q) == 6) bias_load(1, (T) < 15 ? bl_ : bnx_, ((T) + 1) & 15);                    \
        if ((Q) == 7) tbl_load(0, ml_, wo_, (T), LASTF); /* sub-tile 0 of THIS tile's epilogue (runs next tile) */ \
        MSIREN_X1N_KSTEP(IN, T, Q, 0, 0);                                                     \
        MSIREN_X1N_KSTEP(IN, T, Q, 0, 1);                                                     \
        MSIREN_X1N_KSTEP(IN, T, Q, 1, 0);                                                     \
        MSIREN_X1N_KSTEP(IN, T, Q, 1, 1);                                                     \
        MSIREN_X1N_SGB();                                                                     \
    } while (0)

    // One tile = one 32 KB weight chunk; ring synchronised in the MIDDLE of the tile
#define MSIREN_X1N_TILE(IN, OUT, T, LASTF)                                                    \
    do {                                                                                      \
        const u32x4* ring_ = reinterpret_cast<const u32x4*>(smem + LY::ring + rd_buf * X1_CHUNK_BYTES) + lane; \
        rd_buf = rd_buf + 1 == R ? 0 : rd_buf + 1;                                            \
        const u32x4* ringn_ = reinterpret_cast<const u32x4*>(smem + LY::ring + rd_buf * X1_CHUNK_BYTES) + lane; \
        MSIREN_X1N_GROUP(IN, OUT, T, 0, LASTF);                                               \
        MSIREN_X1N_GROUP(IN, OUT, T, 1, LASTF);                                               \
        MSIREN_X1N_GROUP(IN, OUT, T, 2, LASTF);                                               \
        MSIREN_X1N_GROUP(IN, OUT, T, 3, LASTF);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 8) : "memory");                    \
        __builtin_amdgcn_s_barrier();                                                         \
        dma_begin();                                                                          \
        MSIREN_X1N_GROUP(IN, OUT, T, 4, LASTF);                                               \
        MSIREN_X1N_GROUP(IN, OUT, T, 5, LASTF);                                               \
        MSIREN_X1N_GROUP(IN, OUT, T, 6, LASTF);                                               \
        MSIREN_X1N_GROUP(IN, OUT, T, 7, LASTF);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if ((T) > 0 && !(LASTF)) epi_store2(1, OUT[2 * ((T) - 1) + 1]);                       \
    } while (0)

    // one hidden layer: IN -> OUT (the previous layer's last tile is pending in acc[1])
#define MSIREN_X1N_LAYER(IN, OUT, LIDX, LASTF)                                                \
    do {                                                                                      \
        const int l_ = (LIDX);                                                                \
        const unsigned char* wo_ = woutB; /* read by the final-layer instances only */        \
        const unsigned char* bl_ = biasB + (l_ - 1) * 2048;                                   \
        /* bias rows of the layer after this one; after the final hidden layer: layer 1 of the next pass */ \
        const unsigned char* bnx_ = (LASTF) ? biasB : biasB + l_ * 2048;                      \
        const unsigned char* ml_ = modB + l_ * 1024;                                          \
        const unsigned char* mlp_ = modB + (l_ - 1) * 1024;                                   \
        const float wi_ = winvT[l_ - 1], wip_ = l_ > 1 ? winvT[l_ - 2] : 1.0f;              \
        const float rfp_ = l_ > 1 ? 1.0f : 0.0f; /* layer 0 has no skip connection */         \
        MSIREN_X1N_TILE(IN, OUT, 0, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 1, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 2, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 3, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 4, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 5, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 6, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 7, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 8, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 9, LASTF);                                                   \
        MSIREN_X1N_TILE(IN, OUT, 10, LASTF);                                                  \
        MSIREN_X1N_TILE(IN, OUT, 11, LASTF);                                                  \
        MSIREN_X1N_TILE(IN, OUT, 12, LASTF);                                                  \
        MSIREN_X1N_TILE(IN, OUT, 13, LASTF);                                                  \
        MSIREN_X1N_TILE(IN, OUT, 14, LASTF);                                                  \
        MSIREN_X1N_TILE(IN, OUT, 15, LASTF);                                                  \
    } while (0)

// the final hidden layer's last tile is still pending in acc[1]: its contribution to `part`; its residual input sits in the
// array that layer read (IN)
#define MSIREN_X1N_FINAL(IN)                                                                  \
    do {                                                                                      \
        tbl_load(1, modB + (L - 1) * 1024, woutB, 15, true);                                  \
        _Pragma("unroll") for (int pt = 0; pt < 4; ++pt) {                                    \
            epi_half(acc[1][pt], winvT[L - 2], pt, 0, IN[30 + (pt >> 1)], 1.0f, true, false, false); \
            epi_half(acc[1][pt], winvT[L - 2], pt, 1, IN[30 + (pt >> 1)], 1.0f, true, false, false); \
        }                                                                                     \
    } while (0)

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 8) : "memory");
    __syncthreads();  // tables + first chunk visible
    bias_load(0, biasB, 0);  // layer 1, tile 0 (later passes: loaded at the end of the pass before)
    bias_load(1, biasB, 0);
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
        // the lane's two coordinates (column groups 0 and 1)
        int pc0 = cu * 32 + n16, pc1 = cu * 32 + 16 + n16;
        const bool pv0 = active && pc0 < p.P, pv1 = active && pc1 < p.P;
        pc0 = pc0 < p.P ? pc0 : p.P - 1;
        pc1 = pc1 < p.P ? pc1 : p.P - 1;

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
        if (tid == 0) qslot[(pass + 1) & 1] = nxt;  // read after >= 64 workgroup barriers

        // ---- layer 0 (K = 2) from the per-weight-set table act0(W0 x_p + b0), directly in B-operand order: k-steps 0..14
        //      are finished here (three batches of five k-steps: 20 table loads in flight at once), the last 32 features
        //      ("tile 15") wait in acc[1] as they come from the table, where the first hidden layer's pending-epilogue slot
        //      turns them into X[30], X[31] (modulation only: `ready`)
        const f32x4* s0a = reinterpret_cast<const f32x4*>(p.s0t) + (size_t)q * p.P + pc0;
        const f32x4* s0b = reinterpret_cast<const f32x4*>(p.s0t) + (size_t)q * p.P + pc1;
#pragma unroll
        for (int sb = 0; sb < 15; sb += 5) {
            f32x4 raw[10][2];  // [2 (s - sb) + g][sub]
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    raw[2 * s + 0][sub] = s0a[(size_t)(8 * (sb + s) + 4 * sub) * p.P];
                    raw[2 * s + 1][sub] = s0b[(size_t)(8 * (sb + s) + 4 * sub) * p.P];
                }
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                hf4 m4[2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) m4[sub] = *reinterpret_cast<const hf4*>(modB + (32 * (sb + s) + 16 * sub) * 2);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    u32x4 u;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const f32x4 a = raw[2 * s + g][sub];
                        u[2 * sub] = x1_pack2<BF>(a[0] * (float)m4[sub][0], a[1] * (float)m4[sub][1]);
                        u[2 * sub + 1] = x1_pack2<BF>(a[2] * (float)m4[sub][2], a[3] * (float)m4[sub][3]);
                    }
                    X[2 * (sb + s) + g] = x1_to_acc_file(u);
                }
            }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            acc[1][0 + sub] = s0a[(size_t)(120 + 4 * sub) * p.P];
            acc[1][2 + sub] = s0b[(size_t)(120 + 4 * sub) * p.P];
        }
        tbl_load(0, modB, woutB, 15, false);  // sub-tile 0 of the layer-0 "pending" tile

        part[0] = part[1] = 0.f;
        // hidden layers alternate X->Y / Y->X; the final hidden layer has its own instances (one per input array)
        for (int l = 1;;) {
            if (l == L - 1) {
                MSIREN_X1N_LAYER(X, Y, l, true);
                MSIREN_X1N_FINAL(X);
                break;
            }
            MSIREN_X1N_LAYER(X, Y, l, false);
            ++l;
            if (l == L - 1) {
                MSIREN_X1N_LAYER(Y, X, l, true);
                MSIREN_X1N_FINAL(Y);
                break;
            }
            MSIREN_X1N_LAYER(Y, X, l, false);
            ++l;
        }
        // sum over the four feature sub-groups q; lanes q == 0 / q == 1 store column group 0 / 1
        const float s0v = sum_over_q(part[0]), s1v = sum_over_q(part[1]);
        {
            const float sv = q == 0 ? s0v : s1v;
            const int pc = q == 0 ? pc0 : pc1;
            const bool pv = q == 0 ? pv0 : pv1;
            const float o_ = sin_rev(sv + p.bout);
            if (q < 2 && pv) {
                p.out[(size_t)b * p.P + pc] = o_;
                if constexpr (!BF) {
                    if (!(__builtin_fabsf(o_) <= 2.f) && p.status) *p.status = p.status_val;  // NaN: the fp16 domain was left (or the input was NaN)
                }
            }
        }
        cur_pass = __builtin_amdgcn_readfirstlane(qslot[(pass + 1) & 1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may be in flight when the LDS is released
#undef MSIREN_X1N_FINAL
#undef MSIREN_X1N_LAYER
#undef MSIREN_X1N_TILE
#undef MSIREN_X1N_GROUP
#undef MSIREN_X1N_KSTEP
#undef MSIREN_X1N_SGB
#undef MSIREN_X1N_DMA
}

}  // namespace msiren
