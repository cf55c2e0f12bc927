// Fused modulated-SIREN trunk for gfx950, split-fp16 path on 16x16x32 MFMA tiles ("f16x3n": narrow tiles).
//
// Same arithmetic and data flow as siren_trunk_f16x3.hip.h (three fp16 MFMAs per product, activations in
// registers, weights streamed through an LDS ring, persistent grid with a pass queue) -- read that header
// first -- but the contraction is issued as v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16.
// Why: under the board power limit the chip sustains 1.10-1.12x the FLOP/s on the 16x16x32 shape at the
// same bytes per FLOP (tools/mfma_shape_probe.hip, profiles/r2/01_*: 1.51-1.54 vs 1.36-1.38 PFLOP/s with the
// trunk's LDS fragment reads and epilogue VALU riding along), and the trunk is power-limited on the full chip
// (DESIGN.md §4.2), so the shape, not the schedule, is what moves the clock.
//
// What changes with the shape (reference maths unchanged: src/networks/modulated_siren.py:215-233):
//   * a wave still owns one UNIT = 32 coordinates of one patch, now as TWO column groups of 16: lane
//     (n = lane & 15, q = lane >> 4) holds, for coordinates n and 16 + n, the features 4q..4q+3 of every
//     16-feature output tile (D layout: col = lane & 15, row = 4 (lane >> 4) + reg);
//   * one k-step is 32 features: B fragment [2 s + g] (g = column group) holds, in element j, feature
//         32 s + 16 (j >> 2) + 4 q + (j & 3)
//     i.e. the four accumulator registers of the two 16-feature tiles of a 32-feature tile pair -- the
//     epilogue output is again bit for bit the next layer's B operand, and the host packs the weights
//     (A operand: lane (r = lane & 15, q) element j of k-step s) in that k order;
//   * a weight fragment (16 features x 32 k, 1 KB) feeds the MFMAs of BOTH column groups, so a group of
//     12 MFMAs (192 cycles) still reads 4 fragments -- the LDS traffic per FLOP is that of the 32x32 kernel;
//   * bias / modulation / last_layer tables are per feature, hence shared by the two column groups: they are
//     read once per 16-feature sub-tile (half the table reads of the 32x32 kernel);
//   * the accumulator IS the sine argument: the bias (in revolutions) enters as the C operand of a tile's first
//     MFMAs, and the power-of-two weight scale 2^a of layer l is undone on the other operand -- the modulation
//     table row of layer l-1 is multiplied by 2^-a when it is copied to LDS -- so W' x' = W x exactly and the
//     epilogue has no scale/bias FMA.  (a is chosen so that rms|W'| ~ 0.1: the fp16 lo parts of W' and x' then
//     both sit at the edge of the subnormal range, whose fixed 2^-24 step costs 1-2 of the 22 bits; measured
//     against the fp64 oracle the kernel stays at the fp32 noise floor, tests/test_gpu_parity.py);
//   * the epilogue never forms activation x modulation in fp32: hi = f16(a*m) and lo = f16(a*m - hi) are one
//     v_fma_mix{lo,hi}_f16 each (split_products_pk) -- 2 VALU per element for multiply + split instead of 3.
#pragma once
#include <hip/hip_runtime.h>

#include "siren_trunk_f16_common.hip.h"

namespace msiren {

// First MFMA of an accumulator: D = A*B + C with C = the bias fragment, a register quad of its own (shared by the two
// column groups).  The builtin, so that hipcc sees every MFMA -> MFMA dependency and pads the hazards itself (an asm
// MFMA here once produced wrong results under another sched_group_barrier pattern: hipcc cannot know that an asm
// statement is an MFMA whose result the next MFMA must not read as SrcC a few cycles later).
__device__ __forceinline__ void mfma_n16_first(f32x4& d, const h8& a, const h8& b, const f32x4& c) {
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mfma_n16_acc(f32x4& d, const h8& a, const h8& b) {
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
}

// fp16 hi/lo split of two PRODUCTS a*m (activation x modulation) without forming them in fp32 first:
//   hi = f16(a*m), lo = f16(a*m - hi), each one v_fma_mix{lo,hi}_f16 -- a single rounding of the exact product,
// so hi + lo carries 22 bits of a*m.  Four instructions per pair where multiply + v_cvt_pkrtz + residual took five.
__device__ __forceinline__ void split_products_pk(float a0, float m0, float a1, float m1, fp16x2& hi, fp16x2& lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(a0), "v"(m0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(a1), "v"(m1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(m0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(m1), "v"(h));
    hi = __builtin_bit_cast(fp16x2, h);
    lo = __builtin_bit_cast(fp16x2, l);
}

// Sum of a lane's value over the four feature sub-groups q (lanes 16 apart), in every lane: rows 16 apart first, halves 32
// apart second -- two lane swaps (gfx950) instead of two LDS-routed shuffles.
__device__ __forceinline__ float sum_over_q(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// THE ORDER OF THE last_layer SUM (every split-fp16 trunk produces the same bits for a coordinate, whichever instance and
// batch it travelled in): per column, the 256 products v_f * (m_f * w_f) are accumulated in four chains of 64 features
// (features 64 c .. 64 c + 63, c = 0..3: what one wave of the weight-stationary kernel owns), each chain a sequence of FMAs
// over the lane's features in increasing order, summed over the four feature sub-groups q (sum_over_q), and the four
// chains are added as (c0 + c1) + (c2 + c3); then + bias, then the sine.

// LFIX: 0 = any depth (layer loop at run time); 5 = the YAML depth (num_layers = 5 in every shipped configuration)
// with the four hidden layers as straight-line code.  The register-resident arrays X, Y then never meet at a loop
// header, so register allocation does not depend on hipcc coalescing 256 phi copies (which it does for some
// formulations of the epilogue and not for others: 172 + 228 registers here, 256 + 256 and scratch in the loop form).
template <int ACT, int R, int LFIX = 0, int DBG = 0>
__global__ __launch_bounds__(256, 1) void siren_trunk_f16x3n_kernel(TrunkF16Params p) {
    using LY = F16Lds<R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;     // which four features of a 16-feature tile this lane holds
    const int n16 = lane & 15;   // coordinate inside a 16-column group
    const int L = p.L;  // == LFIX when LFIX != 0; kept a run-time value: as a constant it lets hipcc unroll the table loops and hoist 60 registers' worth of loads
    const int nchunks = (L - 1) * 8;

    // Per-lane byte bases of the LDS tables: every access below is `base + compile-time constant`.
    const unsigned char* woutB = smem + LY::wout + q * 16;  // float per feature
    const unsigned char* zeroB = smem + LY::zero + q * 16;
    const unsigned char* biasB = smem + LY::bias + q * 16;
    float* modT = reinterpret_cast<float*>(smem + LY::mods(L)) + wave * (L * 256);
    const unsigned char* modB = reinterpret_cast<const unsigned char*>(modT) + q * 16;

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

    // ---- weight ring (as in the 32x32 kernel) ---------------------------------------------------------
    volatile int* qslot = reinterpret_cast<volatile int*>(smem + LY::queue(L));
    // p.winv[l] here: factor of the modulation row of layer l (2^-a of layer l+1; 1 for the last hidden layer)
    float* mscaleT = reinterpret_cast<float*>(smem + LY::winv(L));
    if (tid < 16) mscaleT[tid] = p.winv[tid];
    int cur_pass = (int)blockIdx.x;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp) + wave * 8192 + lane * 16 + 4096;
    int dma_id = 0, dma_buf = 0, rd_buf = 0;
    const unsigned char* dsrc_ = wsrc;
    unsigned char* ddst_ = smem + LY::ring + wave * 8192 + 4096;
    auto dma_begin = [&]() {
        dsrc_ = wsrc + (size_t)dma_id * F16_CHUNK_BYTES;
        ddst_ = smem + LY::ring + dma_buf * F16_CHUNK_BYTES + wave * 8192 + 4096;
        dma_id = dma_id + 1 == nchunks ? 0 : dma_id + 1;
        dma_buf = dma_buf + 1 == R ? 0 : dma_buf + 1;
    };
#define MSIREN_DMA_PIECE(I)                                                                               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)dsrc_,                \
                                     (__attribute__((address_space(3))) void*)ddst_, 16, (I) * 1024 - 4096, 0)
    auto dma_next = [&]() {
        dma_begin();
        MSIREN_DMA_PIECE(0);
        MSIREN_DMA_PIECE(1);
        MSIREN_DMA_PIECE(2);
        MSIREN_DMA_PIECE(3);
        MSIREN_DMA_PIECE(4);
        MSIREN_DMA_PIECE(5);
        MSIREN_DMA_PIECE(6);
        MSIREN_DMA_PIECE(7);
    };
    const int total_units = __builtin_amdgcn_readfirstlane(p.plan ? p.plan[1] : p.total_units);
    // unsigned compare: a pass id that came out negative (host/device counter disagreement) ends the workgroup
    const unsigned npasses = (unsigned)(total_units + 3) >> 2;
    if ((unsigned)cur_pass >= npasses) return;
#pragma unroll
    for (int s = 0; s < R - 1; ++s) dma_next();

    h8 Xh[16], Xl[16], Yh[16], Yl[16];  // B fragments [2 * k-step + column group]
    f32x4 acc[2][4];                    // [tile parity][part], part = 2 * column group + sub-tile
    float part4[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // last_layer dot product: [column group][chain of 64 features]

    // epilogue of one 32-feature tile = 4 parts (column group g, 16-feature sub-tile sub), 4 elements each:
    // acc -> (revolutions) -> activation -> modulation -> fp16 split; parts (g, 0) and (g, 1) make up B fragment
    // [2 t + g] of the next layer.
    fp16x2 eh[4][2], el[4][2];
    // modulation / last_layer weight of the two sub-tiles of the tile whose epilogue is in flight, and the bias
    // fragments (C operand) of the two sub-tiles of the NEXT tile
    f32x4 tb_m[2], bia[2];
    auto tbl_load = [&](int sub, const unsigned char* ml, const unsigned char* wo, int t, bool withw) {
        const int fo = (32 * t + 16 * sub) * 4;  // compile-time byte offset
        tb_m[sub] = *reinterpret_cast<const f32x4*>(ml + fo);
        if (withw) tb_m[sub] *= *reinterpret_cast<const f32x4*>(wo + fo);  // final layer: modulation x last_layer.weight
    };
    auto bias_load = [&](int sub, const unsigned char* bl, int t) {
        bia[sub] = *reinterpret_cast<const f32x4*>(bl + (32 * t + 16 * sub) * 4);
        asm("; bias fragment stays in arch VGPRs" : "+v"(bia[sub]));  // left alone hipcc moves it (and the accumulators) to AGPRs and spills
    };
    // half `hh` (elements 2hh, 2hh+1) of part pt: see epi_half of the 32x32 kernel
    // `fresh`: the accumulator was written by the MFMAs just before (the "pending" tile of the previous layer): plain
    // builtins, so that hipcc pads the MFMA -> VALU read hazard.  Elsewhere the accumulator is one tile (>= 12 MFMAs) old.
    // `ready`: the values are layer 0's last 32 features straight from the activation table (already activated).
    auto epi_half = [&](const f32x4& a, float cgl, int pt, int hh, bool lastl, bool fresh = false, int chain = 0, bool ready = false) {
        const int sub = pt & 1, g = pt >> 1;
        float v[2];
        if (fresh) {
            v[0] = ready ? a[2 * hh] : activate<ACT>(a[2 * hh], cgl);
            v[1] = ready ? a[2 * hh + 1] : activate<ACT>(a[2 * hh + 1], cgl);
        } else if constexpr (ACT == 0) {
            // The sine reads the accumulator (= its argument, in revolutions) directly.  Issued through asm so that it is
            // anchored to its MFMA group: instruction selection orders pure VALU code only by data dependence and would
            // emit the whole tile's epilogue in one block ahead of the MFMAs.
            asm volatile("v_sin_f32 %0, %1" : "=v"(v[0]) : "v"(a[2 * hh]));
            asm volatile("v_sin_f32 %0, %1" : "=v"(v[1]) : "v"(a[2 * hh + 1]));
        } else {
            // Morlet: sin(2 pi r) * exp2(cg r^2).  The sine and the first factor of the exponent are issued through one asm
            // (the anchor of this slice; no copy of the accumulator), the rest depends on its outputs.
            float s0, s1, t0, t1;
            asm volatile("v_sin_f32 %0, %2\n\tv_mul_f32 %1, %3, %2" : "=&v"(s0), "=&v"(t0) : "v"(a[2 * hh]), "v"(cgl));
            asm volatile("v_sin_f32 %0, %2\n\tv_mul_f32 %1, %3, %2" : "=&v"(s1), "=&v"(t1) : "v"(a[2 * hh + 1]), "v"(cgl));
            v[0] = s0 * __builtin_amdgcn_exp2f(t0 * a[2 * hh]);
            v[1] = s1 * __builtin_amdgcn_exp2f(t1 * a[2 * hh + 1]);
        }
        if (lastl) {
#pragma unroll
            for (int e = 0; e < 2; ++e) part4[g][chain] = __builtin_fmaf(v[e], tb_m[sub][2 * hh + e], part4[g][chain]);
        } else {
            split_products_pk(v[0], tb_m[sub][2 * hh], v[1], tb_m[sub][2 * hh + 1], eh[pt][hh], el[pt][hh]);
        }
    };
    auto epi_store2 = [&](int g, h8& dh, h8& dl) {  // column group g of the tile = parts 2g (sub-tile 0), 2g+1 (sub-tile 1)
        dh = to_acc_file(pack_h8(eh[2 * g][0], eh[2 * g][1], eh[2 * g + 1][0], eh[2 * g + 1][1]));
        dl = to_acc_file(pack_h8(el[2 * g][0], el[2 * g][1], el[2 * g + 1][0], el[2 * g + 1][1]));
    };
    h8 wf_[2][4];  // weight fragments of the k-step in flight / the next one: [hi, lo] of sub-tile 0, [hi, lo] of sub-tile 1

    // k-step Q of tile T, sub-tile SUB: 6 MFMAs (3 products x 2 column groups); the same A operand feeds
    // consecutive MFMAs
#define MSIREN_N16_KSTEP(INh, INl, T, Q, SUB)                                                             \
    do {                                                                                                  \
        if ((Q) == 0) { /* C = bias (revolutions) of the sub-tile's features, the same for both column groups */ \
            mfma_n16_first(acc[(T) & 1][0 + (SUB)], wf_[(Q) & 1][2 * (SUB) + 1], INh[2 * (Q) + 0], bia[SUB]); \
            mfma_n16_first(acc[(T) & 1][2 + (SUB)], wf_[(Q) & 1][2 * (SUB) + 1], INh[2 * (Q) + 1], bia[SUB]); \
        } else {                                                                                          \
            mfma_n16_acc(acc[(T) & 1][0 + (SUB)], wf_[(Q) & 1][2 * (SUB) + 1], INh[2 * (Q) + 0]);          \
            mfma_n16_acc(acc[(T) & 1][2 + (SUB)], wf_[(Q) & 1][2 * (SUB) + 1], INh[2 * (Q) + 1]);          \
        }                                                                                                 \
        mfma_n16_acc(acc[(T) & 1][0 + (SUB)], wf_[(Q) & 1][2 * (SUB)], INl[2 * (Q) + 0]);                  \
        mfma_n16_acc(acc[(T) & 1][2 + (SUB)], wf_[(Q) & 1][2 * (SUB)], INl[2 * (Q) + 1]);                  \
        mfma_n16_acc(acc[(T) & 1][0 + (SUB)], wf_[(Q) & 1][2 * (SUB)], INh[2 * (Q) + 0]);                  \
        mfma_n16_acc(acc[(T) & 1][2 + (SUB)], wf_[(Q) & 1][2 * (SUB)], INh[2 * (Q) + 1]);                  \
    } while (0)

// Requested issue order inside a group (12 MFMAs of 16 cycles; an MFMA holds the vector issue port for 8 of
// them, so one or two short VALU instructions ride in each gap): the four weight-fragment reads first, then the
// table reads, the VALU of the epilogue slice spread over the rest.
#ifndef MSIREN_N16_SGB_VARIANT
#define MSIREN_N16_SGB_VARIANT 1
#endif
#if MSIREN_N16_SGB_VARIANT == 0
#define MSIREN_N16_SGB() do {} while (0)
#elif MSIREN_N16_SGB_VARIANT == 1
#define MSIREN_N16_SGB()                                                                      \
    do {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
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
#elif MSIREN_N16_SGB_VARIANT == 3  /* all LDS reads of the group up front, then one VALU per MFMA */
#define MSIREN_N16_SGB()                                                                      \
    do {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                    \
    } while (0)
#elif MSIREN_N16_SGB_VARIANT == 4  /* two LDS reads behind each of the first MFMAs, VALU from the fifth MFMA on */
#define MSIREN_N16_SGB()                                                                      \
    do {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                    \
    } while (0)
#else  /* 2: MFMAs in pairs (same A operand back to back), two VALU after each pair */
#define MSIREN_N16_SGB()                                                                      \
    do {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);                                   \
    } while (0)
#endif

    // Group Q of tile T = one scheduling region: k-step Q (12 MFMAs, 192 cycles), the LDS reads of the NEXT
    // k-step's four weight fragments (Q == 7: the next tile's first k-step, from the next ring buffer, which the
    // mid-tile barrier has already published), one slice of the previous tile's epilogue and the table reads it
    // needs later.
    // Epilogue schedule.  T > 0: tile T-1, half (Q & 1) of part Q >> 1 per group.  T == 0: the previous layer's tile 7
    // ("pending"), whose result feeds k-step 7 of THIS tile: parts 0..3 in groups 0..3, stores in groups 4 and 5.
    // Tables: sub-tile 0's are read in group 7 of the tile itself, sub-tile 1's in group 0 of the next tile (its
    // registers are still in use by part 3 of the tile before until group 7).
// Ablation builds (timing only, results wrong; never shipped): -DMSIREN_N16_ABL=bitmask
//   1 = no epilogue work in the groups, 2 = no ring barrier / vmcnt wait, 4 = no weight-fragment LDS reads, 8 = no DMA
#ifndef MSIREN_N16_ABL
#define MSIREN_N16_ABL 0
#endif
#define MSIREN_N16_GROUP(INh, INl, OUTh, OUTl, T, Q, LASTF)                                       \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if ((Q) >= 4 && !(MSIREN_N16_ABL & 8)) { /* two of the eight DMA pieces of chunk c+R-1 per group */ \
            MSIREN_DMA_PIECE(2 * ((Q) & 3));                                                  \
            MSIREN_DMA_PIECE(2 * ((Q) & 3) + 1);                                              \
        }                                                                                     \
        if (MSIREN_N16_ABL & 4) { /* fragments stay what they are, opaquely */               \
            asm volatile("" : "+v"(wf_[((Q) + 1) & 1][0]), "+v"(wf_[((Q) + 1) & 1][1]), "+v"(wf_[((Q) + 1) & 1][2]), "+v"(wf_[((Q) + 1) & 1][3])); \
        } else {                                                                              \
            const h8* src_ = (Q) < 7 ? ring_ + (4 * (((Q) + 1) & 7)) * 64 : ringn_;           \
            wf_[((Q) + 1) & 1][0] = src_[0 * 64];                                             \
            wf_[((Q) + 1) & 1][1] = src_[1 * 64];                                             \
            wf_[((Q) + 1) & 1][2] = src_[2 * 64];                                             \
            wf_[((Q) + 1) & 1][3] = src_[3 * 64];                                             \
        }                                                                                     \
        if (MSIREN_N16_ABL & 1) { /* keep the accumulators alive so the MFMAs are not dead code */ \
            if ((Q) == 0) asm volatile("" ::"v"(acc[((T) + 1) & 1][0]), "v"(acc[((T) + 1) & 1][1]), "v"(acc[((T) + 1) & 1][2]), "v"(acc[((T) + 1) & 1][3])); \
        } else if ((T) == 0) {                                                                       \
            if ((Q) == 0) tbl_load(1, mlp_, zeroB, 7, false);                                 \
            if ((Q) < 4) {                                                                    \
                epi_half(acc[1][(Q) & 3], p.cg, (Q) & 3, 0, false, true, 0, l_ == 1);           \
                epi_half(acc[1][(Q) & 3], p.cg, (Q) & 3, 1, false, true, 0, l_ == 1);           \
            }                                                                                 \
            if ((Q) == 4) epi_store2(0, INh[14], INl[14]);                                    \
            if ((Q) == 5) epi_store2(1, INh[15], INl[15]);                                    \
        } else {                                                                              \
            if ((Q) == 0) tbl_load(1, ml_, wo_, ((T) + 7) & 7, LASTF);                        \
            epi_half(acc[((T) + 1) & 1][(Q) >> 1], p.cg, (Q) >> 1, (Q) & 1, LASTF, false, (((T) + 7) & 7) >> 1); \
            if ((Q) == 5 && !(LASTF)) epi_store2(0, OUTh[(2 * (T) + 14) & 15], OUTl[(2 * (T) + 14) & 15]); \
        }                                                                                     \
        /* bias fragments of the NEXT tile (its first MFMAs are a group or two away; bia is free after group 0) */ \
        if ((Q) == 5) bias_load(0, (T) < 7 ? bl_ : bnx_, ((T) + 1) & 7);                      \
        if ((Q) == 6) bias_load(1, (T) < 7 ? bl_ : bnx_, ((T) + 1) & 7);                      \
        if ((Q) == 7) tbl_load(0, ml_, wo_, (T), LASTF); /* sub-tile 0 of THIS tile's epilogue (runs next tile) */ \
        MSIREN_N16_KSTEP(INh, INl, T, Q, 0);                                                  \
        MSIREN_N16_KSTEP(INh, INl, T, Q, 1);                                                  \
        MSIREN_N16_SGB();                                                                     \
    } while (0)

    // One tile = one 32 KB weight chunk; ring synchronised in the MIDDLE of the tile (see the 32x32 kernel).
#define MSIREN_N16_TILE(INh, INl, OUTh, OUTl, T, LASTF)                                           \
    do {                                                                                      \
        const h8* ring_ = reinterpret_cast<const h8*>(smem + LY::ring + rd_buf * F16_CHUNK_BYTES) + lane; \
        rd_buf = rd_buf + 1 == R ? 0 : rd_buf + 1;                                            \
        const h8* ringn_ = reinterpret_cast<const h8*>(smem + LY::ring + rd_buf * F16_CHUNK_BYTES) + lane; \
        MSIREN_N16_GROUP(INh, INl, OUTh, OUTl, T, 0, LASTF);                                    \
        MSIREN_N16_GROUP(INh, INl, OUTh, OUTl, T, 1, LASTF);                                    \
        MSIREN_N16_GROUP(INh, INl, OUTh, OUTl, T, 2, LASTF);                                    \
        MSIREN_N16_GROUP(INh, INl, OUTh, OUTl, T, 3, LASTF);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (!(MSIREN_N16_ABL & 2)) {                                                          \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 8) : "memory");                \
            __builtin_amdgcn_s_barrier();                                                     \
        }                                                                                     \
        dma_begin();                                                                          \
        MSIREN_N16_GROUP(INh, INl, OUTh, OUTl, T, 4, LASTF);                                    \
        MSIREN_N16_GROUP(INh, INl, OUTh, OUTl, T, 5, LASTF);                                    \
        MSIREN_N16_GROUP(INh, INl, OUTh, OUTl, T, 6, LASTF);                                    \
        MSIREN_N16_GROUP(INh, INl, OUTh, OUTl, T, 7, LASTF);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if ((T) > 0 && !(LASTF) && !(MSIREN_N16_ABL & 1)) epi_store2(1, OUTh[(2 * (T) + 15) & 15], OUTl[(2 * (T) + 15) & 15]); \
        if constexpr (DBG) { stamp(8 + dbg_tile); ++dbg_tile; }                               \
    } while (0)

    // one hidden layer: IN -> OUT (see the 32x32 kernel: the previous layer's last tile is pending in acc[1])
#define MSIREN_N16_LAYER(INh, INl, OUTh, OUTl, LIDX, LASTF)                                       \
    do {                                                                                      \
        const int l_ = (LIDX);                                                                \
        const unsigned char* wo_ = woutB; /* read by the final-layer instance only */         \
        const unsigned char* bl_ = biasB + (l_ - 1) * 1024;                                   \
        /* bias rows of the layer after this one; after the final hidden layer: layer 1 of the next pass */ \
        const unsigned char* bnx_ = (LASTF) ? biasB : biasB + l_ * 1024;                      \
        const unsigned char* ml_ = modB + l_ * 1024;                                          \
        const unsigned char* mlp_ = modB + (l_ - 1) * 1024;                                   \
        MSIREN_N16_TILE(INh, INl, OUTh, OUTl, 0, LASTF);                                        \
        MSIREN_N16_TILE(INh, INl, OUTh, OUTl, 1, LASTF);                                        \
        MSIREN_N16_TILE(INh, INl, OUTh, OUTl, 2, LASTF);                                        \
        MSIREN_N16_TILE(INh, INl, OUTh, OUTl, 3, LASTF);                                        \
        MSIREN_N16_TILE(INh, INl, OUTh, OUTl, 4, LASTF);                                        \
        MSIREN_N16_TILE(INh, INl, OUTh, OUTl, 5, LASTF);                                        \
        MSIREN_N16_TILE(INh, INl, OUTh, OUTl, 6, LASTF);                                        \
        MSIREN_N16_TILE(INh, INl, OUTh, OUTl, 7, LASTF);                                        \
    } while (0)

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 8) : "memory");
    __syncthreads();  // tables + first chunk visible
    bias_load(0, biasB, 0);  // layer 1, tile 0 (later passes: loaded at the end of the pass before)
    bias_load(1, biasB, 0);
    {   // first weight fragments of the very first tile
        const h8* r0 = reinterpret_cast<const h8*>(smem + LY::ring) + lane;
        wf_[0][0] = r0[0 * 64];
        wf_[0][1] = r0[1 * 64];
        wf_[0][2] = r0[2 * 64];
        wf_[0][3] = r0[3 * 64];
    }

    for (int pass = 0; (unsigned)cur_pass < npasses; ++pass) {
        auto stamp = [&](int i) {
            if constexpr (DBG) {
                const unsigned long long t = i == 7 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
                if (tid == 0 && pass < 4) p.stamps[((size_t)blockIdx.x * 4 + pass) * 48 + i] = t;
            }
        };
        stamp(0);
        int dbg_tile = 0;
        (void)dbg_tile;
        int unit = cur_pass * 4 + wave;
        const bool active = unit < total_units;
        unit = (active ? unit : total_units - 1) + p.unit_base;
        const int b = unit / p.units_per_patch;
        const int cu = unit - b * p.units_per_patch;
        // the lane's two coordinates (column groups 0 and 1)
        int pc0 = cu * 32 + n16, pc1 = cu * 32 + 16 + n16;
        const bool pv0 = active && pc0 < p.P, pv1 = active && pc1 < p.P;
        pc0 = pc0 < p.P ? pc0 : p.P - 1;
        pc1 = pc1 < p.P ? pc1 : p.P - 1;

        // the next pass id is fetched a whole pass ahead, together with the loads below (one wait)
        int nxt = 0;
        if (tid == 0) nxt = (int)((unsigned)atomicAdd(p.pass_counter, 1) - p.pass_base) + (int)gridDim.x;
        // this wave's modulation table: (L, 256) floats of patch b
        bool bad_mod = false;
        for (int l = 0; l < L; ++l) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(p.mods + ((size_t)l * p.B + b) * 256 + lane * 4);
            const f32x4 ms = m * mscaleT[l];  // exact: a power of two
            bad_mod |= f16_out_of_range(ms);
            *reinterpret_cast<f32x4*>(modT + l * 256 + lane * 4) = ms;
        }
        if (bad_mod && p.status) *p.status = p.status_val;
        if (tid == 0) qslot[(pass + 1) & 1] = nxt;  // read after >= 32 workgroup barriers

        // ---- layer 0 (K = 2) from the per-weight-set table act0(W0 x_p + b0), directly in B-operand order:
        //      element j of fragment [2 s + g] is feature 32 s + 16 (j >> 2) + 4 q + (j & 3) at the lane's
        //      coordinate of column group g.  K-steps 0..6 are finished here; the last 32 features ("tile 7")
        //      wait in acc[1] as they come from the table, where the first hidden layer's pending-epilogue slot
        //      turns them into X[14], X[15] (modulation and split only: `ready`).  Every split-fp16 trunk takes all
        //      256 layer-0 activations from this table, so that they agree bit for bit.
        const f32x4* s0a = reinterpret_cast<const f32x4*>(p.s0t) + (size_t)q * p.P + pc0;
        const f32x4* s0b = reinterpret_cast<const f32x4*>(p.s0t) + (size_t)q * p.P + pc1;
        f32x4 raw[14][2];  // [2 s + g][sub]: all 28 loads in flight at once
#pragma unroll
        for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                raw[2 * s + 0][sub] = s0a[(size_t)(8 * s + 4 * sub) * p.P];
                raw[2 * s + 1][sub] = s0b[(size_t)(8 * s + 4 * sub) * p.P];
            }
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            f32x4 m4[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) m4[sub] = *reinterpret_cast<const f32x4*>(modB + (32 * s + 16 * sub) * 4);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                fp16x2 hh[2][2], ll[2][2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const f32x4 a = raw[2 * s + g][sub];
                    split_products_pk(a[0], m4[sub][0], a[1], m4[sub][1], hh[sub][0], ll[sub][0]);
                    split_products_pk(a[2], m4[sub][2], a[3], m4[sub][3], hh[sub][1], ll[sub][1]);
                }
                Xh[2 * s + g] = to_acc_file(pack_h8(hh[0][0], hh[0][1], hh[1][0], hh[1][1]));
                Xl[2 * s + g] = to_acc_file(pack_h8(ll[0][0], ll[0][1], ll[1][0], ll[1][1]));
            }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            acc[1][0 + sub] = s0a[(size_t)(56 + 4 * sub) * p.P];
            acc[1][2 + sub] = s0b[(size_t)(56 + 4 * sub) * p.P];
        }
        tbl_load(0, modB, zeroB, 7, false);  // sub-tile 0 of the layer-0 "pending" tile

#pragma unroll
        for (int c = 0; c < 4; ++c) part4[0][c] = part4[1][c] = 0.f;
        stamp(1);
        // Hidden layers alternate X->Y and Y->X; the final hidden layer has its own instances (see the 32x32 kernel).
        if constexpr (LFIX == 5) {
            MSIREN_N16_LAYER(Xh, Xl, Yh, Yl, 1, false);
            MSIREN_N16_LAYER(Yh, Yl, Xh, Xl, 2, false);
            MSIREN_N16_LAYER(Xh, Xl, Yh, Yl, 3, false);
            MSIREN_N16_LAYER(Yh, Yl, Xh, Xl, 4, true);
        } else {
            for (int l = 1;;) {
                if (l == L - 1) {
                    MSIREN_N16_LAYER(Xh, Xl, Yh, Yl, l, true);
                    break;
                }
                MSIREN_N16_LAYER(Xh, Xl, Yh, Yl, l, false);
                ++l;
                if (l == L - 1) {
                    MSIREN_N16_LAYER(Yh, Yl, Xh, Xl, l, true);
                    break;
                }
                MSIREN_N16_LAYER(Yh, Yl, Xh, Xl, l, false);
                ++l;
            }
        }
        stamp(2);
        // the final hidden layer's last tile is still pending: its contribution to `part`
        tbl_load(1, modB + (L - 1) * 1024, woutB, 7, true);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            epi_half(acc[1][pt], p.cg, pt, 0, true, true, 3);
            epi_half(acc[1][pt], p.cg, pt, 1, true, true, 3);
        }
        // the canonical last_layer sum (see above); lanes q == 0 / q == 1 store column group 0 / 1
        const float s0v = (sum_over_q(part4[0][0]) + sum_over_q(part4[0][1])) + (sum_over_q(part4[0][2]) + sum_over_q(part4[0][3]));
        const float s1v = (sum_over_q(part4[1][0]) + sum_over_q(part4[1][1])) + (sum_over_q(part4[1][2]) + sum_over_q(part4[1][3]));
        {
            const float sv = q == 0 ? s0v : s1v;
            const int pc = q == 0 ? pc0 : pc1;
            const bool pv = q == 0 ? pv0 : pv1;
            if (q < 2 && pv) p.out[(size_t)b * p.P + pc] = sin_rev(sv + p.bout);
        }
        cur_pass = __builtin_amdgcn_readfirstlane(qslot[(pass + 1) & 1]);
        stamp(6);
        stamp(7);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may be in flight when the LDS is released
#undef MSIREN_N16_LAYER
#undef MSIREN_N16_TILE
#undef MSIREN_N16_GROUP
#undef MSIREN_N16_KSTEP
#undef MSIREN_N16_SGB
#undef MSIREN_DMA_PIECE
}

}  // namespace msiren
