// Split-fp16 trunk, HALF-unit instance ("f16x3h"): a wave owns 16 coordinates (one 16-column MFMA group) instead of 32.
//
// Same arithmetic, weight stream, LDS layout, tables and pass queue as siren_trunk_f16x3n.hip.h (read that header
// first; reference maths: src/networks/modulated_siren.py:215-233) -- the kernels are interchangeable unit for unit
// and produce the same bits.  A wave here issues half the MFMAs per layer for the same weight-fragment reads, so per
// coordinate it costs more LDS traffic; what it buys is LATENCY where the chip is not full anyway: small batches
// (BASELINE configs[0], a single tile) -- twice as many waves share the work, a pass takes about half the time
// (launch_dispatch.hip: launch_trunk_f16x3 selects it when all units fit in one round even as half-units).
// Layout differences from the 32-coordinate kernel: B fragment [s] is k-step s (one column group), accumulators
// acc[tile parity][sub-tile], the epilogue of a 32-feature tile is 2 parts (sub-tiles) x 2 halves, spread over groups
// 0, 2, 4, 6 of the next tile.  No sched_group_barrier choreography: this instance is not the throughput path.
#pragma once
#include <hip/hip_runtime.h>

#include "siren_trunk_f16x3n.hip.h"

namespace msiren {

template <int ACT, int R, int LFIX = 0>
__global__ __launch_bounds__(256, 1) void siren_trunk_f16x3h_kernel(TrunkF16Params p) {
    using LY = F16Lds<R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;     // which four features of a 16-feature tile this lane holds
    const int n16 = lane & 15;   // coordinate inside the unit
    const int L = p.L;
    const int nchunks = (L - 1) * 8;

    const unsigned char* woutB = smem + LY::wout + q * 16;
    const unsigned char* zeroB = smem + LY::zero + q * 16;
    const unsigned char* biasB = smem + LY::bias + q * 16;
    float* modT = reinterpret_cast<float*>(smem + LY::mods(L)) + wave * (L * 256);
    const unsigned char* modB = reinterpret_cast<const unsigned char*>(modT) + q * 16;

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

    volatile int* qslot = reinterpret_cast<volatile int*>(smem + LY::queue(L));
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
    const unsigned npasses = (unsigned)(total_units + 3) >> 2;
    if ((unsigned)cur_pass >= npasses) return;
#pragma unroll
    for (int s = 0; s < R - 1; ++s) dma_next();

    h8 Xh[8], Xl[8], Yh[8], Yl[8];  // B fragments [k-step]
    f32x4 acc[2][2];                // [tile parity][sub-tile]
    float part4[4] = {0.f, 0.f, 0.f, 0.f};  // last_layer dot product, four chains of 64 features (canonical order: f16x3n header)

    fp16x2 eh[2][2], el[2][2];      // [sub-tile][half]
    f32x4 tb_m[2], bia[2];
    auto tbl_load = [&](int sub, const unsigned char* ml, const unsigned char* wo, int t, bool withw) {
        const int fo = (32 * t + 16 * sub) * 4;
        tb_m[sub] = *reinterpret_cast<const f32x4*>(ml + fo);
        if (withw) tb_m[sub] *= *reinterpret_cast<const f32x4*>(wo + fo);  // final layer: modulation x last_layer.weight
    };
    auto bias_load = [&](int sub, const unsigned char* bl, int t) {
        bia[sub] = *reinterpret_cast<const f32x4*>(bl + (32 * t + 16 * sub) * 4);
        asm("; bias fragment stays in arch VGPRs" : "+v"(bia[sub]));
    };
    auto epi_half = [&](const f32x4& a, float cgl, int sub, int hh, bool lastl, bool fresh = false, int chain = 0, bool ready = false) {
        float v[2];
        if (fresh) {  // accumulator written by the MFMAs just before: builtins, hipcc pads the MFMA -> VALU hazard
            v[0] = ready ? a[2 * hh] : activate<ACT>(a[2 * hh], cgl);       // (ready: layer 0's table values)
            v[1] = ready ? a[2 * hh + 1] : activate<ACT>(a[2 * hh + 1], cgl);
        } else if constexpr (ACT == 0) {
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
            for (int e = 0; e < 2; ++e) part4[chain] = __builtin_fmaf(v[e], tb_m[sub][2 * hh + e], part4[chain]);
        } else {
            split_products_pk(v[0], tb_m[sub][2 * hh], v[1], tb_m[sub][2 * hh + 1], eh[sub][hh], el[sub][hh]);
        }
    };
    auto epi_store = [&](h8& dh, h8& dl) {
        dh = to_acc_file(pack_h8(eh[0][0], eh[0][1], eh[1][0], eh[1][1]));
        dl = to_acc_file(pack_h8(el[0][0], el[0][1], el[1][0], el[1][1]));
    };
    h8 wf_[2][4];

#define MSIREN_H16_KSTEP(INh, INl, T, Q, SUB)                                                             \
    do {                                                                                                  \
        if ((Q) == 0) mfma_n16_first(acc[(T) & 1][SUB], wf_[(Q) & 1][2 * (SUB) + 1], INh[Q], bia[SUB]);   \
        else mfma_n16_acc(acc[(T) & 1][SUB], wf_[(Q) & 1][2 * (SUB) + 1], INh[Q]);                         \
        mfma_n16_acc(acc[(T) & 1][SUB], wf_[(Q) & 1][2 * (SUB)], INl[Q]);                                  \
        mfma_n16_acc(acc[(T) & 1][SUB], wf_[(Q) & 1][2 * (SUB)], INh[Q]);                                  \
    } while (0)

    // Epilogue schedule.  T > 0: tile T-1, half ((Q >> 1) & 1) of sub-tile Q >> 2 in the even groups.  T == 0: the previous
    // layer's tile 7 ("pending"): sub-tiles 0, 1 in groups 0, 1, store in group 2 (feeds k-step 7 of this tile).
#define MSIREN_H16_GROUP(INh, INl, OUTh, OUTl, T, Q, LASTF)                                       \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if ((Q) >= 4) {                                                                       \
            MSIREN_DMA_PIECE(2 * ((Q) & 3));                                                  \
            MSIREN_DMA_PIECE(2 * ((Q) & 3) + 1);                                              \
        }                                                                                     \
        {                                                                                     \
            const h8* src_ = (Q) < 7 ? ring_ + (4 * (((Q) + 1) & 7)) * 64 : ringn_;           \
            wf_[((Q) + 1) & 1][0] = src_[0 * 64];                                             \
            wf_[((Q) + 1) & 1][1] = src_[1 * 64];                                             \
            wf_[((Q) + 1) & 1][2] = src_[2 * 64];                                             \
            wf_[((Q) + 1) & 1][3] = src_[3 * 64];                                             \
        }                                                                                     \
        if ((T) == 0) {                                                                       \
            if ((Q) == 0) tbl_load(1, mlp_, zeroB, 7, false);                                 \
            if ((Q) < 2) {                                                                    \
                epi_half(acc[1][(Q) & 1], p.cg, (Q) & 1, 0, false, true, 0, l_ == 1);           \
                epi_half(acc[1][(Q) & 1], p.cg, (Q) & 1, 1, false, true, 0, l_ == 1);           \
            }                                                                                 \
            if ((Q) == 2) epi_store(INh[7], INl[7]);                                          \
        } else {                                                                              \
            if ((Q) == 0) tbl_load(1, ml_, wo_, ((T) + 7) & 7, LASTF);                        \
            if (((Q) & 1) == 0) epi_half(acc[((T) + 1) & 1][(Q) >> 2], p.cg, (Q) >> 2, ((Q) >> 1) & 1, LASTF, false, (((T) + 7) & 7) >> 1); \
            if ((Q) == 7 && !(LASTF)) epi_store(OUTh[((T) + 7) & 7], OUTl[((T) + 7) & 7]);    \
        }                                                                                     \
        if ((Q) == 5) bias_load(0, (T) < 7 ? bl_ : bnx_, ((T) + 1) & 7);                      \
        if ((Q) == 6) bias_load(1, (T) < 7 ? bl_ : bnx_, ((T) + 1) & 7);                      \
        if ((Q) == 7) tbl_load(0, ml_, wo_, (T), LASTF);                                      \
        MSIREN_H16_KSTEP(INh, INl, T, Q, 0);                                                  \
        MSIREN_H16_KSTEP(INh, INl, T, Q, 1);                                                  \
    } while (0)

#define MSIREN_H16_TILE(INh, INl, OUTh, OUTl, T, LASTF)                                           \
    do {                                                                                      \
        const h8* ring_ = reinterpret_cast<const h8*>(smem + LY::ring + rd_buf * F16_CHUNK_BYTES) + lane; \
        rd_buf = rd_buf + 1 == R ? 0 : rd_buf + 1;                                            \
        const h8* ringn_ = reinterpret_cast<const h8*>(smem + LY::ring + rd_buf * F16_CHUNK_BYTES) + lane; \
        MSIREN_H16_GROUP(INh, INl, OUTh, OUTl, T, 0, LASTF);                                    \
        MSIREN_H16_GROUP(INh, INl, OUTh, OUTl, T, 1, LASTF);                                    \
        MSIREN_H16_GROUP(INh, INl, OUTh, OUTl, T, 2, LASTF);                                    \
        MSIREN_H16_GROUP(INh, INl, OUTh, OUTl, T, 3, LASTF);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 8) : "memory");                    \
        __builtin_amdgcn_s_barrier();                                                         \
        dma_begin();                                                                          \
        MSIREN_H16_GROUP(INh, INl, OUTh, OUTl, T, 4, LASTF);                                    \
        MSIREN_H16_GROUP(INh, INl, OUTh, OUTl, T, 5, LASTF);                                    \
        MSIREN_H16_GROUP(INh, INl, OUTh, OUTl, T, 6, LASTF);                                    \
        MSIREN_H16_GROUP(INh, INl, OUTh, OUTl, T, 7, LASTF);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)

#define MSIREN_H16_LAYER(INh, INl, OUTh, OUTl, LIDX, LASTF)                                       \
    do {                                                                                      \
        const int l_ = (LIDX);                                                                \
        const unsigned char* wo_ = woutB;                                                     \
        const unsigned char* bl_ = biasB + (l_ - 1) * 1024;                                   \
        const unsigned char* bnx_ = (LASTF) ? biasB : biasB + l_ * 1024;                      \
        const unsigned char* ml_ = modB + l_ * 1024;                                          \
        const unsigned char* mlp_ = modB + (l_ - 1) * 1024;                                   \
        MSIREN_H16_TILE(INh, INl, OUTh, OUTl, 0, LASTF);                                        \
        MSIREN_H16_TILE(INh, INl, OUTh, OUTl, 1, LASTF);                                        \
        MSIREN_H16_TILE(INh, INl, OUTh, OUTl, 2, LASTF);                                        \
        MSIREN_H16_TILE(INh, INl, OUTh, OUTl, 3, LASTF);                                        \
        MSIREN_H16_TILE(INh, INl, OUTh, OUTl, 4, LASTF);                                        \
        MSIREN_H16_TILE(INh, INl, OUTh, OUTl, 5, LASTF);                                        \
        MSIREN_H16_TILE(INh, INl, OUTh, OUTl, 6, LASTF);                                        \
        MSIREN_H16_TILE(INh, INl, OUTh, OUTl, 7, LASTF);                                        \
    } while (0)

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 8) : "memory");
    __syncthreads();
    bias_load(0, biasB, 0);
    bias_load(1, biasB, 0);
    {
        const h8* r0 = reinterpret_cast<const h8*>(smem + LY::ring) + lane;
        wf_[0][0] = r0[0 * 64];
        wf_[0][1] = r0[1 * 64];
        wf_[0][2] = r0[2 * 64];
        wf_[0][3] = r0[3 * 64];
    }

    for (int pass = 0; (unsigned)cur_pass < npasses; ++pass) {
        int unit = cur_pass * 4 + wave;
        const bool active = unit < total_units;
        unit = (active ? unit : total_units - 1) + p.unit_base;
        const int b = unit / p.units_per_patch;
        const int cu = unit - b * p.units_per_patch;
        int pc0 = cu * 16 + n16;
        const bool pv0 = active && pc0 < p.P;
        pc0 = pc0 < p.P ? pc0 : p.P - 1;

        int nxt = 0;
        if (tid == 0) nxt = (int)((unsigned)atomicAdd(p.pass_counter, 1) - p.pass_base) + (int)gridDim.x;
        bool bad_mod = false;
        for (int l = 0; l < L; ++l) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(p.mods + ((size_t)l * p.B + b) * 256 + lane * 4);
            const f32x4 ms = m * mscaleT[l];  // exact: a power of two
            bad_mod |= f16_out_of_range(ms);
            *reinterpret_cast<f32x4*>(modT + l * 256 + lane * 4) = ms;
        }
        if (bad_mod && p.status) *p.status = p.status_val;
        if (tid == 0) qslot[(pass + 1) & 1] = nxt;

        // layer 0 from the table: k-steps 0..6 finished here, the last 32 features wait in acc[1] for the first hidden
        // layer's pending-epilogue slot (see the 32-coordinate kernel)
        const f32x4* s0a = reinterpret_cast<const f32x4*>(p.s0t) + (size_t)q * p.P + pc0;
        f32x4 raw[7][2];
#pragma unroll
        for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) raw[s][sub] = s0a[(size_t)(8 * s + 4 * sub) * p.P];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            fp16x2 hh[2][2], ll[2][2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const f32x4 m4 = *reinterpret_cast<const f32x4*>(modB + (32 * s + 16 * sub) * 4);
                const f32x4 a = raw[s][sub];
                split_products_pk(a[0], m4[0], a[1], m4[1], hh[sub][0], ll[sub][0]);
                split_products_pk(a[2], m4[2], a[3], m4[3], hh[sub][1], ll[sub][1]);
            }
            Xh[s] = to_acc_file(pack_h8(hh[0][0], hh[0][1], hh[1][0], hh[1][1]));
            Xl[s] = to_acc_file(pack_h8(ll[0][0], ll[0][1], ll[1][0], ll[1][1]));
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) acc[1][sub] = s0a[(size_t)(56 + 4 * sub) * p.P];
        tbl_load(0, modB, zeroB, 7, false);

        part4[0] = part4[1] = part4[2] = part4[3] = 0.f;
        if constexpr (LFIX == 5) {
            MSIREN_H16_LAYER(Xh, Xl, Yh, Yl, 1, false);
            MSIREN_H16_LAYER(Yh, Yl, Xh, Xl, 2, false);
            MSIREN_H16_LAYER(Xh, Xl, Yh, Yl, 3, false);
            MSIREN_H16_LAYER(Yh, Yl, Xh, Xl, 4, true);
        } else {
            for (int l = 1;;) {
                if (l == L - 1) {
                    MSIREN_H16_LAYER(Xh, Xl, Yh, Yl, l, true);
                    break;
                }
                MSIREN_H16_LAYER(Xh, Xl, Yh, Yl, l, false);
                ++l;
                if (l == L - 1) {
                    MSIREN_H16_LAYER(Yh, Yl, Xh, Xl, l, true);
                    break;
                }
                MSIREN_H16_LAYER(Yh, Yl, Xh, Xl, l, false);
                ++l;
            }
        }
        tbl_load(1, modB + (L - 1) * 1024, woutB, 7, true);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            epi_half(acc[1][sub], p.cg, sub, 0, true, true, 3);
            epi_half(acc[1][sub], p.cg, sub, 1, true, true, 3);
        }
        const float sv = (sum_over_q(part4[0]) + sum_over_q(part4[1])) + (sum_over_q(part4[2]) + sum_over_q(part4[3]));
        if (q == 0 && pv0) p.out[(size_t)b * p.P + pc0] = sin_rev(sv + p.bout);
        cur_pass = __builtin_amdgcn_readfirstlane(qslot[(pass + 1) & 1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef MSIREN_H16_LAYER
#undef MSIREN_H16_TILE
#undef MSIREN_H16_GROUP
#undef MSIREN_H16_KSTEP
#undef MSIREN_DMA_PIECE
}

}  // namespace msiren
