// RECORD ONLY -- not part of the product build since round 4 (DESIGN.md §4.2a has its measurements).
// Round 1's split-fp16 trunk on 32x32x16 MFMA tiles; it built against mri_inr_amd/csrc at commit 8569a88
// (make AB32=1, MSIREN_F16_TILE=32).  The shared definitions it used now live in siren_trunk_f16_common.hip.h.
#pragma once
#include "../../mri_inr_amd/csrc/siren_trunk_f16_common.hip.h"

namespace msiren {

template <int ACT, int R, int DBG = 0>
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
#ifdef MSIREN_CLAIM_RF
    asm volatile("; claim the whole register file" ::: "v255", "a255");
#endif

    // Per-lane byte bases of the LDS tables.  Every table access below is `base + compile-time
    // constant` so that it folds into the ds_read offset field: written as index arithmetic on the
    // lane id, hipcc materialises (and then spills) one address VGPR per unrolled access.
    const unsigned char* l0B = smem + LY::l0 + half * 64;      // float4 per feature, features 4*half..
    const unsigned char* woutB = smem + LY::wout + half * 16;  // float per feature
    const unsigned char* zeroB = smem + LY::zero + half * 16;
    const unsigned char* biasB = smem + LY::bias + half * 16;
    float* modT = reinterpret_cast<float*>(smem + LY::mods(L)) + wave * (L * 256);
    const unsigned char* modB = reinterpret_cast<const unsigned char*>(modT) + half * 16;

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
    // Passes (4 units each) are handed out through a device-wide counter, not by a fixed stride: with two
    // launches in flight on different streams a workgroup may start late on a CU the previous launch has
    // just released, and then simply takes fewer passes -- every CU stays busy until the queue is empty.
    volatile int* qslot = reinterpret_cast<volatile int*>(smem + LY::queue(L));
    // The per-layer inverse scales are indexed at run time.  Read straight from the kernel-argument segment
    // (host-visible memory) every such s_load that misses the scalar cache costs microseconds -- measured:
    // ~13k cycles at every layer boundary.  They are copied to LDS once instead.
    float* winvT = reinterpret_cast<float*>(smem + LY::winv(L));
    if (tid < 16) winvT[tid] = p.winv[tid];
    int cur_pass = (int)blockIdx.x;
    // each wave moves its 8 KB slice of a chunk: 8 x 1 KB global_load_lds_dwordx4, one base address
    // pair (biased by +4 KB so that the eight 1 KB steps fit the 13-bit signed immediate, which the
    // instruction applies to the global AND the LDS address)
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp) + wave * 8192 + lane * 16 + 4096;
    int dma_id = 0, dma_buf = 0, rd_buf = 0;
    // One chunk = 8 pieces of 1 KB per wave.  dma_next() issues them as a block (prologue only); inside a
    // tile they are spread over the four scheduling groups after the ring barrier (MSIREN_DMA_PIECE), so
    // the MFMA stream is not interrupted by a block of eight address/M0/DMA issues.
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
    // the kept patches only: known on the device (scalar: a vector register here costs the co-residency of §4.3)
    const int total_units = __builtin_amdgcn_readfirstlane(p.plan ? p.plan[1] : p.total_units);
    // unsigned compare: a pass id that came out negative (host/device counter disagreement) ends the workgroup
    // instead of indexing out of bounds
    const unsigned npasses = (unsigned)(total_units + 3) >> 2;
    if ((unsigned)cur_pass >= npasses) return;
#pragma unroll
    for (int s = 0; s < R - 1; ++s) dma_next();

    h8 Xh[16], Xl[16], Yh[16], Yl[16];
    f32x16 acc[2];
    float part = 0.f;

    // epilogue of one 32-feature tile, in four parts (g = 0..3: features 8g..8g+7 of the tile, 4 per
    // half-wave): acc -> (revolutions) -> activation -> modulation -> fp16 split; parts 0,1 make up
    // k-step 2t and parts 2,3 k-step 2t+1 of the next layer's B operand.  Also accumulates last_layer's
    // dot product with `wo` (the zero table on all but the final hidden layer).
    fp16x2 eh[4][2], el[4][2];
    // bias / modulation / last_layer weight of the epilogue part in flight and of the next one
    // (two register sets, loaded one scheduling group ahead of their use)
    f32x4 tb_b[2], tb_m[2], tb_w[2];
    auto tbl_load = [&](int set, const unsigned char* bl, const unsigned char* ml, const unsigned char* wo, int t, int g,
                        bool withw) {
        const int fo = (32 * t + 8 * g) * 4;  // compile-time byte offset
        tb_b[set] = *reinterpret_cast<const f32x4*>(bl + fo);
        tb_m[set] = *reinterpret_cast<const f32x4*>(ml + fo);
        if (withw) tb_w[set] = *reinterpret_cast<const f32x4*>(wo + fo);
    };
    // half `hh` (elements 2hh, 2hh+1) of part g of an accumulator tile: acc -> (revolutions) -> activation
    // -> modulation, then either (lastl = false) the fp16 hi/lo pair for the next layer's B operand, or
    // (lastl = true: final hidden layer, separate code instance) last_layer's dot product -- no split,
    // no pack, no AGPR store there, and no dot-product FMA anywhere else.
    auto epi_half = [&](const f32x16& a, float winv, float cgl, int set, int g, int hh, bool lastl) {
        // The two accumulator elements pass through an opaque asm: instruction selection orders pure
        // VALU code only by data dependence, so without an anchor the whole tile's epilogue is emitted
        // in one block ahead of the MFMAs and the sched_barrier-delimited groups are empty of VALU.
        float a0 = a[4 * g + 2 * hh], a1 = a[4 * g + 2 * hh + 1];
        asm volatile("; epilogue slice anchored to its MFMA group" : "+v"(a0), "+v"(a1));
        const float ain[2] = {a0, a1};
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float r = __builtin_fmaf(ain[e], winv, tb_b[set][2 * hh + e]);
            v[e] = activate<ACT>(r, cgl) * tb_m[set][2 * hh + e];
            if (lastl) part = __builtin_fmaf(v[e], tb_w[set][2 * hh + e], part);
        }
        if (!lastl) {
            const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
            eh[g][hh] = h;
            el[g][hh] = __builtin_amdgcn_cvt_pkrtz(v[0] - (float)h[0], v[1] - (float)h[1]);
        }
    };
    auto epi_store2 = [&](int ks, h8& dh, h8& dl) {  // k-step ks (0/1) of the tile = parts 2ks, 2ks+1
        dh = to_acc_file(pack_h8(eh[2 * ks][0], eh[2 * ks][1], eh[2 * ks + 1][0], eh[2 * ks + 1][1]));
        dl = to_acc_file(pack_h8(el[2 * ks][0], el[2 * ks][1], el[2 * ks + 1][0], el[2 * ks + 1][1]));
    };
    h8 wf_[2][4];  // weight fragments of the k-step group in flight / the next one

    // The tile / group / k-step structure is expanded by the preprocessor, not by `#pragma unroll`:
    // with asm statements in the body hipcc only partially unrolls, the activation arrays then get
    // runtime indices and are demoted to scratch memory.
#define MSIREN_F16_KSTEP(INh, INl, T, Q, J)                                                   \
    do {                                                                                      \
        if (2 * (Q) + (J) == 0) mfma_f16_first(acc[(T) & 1], wf_[(Q) & 1][2 * (J) + 1], INh[2 * (Q) + (J)]); \
        else mfma_f16_acc(acc[(T) & 1], wf_[(Q) & 1][2 * (J) + 1], INh[2 * (Q) + (J)]);       \
        mfma_f16_acc(acc[(T) & 1], wf_[(Q) & 1][2 * (J)], INl[2 * (Q) + (J)]);                \
        mfma_f16_acc(acc[(T) & 1], wf_[(Q) & 1][2 * (J)], INh[2 * (Q) + (J)]);                \
    } while (0)

    // Group Q of tile T = one scheduling region: 2 k-steps (6 MFMAs, 192 cycles), the LDS reads of the
    // NEXT group's weight fragments (for Q == 7: the next tile's first group, from the next ring
    // buffer, which the mid-tile barrier has already published), one slice of the previous tile's
    // epilogue, and the table reads of the epilogue part after that.
    // Epilogue schedule.  T > 0: tile T-1, half (Q&1) of part Q>>1 per group; T == 0: the previous layer's
    // tile 7 ("pending"), whose result feeds k-steps 14, 15 of THIS tile, so: parts 0..3 in groups 0..3,
    // stores in groups 4 and 5.
// Ablation builds (timing only, results wrong; never shipped): -DMSIREN_ABL=bitmask
//   1 = no epilogue work in the groups, 2 = no ring barrier / vmcnt wait, 4 = no weight-fragment LDS reads
#ifndef MSIREN_ABL
#define MSIREN_ABL 0
#endif
#ifndef MSIREN_SGB_VARIANT
#define MSIREN_SGB_VARIANT 1
#endif
#if MSIREN_SGB_VARIANT == 0
#define MSIREN_F16_SGB() do {} while (0)
#elif MSIREN_SGB_VARIANT == 1
#define MSIREN_F16_SGB()                                                                      \
    do {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                    \
    } while (0)
#else  /* 2: three VALU after every MFMA, the DS reads up front */
#define MSIREN_F16_SGB()                                                                      \
    do {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                    \
    } while (0)
#endif

#define MSIREN_F16_GROUP(INh, INl, OUTh, OUTl, T, Q, LASTF)                                       \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if ((Q) >= 4) { /* two of the eight DMA pieces of chunk c+R-1 per group */            \
            MSIREN_DMA_PIECE(2 * ((Q) & 3));                                                  \
            MSIREN_DMA_PIECE(2 * ((Q) & 3) + 1);                                              \
        }                                                                                     \
        if (MSIREN_ABL & 4) { /* fragments stay what they are, opaquely */                   \
            asm volatile("" : "+v"(wf_[((Q) + 1) & 1][0]), "+v"(wf_[((Q) + 1) & 1][1]), "+v"(wf_[((Q) + 1) & 1][2]), "+v"(wf_[((Q) + 1) & 1][3])); \
        } else {                                                                              \
            const h8* src_ = (Q) < 7 ? ring_ + (4 * (((Q) + 1) & 7)) * 64 : ringn_;           \
            wf_[((Q) + 1) & 1][0] = src_[0 * 64];                                             \
            wf_[((Q) + 1) & 1][1] = src_[1 * 64];                                             \
            wf_[((Q) + 1) & 1][2] = src_[2 * 64];                                             \
            wf_[((Q) + 1) & 1][3] = src_[3 * 64];                                             \
        }                                                                                     \
        if (MSIREN_ABL & 1) { /* keep the accumulators alive so the MFMAs are not dead code */  \
            if ((Q) == 0) asm volatile("" ::"v"(acc[((T) + 1) & 1]));                         \
        } else if ((T) == 0) {                                                                \
            if ((Q) < 3) tbl_load(((Q) + 1) & 1, blp_, mlp_, zeroB, 7, ((Q) + 1) & 3, false); \
            if ((Q) < 4) {                                                                    \
                epi_half(acc[1], wip_, cgp_, (Q) & 1, (Q) & 3, 0, false);                     \
                epi_half(acc[1], wip_, cgp_, (Q) & 1, (Q) & 3, 1, false);                     \
            }                                                                                 \
            if ((Q) == 4) epi_store2(0, INh[14], INl[14]);                                    \
            if ((Q) == 5) epi_store2(1, INh[15], INl[15]);                                    \
        } else {                                                                              \
            if (((Q) & 1) == 1 && (Q) < 7) tbl_load((((Q) >> 1) + 1) & 1, bl_, ml_, wo_, ((T) + 7) & 7, (((Q) >> 1) + 1) & 3, LASTF); \
            epi_half(acc[((T) + 1) & 1], wi_, p.cg, ((Q) >> 1) & 1, (Q) >> 1, (Q) & 1, LASTF); \
            if ((Q) == 5 && !(LASTF)) epi_store2(0, OUTh[(2 * (T) + 14) & 15], OUTl[(2 * (T) + 14) & 15]); \
        }                                                                                     \
        if ((Q) == 7 && !(MSIREN_ABL & 1)) tbl_load(0, bl_, ml_, wo_, (T), 0, LASTF); /* part 0 of THIS tile's epilogue (runs next tile) */ \
        MSIREN_F16_KSTEP(INh, INl, T, Q, 0);                                                  \
        MSIREN_F16_KSTEP(INh, INl, T, Q, 1);                                                  \
        /* requested issue order inside the region: the weight-fragment reads first, VALU spread */ \
        MSIREN_F16_SGB();                                                                     \
    } while (0)

    // One tile = one 32 KB weight chunk.  The ring is synchronised in the MIDDLE of the tile: by then
    // every wave has finished the previous tile (so its buffer may be refilled: DMA of chunk c+R-1) and,
    // after the counted vmcnt + barrier, chunk c+1 is visible to all -- early enough for group 7 to
    // prefetch the next tile's first weight fragments, so no LDS latency is exposed at tile boundaries.
#define MSIREN_F16_TILE(INh, INl, OUTh, OUTl, T, LASTF)                                           \
    do {                                                                                      \
        const h8* ring_ = reinterpret_cast<const h8*>(smem + LY::ring + rd_buf * F16_CHUNK_BYTES) + lane; \
        rd_buf = rd_buf + 1 == R ? 0 : rd_buf + 1;                                            \
        const h8* ringn_ = reinterpret_cast<const h8*>(smem + LY::ring + rd_buf * F16_CHUNK_BYTES) + lane; \
        MSIREN_F16_GROUP(INh, INl, OUTh, OUTl, T, 0, LASTF);                                    \
        MSIREN_F16_GROUP(INh, INl, OUTh, OUTl, T, 1, LASTF);                                    \
        MSIREN_F16_GROUP(INh, INl, OUTh, OUTl, T, 2, LASTF);                                    \
        MSIREN_F16_GROUP(INh, INl, OUTh, OUTl, T, 3, LASTF);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (!(MSIREN_ABL & 2)) {                                                              \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 8) : "memory");                \
            __builtin_amdgcn_s_barrier();                                                     \
        }                                                                                     \
        dma_begin();                                                                          \
        MSIREN_F16_GROUP(INh, INl, OUTh, OUTl, T, 4, LASTF);                                    \
        MSIREN_F16_GROUP(INh, INl, OUTh, OUTl, T, 5, LASTF);                                    \
        MSIREN_F16_GROUP(INh, INl, OUTh, OUTl, T, 6, LASTF);                                    \
        MSIREN_F16_GROUP(INh, INl, OUTh, OUTl, T, 7, LASTF);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if ((T) > 0 && !(LASTF) && !(MSIREN_ABL & 1)) epi_store2(1, OUTh[(2 * (T) + 15) & 15], OUTl[(2 * (T) + 15) & 15]); \
        if constexpr (DBG) { stamp(8 + dbg_tile); ++dbg_tile; }                               \
    } while (0)

    // one hidden layer: IN -> OUT.  The previous layer's last tile still sits in acc[1]; its epilogue
    // (which produces IN[14], IN[15]) is issued inside this layer's first tile.  For l_ == 1 the
    // "previous layer" is layer 0, whose arguments were left in acc[1] in revolutions: inverse scale 1,
    // no bias, layer-0 modulation and Morlet constant.
#define MSIREN_F16_LAYER(INh, INl, OUTh, OUTl, LIDX, LASTF)                                       \
    do {                                                                                      \
        const int l_ = (LIDX);                                                                \
        const unsigned char* wo_ = woutB; /* read by the final-layer instance only */         \
        const unsigned char* bl_ = biasB + (l_ - 1) * 1024;                                   \
        const unsigned char* ml_ = modB + l_ * 1024;                                          \
        const unsigned char* blp_ = l_ > 1 ? biasB + (l_ - 2) * 1024 : zeroB;                 \
        const unsigned char* mlp_ = modB + (l_ - 1) * 1024;                                   \
        const float wi_ = winvT[l_ - 1], wip_ = l_ > 1 ? winvT[l_ - 2] : 1.0f;              \
        const float cgp_ = l_ > 1 ? p.cg : p.cg0;                                             \
        MSIREN_F16_TILE(INh, INl, OUTh, OUTl, 0, LASTF);                                        \
        MSIREN_F16_TILE(INh, INl, OUTh, OUTl, 1, LASTF);                                        \
        MSIREN_F16_TILE(INh, INl, OUTh, OUTl, 2, LASTF);                                        \
        MSIREN_F16_TILE(INh, INl, OUTh, OUTl, 3, LASTF);                                        \
        MSIREN_F16_TILE(INh, INl, OUTh, OUTl, 4, LASTF);                                        \
        MSIREN_F16_TILE(INh, INl, OUTh, OUTl, 5, LASTF);                                        \
        MSIREN_F16_TILE(INh, INl, OUTh, OUTl, 6, LASTF);                                        \
        MSIREN_F16_TILE(INh, INl, OUTh, OUTl, 7, LASTF);                                        \
    } while (0)

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 8) : "memory");
    __syncthreads();  // tables + first chunk visible
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
        unit = active ? unit : total_units - 1;
        const int b = unit / p.units_per_patch;
        const int cu = unit - b * p.units_per_patch;
        int pc = cu * 32 + c32;
        const bool pvalid = active && pc < p.P;
        pc = pc < p.P ? pc : p.P - 1;

        // the next pass id is fetched a whole pass ahead, together with the loads below (one wait)
        int nxt = 0;
        if (tid == 0) nxt = (int)((unsigned)atomicAdd(p.pass_counter, 1) - p.pass_base) + (int)gridDim.x;
        // this wave's modulation table: (L, 256) floats of patch b
        for (int l = 0; l < L; ++l) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(p.mods + ((size_t)l * p.B + b) * 256 + lane * 4);
            *reinterpret_cast<f32x4*>(modT + l * 256 + lane * 4) = m;
        }
        if (tid == 0) qslot[(pass + 1) & 1] = nxt;  // read after >= 32 workgroup barriers
        const float2 xy = reinterpret_cast<const float2*>(p.grid)[pc];

        // ---- layer 0 (K = 2) directly in B-operand order: element j of k-step s is feature
        //      32*(s>>1) + 16*(s&1) + 8*(j>>2) + 4*half + (j&3).  K-steps 0..13 are finished here; the
        //      last 32 features (k-steps 14, 15 = "tile 7") are left as sine ARGUMENTS in acc[1], where
        //      the first hidden layer's pending-epilogue slot turns them into X[14], X[15].
        // The unmodulated layer-0 activations depend on the pixel only, not on the patch: they come from
        // a table built once per weight set (fp64 on the host), so layer 0 costs one coalesced 16-byte
        // load + 4 multiplies + the fp16 split per four features instead of 4 x (2 FMA + sine).
        const f32x4* s0 = reinterpret_cast<const f32x4*>(p.s0t) + (size_t)half * p.P + pc;
        f32x4 raw[14][2];  // all 28 loads in flight at once: one L2 latency per pass instead of seven
#pragma unroll
        for (int s = 0; s < 14; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int f0 = 32 * (s >> 1) + 16 * (s & 1) + 8 * q;  // + 4*half via the base
                raw[s][q] = s0[(size_t)(f0 / 4) * p.P];
            }
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            fp16x2 hh[2][2], ll[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int f0 = 32 * (s >> 1) + 16 * (s & 1) + 8 * q;
                const f32x4 m4 = *reinterpret_cast<const f32x4*>(modB + f0 * 4);
                split4(raw[s][q] * m4, hh[q][0], hh[q][1], ll[q][0], ll[q][1]);
            }
            Xh[s] = to_acc_file(pack_h8(hh[0][0], hh[0][1], hh[1][0], hh[1][1]));
            Xl[s] = to_acc_file(pack_h8(ll[0][0], ll[0][1], ll[1][0], ll[1][1]));
        }
        {
            f32x16 r7;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(l0B + (224 + 8 * g + e) * 16);
                    r7[4 * g + e] = __builtin_fmaf(xy.y, w[1], __builtin_fmaf(xy.x, w[0], w[2]));
                }
            acc[1] = r7;
        }
        tbl_load(0, zeroB, modB, zeroB, 7, 0, false);  // part 0 of the layer-0 "pending" tile

        part = 0.f;
        stamp(1);
        // Hidden layers alternate X->Y (instance A) and Y->X (instance B); the final hidden layer has its own
        // instances (no fp16 split / AGPR store, the only place the last_layer dot product is accumulated),
        // one per input array.  Straight-line data flow: no merge points for the register-resident arrays.
        for (int l = 1;;) {
            if (l == L - 1) {
                MSIREN_F16_LAYER(Xh, Xl, Yh, Yl, l, true);
                break;
            }
            MSIREN_F16_LAYER(Xh, Xl, Yh, Yl, l, false);
            ++l;
            if (l == L - 1) {
                MSIREN_F16_LAYER(Yh, Yl, Xh, Xl, l, true);
                break;
            }
            MSIREN_F16_LAYER(Yh, Yl, Xh, Xl, l, false);
            ++l;
        }
        stamp(2);
        // the final hidden layer's last tile is still pending: its contribution to `part`
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g > 0) tbl_load(g & 1, biasB + (L - 2) * 1024, modB + (L - 1) * 1024, woutB, 7, g, true);
            epi_half(acc[1], winvT[L - 2], p.cg, g & 1, g, 0, true);
            epi_half(acc[1], winvT[L - 2], p.cg, g & 1, g, 1, true);
        }
        part += __shfl_xor(part, 32);
        if (pvalid && half == 0) p.out[(size_t)b * p.P + pc] = sin_rev(part + p.bout);
        cur_pass = __builtin_amdgcn_readfirstlane(qslot[(pass + 1) & 1]);
        stamp(6);
        stamp(7);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may be in flight when the LDS is released
#undef MSIREN_F16_LAYER
#undef MSIREN_F16_TILE
#undef MSIREN_F16_GROUP
#undef MSIREN_F16_KSTEP
#undef MSIREN_DMA_PIECE
}

}  // namespace msiren
