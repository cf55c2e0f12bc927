// Fused trunk, single-product 16-bit variant for H = 512 (BASELINE config 5), WEIGHT-STATIONARY form ("x1w").
//
// Same maths as siren_trunk_x1n.hip.h (this build's own residual definition, parity unpinned against the reference):
//     x_{l+1} = x_l + mod_l * act(W_l x_l + b_l)   for l >= 1      (layer 0 and last_layer unchanged)
// another data flow.  Why: the register-resident kernel is bound by the LDS, not by its MFMAs -- every wave re-reads every
// weight fragment for its own 32 coordinates, 1 KB per 32 MFMA cycles and wave = the LDS's whole 128 B/clk at full MFMA
// rate (rocprofv3: LDS data path ~77 % busy, matrix pipe 50 %; profiles/r4/04_config5_x1n_vs_x1_ab.txt).  Here, as in
// siren_trunk_f16x3w.hip.h:
//   * a wave owns 64 OUTPUT FEATURES and keeps their weights -- all 512 input features of them, 16 k-steps x 4 tiles x 4
//     registers -- in the accumulator half of the register file, a[0:255] BY NAME (asm loads, asm MFMAs); 4 waves = 256
//     features, so a layer is two N-PASSES over the same input;
//   * the ACTIVATIONS go through LDS: a unit (32 coordinates of one patch) is a 32 KB image of ready-made B fragments
//     [16 k-steps][2 column groups][64 lanes][8 x 16 bit]; all four waves read it (2 x ds_read_b128 per 8 MFMAs: a quarter of
//     the LDS bytes per MFMA of the register-resident kernel) and each wave writes the features it produced back IN PLACE.
//     Both N-passes of a layer read the same image, so the first pass's outputs wait in registers (16 per unit) until the
//     second pass has read it;
//   * a workgroup takes a PASS of 4 units through the layers; slot = (layer, N-pass, unit); the next (layer, N-pass)'s
//     weights are fetched from L2 straight into the fragment registers as the last unit's MFMAs retire them (64 x
//     global_load_dwordx4 per wave, waited for k-step by k-step in the next slot);
//   * the slot is the unit of software pipelining: slot n's 128 MFMAs run beside the epilogue of slot n - 1 -- a value at a
//     time (sine, residual, modulation, pack), one step per MFMA gap, plain C++ statements fenced into their gaps by
//     sched_barriers (the accumulators alternate between two register sets by the unit's parity, so no copies);  a first
//     form with the epilogue BEHIND its MFMAs was correct and 8-13 % slower than the register-resident kernel
//     (profiles/r4/05_*; the source: tools/experiments/siren_trunk_x1w_lite.hip.h of commit 27d6e80).
#pragma once
#include <hip/hip_runtime.h>

#include "siren_trunk_x1n.hip.h"  // TrunkX1Params, x1_pack2 / x1_unpack2, sum_over_q, vector types

// Ablation builds (timing only, results wrong; never shipped): -DMSIREN_X1W_ABL=bitmask -- 1 no epilogue steps in the gaps,
// 2 no slot barrier, 4 no MFMAs, 8 every layer reads layer 1's weights (an L2-resident weight stream)
#ifndef MSIREN_X1W_ABL
#define MSIREN_X1W_ABL 0
#endif
namespace msiren {

struct X1wLds {  // byte offsets into dynamic LDS
    static constexpr int act = 0;                    // 4 unit images of 32 KB
    static constexpr int wout = 4 * 32768;           // 512 x fp16
    static constexpr int bias = wout + 1024;         // (L-1) x 512 x fp32 (initial value of the accumulators)
    static __host__ __device__ constexpr int mods(int L) { return bias + (L - 1) * 2048; }  // 2 layer parities x 4 units x 512 x fp16
    static __host__ __device__ constexpr int red(int L) { return mods(L) + 8192; }          // 4 units x 4 waves x 32 floats
    static __host__ __device__ constexpr int queue(int L) { return red(L) + 2048; }
    static __host__ __device__ constexpr int winv(int L) { return queue(L) + 16; }
    static __host__ __device__ constexpr int total(int L) { return winv(L) + 256; }
};

struct X1wSchedule {
    int n4, n2;
};
__host__ __device__ inline X1wSchedule x1w_schedule(long long units, int grid) {
    X1wSchedule s{0, 0};
    if (units <= 0 || grid <= 0) return s;
    const long long full = units / (4LL * grid);
    s.n4 = (int)(full * grid);
    const long long rem = units - 4LL * s.n4;  // < 4 * grid
    if (rem <= 2LL * grid) s.n2 = (int)((rem + 1) / 2);
    else s.n4 += (int)((rem + 3) / 4);
    return s;
}

// Host: the SMALLEST grid (down to 7/8 of the CUs) that needs no more rounds than all CUs would.  A workgroup owns its CU (512
// registers, 158 KB LDS), so whatever the launch leaves free is where the other stream's encoder / Modulator kernels of the next
// call run meanwhile: one 320x320 slice = 450 passes = two rounds on 256 CUs (the second 3/4 full) and also on 225 -- the same
// launch time, and 31 CUs for the neighbour instead of a queue behind the trunk.
inline int x1w_balanced_grid(long long units, int cus) {
    const long long passes4 = (units + 3) / 4;
    if (passes4 <= cus) return (int)(passes4 < 1 ? 1 : passes4);
    auto half_rounds = [&](int g) {
        const X1wSchedule s = x1w_schedule(units, g);
        return 2LL * ((s.n4 + g - 1) / g) + (s.n2 + g - 1) / g;
    };
    const long long c0 = half_rounds(cus);
    int best = cus;
    for (int g = cus - 1; g >= cus - cus / 8; --g)
        if (half_rounds(g) <= c0) best = g;
    return best;
}

template <int BF, int ACT, int RES>
__global__ __launch_bounds__(256, 1) void siren_trunk_x1w_kernel(TrunkX1Params p) {
    using LY = X1wLds;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;     // which four features of a 16-feature tile this lane holds
    const int n16 = lane & 15;   // coordinate inside a 16-column group
    const int L = p.L;
    const int P = p.P;

    const int total_units = __builtin_amdgcn_readfirstlane(p.plan ? p.plan[1] : p.total_units);
    // passes [0, n4) take 4 units, [n4, n4 + n2) take 2 (x1w_schedule: whole rounds of 4-unit passes; what is left, if it is at
    // most two units per workgroup, as one round of 2-unit passes -- half a round instead of a whole one at a launch's end)
    const X1wSchedule sch = x1w_schedule(total_units, (int)gridDim.x);
    const unsigned npasses = (unsigned)(sch.n4 + sch.n2);
    int cur_pass = (int)blockIdx.x;
    if ((unsigned)cur_pass >= npasses) return;

    unsigned char* const actL = smem + LY::act + lane * 16;  // + unit * 32768 + (2 * k-step + column group) * 1024
    const unsigned char* const biasL = smem + LY::bias + wave * 256 + q * 16;   // + (l - 1) * 2048 + n * 1024 + t * 64
    const unsigned char* const woutL = smem + LY::wout + wave * 128 + q * 8;    // + n * 512 + t * 32
    unsigned char* const modsS = smem + LY::mods(L);                            // [(l & 1) * 4 + unit][512] fp16
    const unsigned char* const modsL = modsS + wave * 128 + q * 8;              // + ((l & 1) * 4 + unit) * 1024 + n * 512 + t * 32
    float* const redT = reinterpret_cast<float*>(smem + LY::red(L));
    volatile int* qslot = reinterpret_cast<volatile int*>(smem + LY::queue(L));
    float* const winvT = reinterpret_cast<float*>(smem + LY::winv(L));

    {   // constant tables
        _Float16* wow = reinterpret_cast<_Float16*>(smem + LY::wout);
        float* bw = reinterpret_cast<float*>(smem + LY::bias);
        for (int i = tid; i < 512; i += 256) wow[i] = p.wout[i];
        for (int i = tid; i < (L - 1) * 512; i += 256) bw[i] = p.bias32[i];
        if (tid < 64) winvT[tid] = p.winv[tid];
    }

    // ---- the weights of the (layer, N-pass) in flight: A fragment (k-step s, tile t) = a[16 s + 4 t .. + 3], BY NAME ---------
    // Stream: block ((l - 1) * 2 + n) * 4 + wave of 64 KB = [16 k-steps][4 tiles][64 lanes][8 x 16 bit].
    const unsigned woff = (unsigned)lane * 16u;
    auto wblock = [&](int l, int n) -> const unsigned char* {
        return reinterpret_cast<const unsigned char*>(p.wp) + ((size_t)(((MSIREN_X1W_ABL & 8) ? 0 : l - 1) * 2 + n) * 4 + wave) * 65536;  // (ablation 8: every layer reads layer 1's 512 KB -- an L2-resident weight stream)
    };
#define MSIREN_X1W_A(S, T) (16 * (S) + 4 * (T))
// (s_nop 4: a VALU-written SGPR -- a pointer the compiler spilled and restores with v_readlane_b32 -- may be read by a vector
// memory instruction only 5 wait states later, and the compiler does not pad in front of an asm statement.  The fp16 instances
// spill pointers the bf16 ones keep; without the pad their first load after such a restore went to a stale address and the
// launch died with an aperture violation -- found with rocgdb's precise-memory mode, kept out by tests/test_asm_hazards.py.)
#define MSIREN_X1W_LOADK(S, WB)                                                                                        \
    asm volatile("s_nop 4\n\t"                                                                                         \
                 "global_load_dwordx4 a[%2:%3], %0, %1 offset:0\n\t"                                                   \
                 "global_load_dwordx4 a[%4:%5], %0, %1 offset:1024\n\t"                                                \
                 "global_load_dwordx4 a[%6:%7], %0, %1 offset:2048\n\t"                                                \
                 "global_load_dwordx4 a[%8:%9], %0, %1 offset:3072"                                                    \
                 :                                                                                                     \
                 : "v"(woff), "s"((WB) + (S) * 4096), "n"(MSIREN_X1W_A(S, 0)), "n"(MSIREN_X1W_A(S, 0) + 3),            \
                   "n"(MSIREN_X1W_A(S, 1)), "n"(MSIREN_X1W_A(S, 1) + 3), "n"(MSIREN_X1W_A(S, 2)), "n"(MSIREN_X1W_A(S, 2) + 3), \
                   "n"(MSIREN_X1W_A(S, 3)), "n"(MSIREN_X1W_A(S, 3) + 3)                                                \
                 : "memory")
// k-step S's fragments have landed once at most 4 (15 - S) younger loads are outstanding (loads return in order)
#define MSIREN_X1W_WAITK(S) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * (15 - (S))) : "memory")
#define MSIREN_X1W_MFMA(ACC, S, T, B)                                                                                  \
    do {                                                                                                               \
        if constexpr (BF)                                                                                              \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%2:%3], %1, %0" : "+v"(ACC) : "v"(B), "n"(MSIREN_X1W_A(S, T)), "n"(MSIREN_X1W_A(S, T) + 3)); \
        else                                                                                                           \
            asm volatile("v_mfma_f32_16x16x32_f16 %0, a[%2:%3], %1, %0" : "+v"(ACC) : "v"(B), "n"(MSIREN_X1W_A(S, T)), "n"(MSIREN_X1W_A(S, T) + 3)); \
    } while (0)

// a slot's first k-step: the accumulator starts from the bias rows (C operand), no copy
#define MSIREN_X1W_MFMA0(ACC, S, T, B, C)                                                                              \
    do {                                                                                                               \
        if constexpr (BF)                                                                                              \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%3:%4], %1, %2" : "=v"(ACC) : "v"(B), "v"(C), "n"(MSIREN_X1W_A(S, T)), "n"(MSIREN_X1W_A(S, T) + 3)); \
        else                                                                                                           \
            asm volatile("v_mfma_f32_16x16x32_f16 %0, a[%3:%4], %1, %2" : "=v"(ACC) : "v"(B), "v"(C), "n"(MSIREN_X1W_A(S, T)), "n"(MSIREN_X1W_A(S, T) + 3)); \
    } while (0)

    // Keeping the register allocator OUT of the accumulator file: 64 placeholder values of AGPR class, defined here and used
    // behind the pass loop, keep all 256 AGPRs allocated as far as the compiler can tell (which placeholder sits in which
    // register is irrelevant: the statements above name the registers themselves; -amdgpu-spill-vgpr-to-agpr=0).
    asm volatile("; a[0:255] weight fragments" ::: "a255");
    h8 wres[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("; placeholder" : "=a"(wres[i]));
#define MSIREN_X1W_HOLD()                                                                                              \
    do {                                                                                                               \
        asm volatile("" ::"a"(wres[0]), "a"(wres[1]), "a"(wres[2]), "a"(wres[3]), "a"(wres[4]), "a"(wres[5]), "a"(wres[6]), "a"(wres[7]),     \
                     "a"(wres[8]), "a"(wres[9]), "a"(wres[10]), "a"(wres[11]), "a"(wres[12]), "a"(wres[13]), "a"(wres[14]), "a"(wres[15]),    \
                     "a"(wres[16]), "a"(wres[17]), "a"(wres[18]), "a"(wres[19]), "a"(wres[20]), "a"(wres[21]), "a"(wres[22]), "a"(wres[23])); \
        asm volatile("" ::"a"(wres[24]), "a"(wres[25]), "a"(wres[26]), "a"(wres[27]), "a"(wres[28]), "a"(wres[29]), "a"(wres[30]), "a"(wres[31]), \
                     "a"(wres[32]), "a"(wres[33]), "a"(wres[34]), "a"(wres[35]), "a"(wres[36]), "a"(wres[37]), "a"(wres[38]), "a"(wres[39]), \
                     "a"(wres[40]), "a"(wres[41]), "a"(wres[42]), "a"(wres[43]), "a"(wres[44]), "a"(wres[45]), "a"(wres[46]), "a"(wres[47])); \
        asm volatile("" ::"a"(wres[48]), "a"(wres[49]), "a"(wres[50]), "a"(wres[51]), "a"(wres[52]), "a"(wres[53]), "a"(wres[54]), "a"(wres[55]), \
                     "a"(wres[56]), "a"(wres[57]), "a"(wres[58]), "a"(wres[59]), "a"(wres[60]), "a"(wres[61]), "a"(wres[62]), "a"(wres[63])); \
    } while (0)

    {   // prologue: the first (layer 1, N-pass 0) weights
        const unsigned char* wb = wblock(1, 0);
        MSIREN_X1W_LOADK(0, wb); MSIREN_X1W_LOADK(1, wb); MSIREN_X1W_LOADK(2, wb); MSIREN_X1W_LOADK(3, wb);
        MSIREN_X1W_LOADK(4, wb); MSIREN_X1W_LOADK(5, wb); MSIREN_X1W_LOADK(6, wb); MSIREN_X1W_LOADK(7, wb);
        MSIREN_X1W_LOADK(8, wb); MSIREN_X1W_LOADK(9, wb); MSIREN_X1W_LOADK(10, wb); MSIREN_X1W_LOADK(11, wb);
        MSIREN_X1W_LOADK(12, wb); MSIREN_X1W_LOADK(13, wb); MSIREN_X1W_LOADK(14, wb); MSIREN_X1W_LOADK(15, wb);
    }
    __syncthreads();  // constant tables visible

    u32x4 held[4][2][2];  // N-pass 0's outputs of the four units, waiting for N-pass 1 to have read the image: [unit][k-step parity][column group]
    float part[4][2];     // last_layer dot product: [unit][column group]
    f32x4 accS[2][4][2];  // two accumulator sets: slot (l, N, U) accumulates into set U & 1 while the set of the slot before is worked off
    // the epilogue in flight (of the slot BEFORE the one whose MFMAs are being issued): a value at a time, one step per MFMA gap
    u32x4 Bf[2][2];  // B fragments of k-steps S (parity S & 1): the next k-step's are read while this one's MFMAs run -- and a slot's
                     // first ones during the last k-step of the slot before (PF / HAVE below)
    f32x4 btmp[4];   // bias rows of the slot's (layer, N-pass), the C operand of its first MFMAs; fetched during the slot before
#pragma unroll
    for (int i = 0; i < 2; ++i) Bf[i][0] = Bf[i][1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < 4; ++t) btmp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float es_ = 0.f, ex_ = 0.f, ev_[2] = {0.f, 0.f};
    u32x4 eold_[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, enf_[2][2];  // (the next group's residual fragment is fetched while this group's is in use)
    hf4 em_[4], ew_[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) em_[t] = ew_[t] = hf4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            enf_[i][j] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < 4; ++t) accS[i][t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // ---- the epilogue of slot (lp, NP, UP) as 128 steps, step J issued behind MFMA J of the NEXT slot ----------------------
    // value vi = J >> 2 = 16 kk + 8 g + 4 sub + e (output tile t = 2 kk + sub of column group g, element e; its place in the
    // image: k-step 8 NP + 2 wave + kk, fragment word 2 sub + (e >> 1)); step J & 3:
    //   0  sine of the accumulator (the bias went in as the MFMAs' C operand; fp16 instance: x 2^-e first)
    //   1  the residual: the layer's input at the same place (bf16 -> fp32)
    //   2  activation x modulation + residual (one v_fma_mix); final layer: + the dot product with last_layer.weight
    //   3  odd e: pack the pair (N-pass 0: into `held`, N-pass 1: into the fragment that goes back in place);
    //      even e: one of the slot's 16 LDS operations -- the residual fragments and modulation rows a few steps ahead of
    //      their first use, the in-place stores of finished fragments (N-pass 1 only; `wr_`: not in a pass's first slot,
    //      whose "slot before" does not exist)
    // Every step sits between two sched_barriers: it stays in its gap.
#define MSIREN_X1W_OLD(KK, G, NP) eold_[(G) & 1] = *reinterpret_cast<const u32x4*>(pimg_ + (2 * (8 * (NP) + 2 * wave + (KK)) + (G)) * 1024)
#define MSIREN_X1W_LDM(T, NP, LASTP)                                                                                        \
    do {                                                                                                               \
        em_[T] = *reinterpret_cast<const hf4*>(pmr_ + (T) * 32);                                                       \
        if (LASTP) ew_[T] = *reinterpret_cast<const hf4*>(woutL + (NP) * 512 + (T) * 32);                              \
    } while (0)
#define MSIREN_X1W_STEP(J, NP, UP, LASTP, PFE)                                                                         \
    do {                                                                                                               \
        constexpr int vi_ = (J) >> 2, st_ = (J) & 3;                                                                   \
        constexpr int e_ = vi_ & 3, sub_ = (vi_ >> 2) & 1, g_ = (vi_ >> 3) & 1, kk_ = vi_ >> 4, t_ = 2 * kk_ + sub_;   \
        if constexpr (st_ == 0) {                                                                                      \
            float r_ = accS[((UP) & 1)][t_][g_][e_];                                                                   \
            if constexpr (!BF) r_ *= pwi_;                                                                             \
            if constexpr (ACT == 0) asm volatile("v_sin_f32 %0, %1" : "=v"(es_) : "v"(r_));                            \
            else es_ = activate<ACT>(r_, p.cg);                                                                        \
        } else if constexpr (st_ == 1) {                                                                               \
            if constexpr (RES) {                                                                                       \
                float x0_, x1_;                                                                                        \
                x1_unpack2<BF>(eold_[g_][2 * sub_ + (e_ >> 1)], x0_, x1_);                                             \
                ex_ = (e_ & 1) ? x1_ : x0_;                                                                            \
            }                                                                                                          \
        } else if constexpr (st_ == 2) {                                                                               \
            if constexpr (RES) ev_[e_ & 1] = __builtin_fmaf(es_, (float)em_[t_][e_], ex_);                             \
            else ev_[e_ & 1] = es_ * (float)em_[t_][e_];                                                               \
            if (LASTP) {                                                                                               \
                part[UP][g_] = __builtin_fmaf(ev_[e_ & 1], (float)ew_[t_][e_], part[UP][g_]);                          \
                /* pinned here: the sum is only read at the end of the pass, and left alone the compiler sinks the whole    \
                   chain of multiply-adds down there -- keeping all 32 sines of the slot alive, i.e. spilling them */    \
                asm volatile("" : "+v"(part[UP][g_]));                                                                 \
            }                                                                                                          \
        } else if constexpr ((e_ & 1) == 1) {                                                                          \
            if (!(LASTP)) {                                                                                            \
                if ((NP) == 0) held[UP][kk_][g_][2 * sub_ + (e_ >> 1)] = x1_pack2<BF>(ev_[0], ev_[1]);                 \
                else enf_[kk_][g_][2 * sub_ + (e_ >> 1)] = x1_pack2<BF>(ev_[0], ev_[1]);                               \
            }                                                                                                          \
        } else {                                                                                                       \
            constexpr bool wr_ok_ = !(LASTP) && (NP) == 1;                                                             \
            if constexpr ((J) == 3) MSIREN_X1W_LDM(1, NP, LASTP);                                                                 \
            if constexpr ((J) == 19 && RES) MSIREN_X1W_OLD(0, 1, NP);                                                      \
            if constexpr ((J) == 43) MSIREN_X1W_LDM(2, NP, LASTP);                                                                \
            if constexpr ((J) == 51 && RES) MSIREN_X1W_OLD(1, 0, NP);                                                      \
            if constexpr ((J) == 59) MSIREN_X1W_LDM(3, NP, LASTP);                                                                \
            if constexpr ((J) == 83 && RES) MSIREN_X1W_OLD(1, 1, NP);                                                      \
            /* PFE: what the NEXT slot's gaps need first (the epilogue of THIS slot): its residual fragment (0, 0) and      \
               modulation row 0 -- their registers are free from steps 95 / 47 on */                                       \
            if constexpr ((PFE) && (J) == 107 && RES) eold_[0] = *reinterpret_cast<const u32x4*>(img_ + (2 * (8 * nN_ + 2 * wave + 0) + 0) * 1024); \
            if constexpr ((PFE) && (J) == 115) {                                                                           \
                em_[0] = *reinterpret_cast<const hf4*>(npmr_);                                                             \
                if (nLAST_) ew_[0] = *reinterpret_cast<const hf4*>(woutL + nN_ * 512);                                     \
            }                                                                                                              \
            if constexpr (wr_ok_) {                                                                                    \
                if (wr_) {                                                                                             \
                    if constexpr ((J) == 11) *reinterpret_cast<u32x4*>(pimg_ + (2 * (2 * wave + 0) + 0) * 1024) = held[UP][0][0]; \
                    if constexpr ((J) == 27) *reinterpret_cast<u32x4*>(pimg_ + (2 * (2 * wave + 0) + 1) * 1024) = held[UP][0][1]; \
                    if constexpr ((J) == 75) *reinterpret_cast<u32x4*>(pimg_ + (2 * (2 * wave + 1) + 0) * 1024) = held[UP][1][0]; \
                    if constexpr ((J) == 91) *reinterpret_cast<u32x4*>(pimg_ + (2 * (2 * wave + 1) + 1) * 1024) = held[UP][1][1]; \
                    if constexpr ((J) == 35) *reinterpret_cast<u32x4*>(pimg_ + (2 * (8 + 2 * wave + 0) + 0) * 1024) = enf_[0][0]; \
                    if constexpr ((J) == 67) *reinterpret_cast<u32x4*>(pimg_ + (2 * (8 + 2 * wave + 0) + 1) * 1024) = enf_[0][1]; \
                    if constexpr ((J) == 99) *reinterpret_cast<u32x4*>(pimg_ + (2 * (8 + 2 * wave + 1) + 0) * 1024) = enf_[1][0]; \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)
// what a slot does before its first MFMA for the epilogue that rides in its gaps, and behind its last one
#define MSIREN_X1W_EPI_BEGIN(NP, UP, LASTP)                                                                            \
    do {                                                                                                               \
        if constexpr (RES) MSIREN_X1W_OLD(0, 0, NP);                                                                       \
        MSIREN_X1W_LDM(0, NP, LASTP);                                                                                             \
    } while (0)
#define MSIREN_X1W_EPI_END(NP, UP, LASTP)                                                                              \
    do {                                                                                                               \
        if (!(LASTP) && (NP) == 1 && wr_) *reinterpret_cast<u32x4*>(pimg_ + (2 * (8 + 2 * wave + 1) + 1) * 1024) = enf_[1][1]; \
    } while (0)

    // One slot = (layer l, N-pass N, unit U); N and U compile-time; (NP, UP, LASTP): the slot before it, whose epilogue rides in
    // this slot's gaps.  U == 0 waits for the fragments fetched during the slot before, U == 3 fetches the next (layer, N-pass)'s
    // behind the MFMAs that retire them.
// Ablation builds (timing only, results wrong; never shipped): -DMSIREN_X1W_ABL=bitmask -- 1 = no epilogue steps in the gaps,
// 2 = no slot barrier, 4 = no MFMAs
#define MSIREN_X1W_M(S, T, G, I, N, U, NP, UP, LASTP, GAPS, PF)                                                        \
    do {                                                                                                               \
        if constexpr ((S) == 0) {                                                                                      \
            if (!(MSIREN_X1W_ABL & 4)) MSIREN_X1W_MFMA0(accS[(U) & 1][T][G], S, T, Bf[(S) & 1][G], btmp[T]);           \
            else asm volatile("" : "=v"(accS[(U) & 1][T][G]) : "v"(Bf[(S) & 1][G]), "v"(btmp[T]));                     \
        } else {                                                                                                       \
            if (!(MSIREN_X1W_ABL & 4)) MSIREN_X1W_MFMA(accS[(U) & 1][T][G], S, T, Bf[(S) & 1][G]);                     \
            else asm volatile("" : "+v"(accS[(U) & 1][T][G]) : "v"(Bf[(S) & 1][G]));                                   \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if ((GAPS) && !(MSIREN_X1W_ABL & 1)) MSIREN_X1W_STEP(8 * (S) + (I), NP, UP, LASTP, PF);                        \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)
#define MSIREN_X1W_KSTEP(N, U, S, NP, UP, LASTP, GAPS, NB, PF)                                                         \
    do {                                                                                                               \
        MSIREN_X1W_HOLD();                                                                                             \
        if ((U) == 0) MSIREN_X1W_WAITK(S);                                                                             \
        if ((S) < 15) {                                                                                                \
            Bf[((S) + 1) & 1][0] = *reinterpret_cast<const u32x4*>(img_ + (2 * ((S) + 1)) * 1024);                     \
            Bf[((S) + 1) & 1][1] = *reinterpret_cast<const u32x4*>(img_ + (2 * ((S) + 1) + 1) * 1024);                 \
        } else if (PF) { /* the next slot's first fragments (its image is not being stored into: see MSIREN_X1W_LAYER) */ \
            Bf[0][0] = *reinterpret_cast<const u32x4*>(simg_);                                                         \
            Bf[0][1] = *reinterpret_cast<const u32x4*>(simg_ + 1024);                                                  \
        }                                                                                                              \
        if ((PF) && (S) == 8) { /* the next slot's bias rows (this slot's were consumed by k-step 0) */                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) btmp[t] = *reinterpret_cast<const f32x4*>(sbias_ + t * 64);  \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        MSIREN_X1W_M(S, 0, 0, 0, N, U, NP, UP, LASTP, GAPS, PF); MSIREN_X1W_M(S, 0, 1, 1, N, U, NP, UP, LASTP, GAPS, PF);          \
        MSIREN_X1W_M(S, 1, 0, 2, N, U, NP, UP, LASTP, GAPS, PF); MSIREN_X1W_M(S, 1, 1, 3, N, U, NP, UP, LASTP, GAPS, PF);          \
        MSIREN_X1W_M(S, 2, 0, 4, N, U, NP, UP, LASTP, GAPS, PF); MSIREN_X1W_M(S, 2, 1, 5, N, U, NP, UP, LASTP, GAPS, PF);          \
        MSIREN_X1W_M(S, 3, 0, 6, N, U, NP, UP, LASTP, GAPS, PF); MSIREN_X1W_M(S, 3, 1, 7, N, U, NP, UP, LASTP, GAPS, PF);          \
        if ((U) == (NB) - 1) MSIREN_X1W_LOADK(S, wnext_);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)

// LAST (compile-time): the final hidden layer -- its outputs only meet last_layer.weight (a dot product per coordinate), no
// image is written.  LASTP: the same for the slot before.  lp_: that slot's layer (l, or l - 1 for a layer's first slot).
// HAVE: what the slot needs first -- its bias rows, its first B fragments, and (GAPS) the first residual fragment and modulation
// row of the epilogue it carries -- was fetched during the slot before (1; 2: if l > 1, a pass's first slot has no slot before);
// PF: it does that for the slot after it.  PF(n) == HAVE(n + 1).  Without it every slot started with an LDS round trip in
// front of its first MFMA and another in front of its second epilogue step.
#define MSIREN_X1W_SLOT(N, U, LAST, NP, UP, LASTP, GAPS, BAR, NB, HAVE, PF)                                             \
    do {                                                                                                               \
        /* the slot after this one: (N, U + 1), or (1 - N, 0) -- of the next layer behind N-pass 1 */                   \
        constexpr int sU_ = (U) < (NB) - 1 ? (U) + 1 : 0, sN_ = (U) < (NB) - 1 ? (N) : 1 - (N);                         \
        constexpr int sDL_ = ((U) == (NB) - 1 && (N) == 1) ? 1 : 0;                                                    \
        /* (opaque: as constants the units' offsets are folded into dozens of loop-invariant address registers) */     \
        unsigned uoff_ = (U) * 32768u, poff_ = (UP) * 32768u, soff_ = sU_ * 32768u;                                    \
        asm volatile("" : "+s"(uoff_), "+s"(poff_), "+s"(soff_));                                                      \
        unsigned char* const img_ = actL + uoff_;                                                                      \
        unsigned char* const pimg_ = actL + poff_;                                                                     \
        [[maybe_unused]] const unsigned char* const simg_ = actL + soff_;                                              \
        [[maybe_unused]] const unsigned char* const sbias_ = biasL + (l - 1 + sDL_) * 2048 + sN_ * 1024;               \
        /* the epilogue of THIS slot rides in the next one's gaps: what it reads first (MSIREN_X1W_STEP, PFE) */        \
        [[maybe_unused]] constexpr int nN_ = (N);                                                                      \
        [[maybe_unused]] constexpr bool nLAST_ = (LAST);                                                               \
        [[maybe_unused]] const unsigned char* const npmr_ = modsL + ((l & 1) * 4 + (U)) * 1024 + (N) * 512;            \
        const int lp_ = ((N) == 0 && (U) == 0) ? l - 1 : l;                                                            \
        const bool wr_ = lp_ >= 1; /* (a pass's first slot: nothing before it) */                                      \
        const unsigned char* const pmr_ = modsL + ((lp_ & 1) * 4 + (UP)) * 1024 + (NP) * 512;                          \
        [[maybe_unused]] const float pwi_ = winvT[lp_ > 0 ? lp_ - 1 : 0];                                              \
        /* what the unit's last slot fetches: (l, 1) behind (l, 0); (l + 1, 0) behind (l, 1); layer 1 of the next pass at the end */ \
        const unsigned char* wnext_ = (N) == 0 ? wblock(l, 1) : wblock((LAST) ? 1 : l + 1, 0);                         \
        asm volatile("" : "+s"(wnext_)); /* (opaque: the 16 k-step pointers of layer 1 are not kept in 32 SGPRs for the whole kernel) */ \
        if (BAR) {                                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
            if (!(MSIREN_X1W_ABL & 2)) __builtin_amdgcn_s_barrier(); /* every wave is past the MFMAs of the slot before: its image may be updated in \
                                             place; what earlier slots' epilogues stored is complete */                \
        }                                                                                                              \
        if (!((HAVE) == 1 || ((HAVE) == 2 && l > 1))) {                                                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                              \
                btmp[t] = *reinterpret_cast<const f32x4*>(biasL + (l - 1) * 2048 + (N) * 1024 + t * 64);               \
            Bf[0][0] = *reinterpret_cast<const u32x4*>(img_);                                                          \
            Bf[0][1] = *reinterpret_cast<const u32x4*>(img_ + 1024);                                                   \
            if (GAPS) MSIREN_X1W_EPI_BEGIN(NP, UP, LASTP);                                                             \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        MSIREN_X1W_KSTEP(N, U, 0, NP, UP, LASTP, GAPS, NB, PF); MSIREN_X1W_KSTEP(N, U, 1, NP, UP, LASTP, GAPS, NB, PF);            \
        MSIREN_X1W_KSTEP(N, U, 2, NP, UP, LASTP, GAPS, NB, PF); MSIREN_X1W_KSTEP(N, U, 3, NP, UP, LASTP, GAPS, NB, PF);            \
        MSIREN_X1W_KSTEP(N, U, 4, NP, UP, LASTP, GAPS, NB, PF); MSIREN_X1W_KSTEP(N, U, 5, NP, UP, LASTP, GAPS, NB, PF);            \
        MSIREN_X1W_KSTEP(N, U, 6, NP, UP, LASTP, GAPS, NB, PF); MSIREN_X1W_KSTEP(N, U, 7, NP, UP, LASTP, GAPS, NB, PF);            \
        MSIREN_X1W_KSTEP(N, U, 8, NP, UP, LASTP, GAPS, NB, PF); MSIREN_X1W_KSTEP(N, U, 9, NP, UP, LASTP, GAPS, NB, PF);            \
        MSIREN_X1W_KSTEP(N, U, 10, NP, UP, LASTP, GAPS, NB, PF); MSIREN_X1W_KSTEP(N, U, 11, NP, UP, LASTP, GAPS, NB, PF);          \
        MSIREN_X1W_KSTEP(N, U, 12, NP, UP, LASTP, GAPS, NB, PF); MSIREN_X1W_KSTEP(N, U, 13, NP, UP, LASTP, GAPS, NB, PF);          \
        MSIREN_X1W_KSTEP(N, U, 14, NP, UP, LASTP, GAPS, NB, PF); MSIREN_X1W_KSTEP(N, U, 15, NP, UP, LASTP, GAPS, NB, PF);          \
        if (GAPS) MSIREN_X1W_EPI_END(NP, UP, LASTP);                                                                   \
    } while (0)
// A final-layer slot's own epilogue, BEHIND its MFMAs and on the accumulator set it has just filled: sine, residual, modulation
// and the dot product with last_layer.weight -- no pack, no store.  (Ridden in the next slot's gaps like the others it needs
// 16 registers more than a normal slot's -- last_layer.weight and the partial sums -- which the register file does not have:
// the compiler spilled ~40 values per final slot and every reload waited out a memory round trip.  Eight slots of 72 pay ~700
// un-overlapped cycles each instead.)
#define MSIREN_X1W_FINAL_EPI(N, U)                                                                                     \
    do {                                                                                                               \
        unsigned poff_ = (U) * 32768u;                                                                                 \
        asm volatile("" : "+s"(poff_));                                                                                \
        unsigned char* const pimg_ = actL + poff_;                                                                     \
        [[maybe_unused]] unsigned char* const img_ = pimg_; /* (names MSIREN_X1W_STEP mentions under PFE, never true here) */ \
        [[maybe_unused]] constexpr int nN_ = 0;                                                                        \
        [[maybe_unused]] constexpr bool nLAST_ = false;                                                                \
        [[maybe_unused]] const unsigned char* const npmr_ = modsL;                                                     \
        [[maybe_unused]] const bool wr_ = false;                                                                       \
        const unsigned char* const pmr_ = modsL + ((l & 1) * 4 + (U)) * 1024 + (N) * 512;                              \
        [[maybe_unused]] const float pwi_ = winvT[l - 1];                                                              \
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); /* the last MFMAs (asm) are still writing the accumulators read below */ \
        MSIREN_X1W_EPI_BEGIN(N, U, true);                                                                              \
        MSIREN_X1W_F16(0, N, U); MSIREN_X1W_F16(16, N, U); MSIREN_X1W_F16(32, N, U); MSIREN_X1W_F16(48, N, U);         \
        MSIREN_X1W_F16(64, N, U); MSIREN_X1W_F16(80, N, U); MSIREN_X1W_F16(96, N, U); MSIREN_X1W_F16(112, N, U);       \
    } while (0)
// (fenced step by step: left alone the compiler issues all 32 asm sines first -- volatile statements keep their order, the rest
// sinks below them -- and spills their results)
#define MSIREN_X1W_F1(J, N, U) MSIREN_X1W_STEP(J, N, U, true, false); __builtin_amdgcn_sched_barrier(0)
#define MSIREN_X1W_F4(J, N, U) MSIREN_X1W_F1(J, N, U); MSIREN_X1W_F1(J + 1, N, U); MSIREN_X1W_F1(J + 2, N, U); MSIREN_X1W_F1(J + 3, N, U)
#define MSIREN_X1W_F16(J, N, U) MSIREN_X1W_F4(J, N, U); MSIREN_X1W_F4(J + 4, N, U); MSIREN_X1W_F4(J + 8, N, U); MSIREN_X1W_F4(J + 12, N, U)
// the eight slots of a layer; LASTP0: whether the slot before the layer's first one belongs to a final layer (never: a pass
// ends with its final layer, whose last epilogue is worked off behind the loop)
// Which slots need the workgroup barrier at their start.  (a) The epilogue in the slot's gaps stores into the image of the slot
// before (N-pass 1 epilogues only): every wave must be past that slot's MFMAs -- slots (0,0) [carrying (l-1, 1, 3)] and (1,1),
// (1,2), (1,3).  (b) An image is read only after every wave's stores into it are complete: unit U's image is stored into
// during slot (1, U+1) -- or (0,0) of the next layer for U = 3 -- and read next in slot (0, U) of the next layer: the
// barriers of (a) lie in between for U = 0..2, and the one at (0,1) covers U = 3.  Slots (0,2), (0,3), (1,0) need none.
#define MSIREN_X1W_LAYER()                                                                                             \
    do {                                                                                                               \
        MSIREN_X1W_SLOT(0, 0, false, 1, 3, false, true, true, 4, 2, true);                                             \
        MSIREN_X1W_SLOT(0, 1, false, 0, 0, false, true, true, 4, 1, true);                                             \
        stage_mods(l + 1); /* (no asm load is in flight here: the compiler's own wait drains nothing of ours) */       \
        MSIREN_X1W_SLOT(0, 2, false, 0, 1, false, true, false, 4, 1, true);                                            \
        MSIREN_X1W_SLOT(0, 3, false, 0, 2, false, true, false, 4, 1, true);                                            \
        MSIREN_X1W_SLOT(1, 0, false, 0, 3, false, true, false, 4, 1, true);                                            \
        MSIREN_X1W_SLOT(1, 1, false, 1, 0, false, true, true, 4, 1, true);                                             \
        MSIREN_X1W_SLOT(1, 2, false, 1, 1, false, true, true, 4, 1, true);                                             \
        MSIREN_X1W_SLOT(1, 3, false, 1, 2, false, true, true, 4, 1, true);                                             \
    } while (0)
// the final hidden layer: its first slot still carries the layer before's last epilogue in its gaps; nothing is stored
// into an image any more (barriers: the two that order the last in-place stores against their readers)
#define MSIREN_X1W_FINAL_LAYER()                                                                                       \
    do {                                                                                                               \
        MSIREN_X1W_SLOT(0, 0, true, 1, 3, false, true, true, 4, 1, false);    MSIREN_X1W_FINAL_EPI(0, 0);              \
        MSIREN_X1W_SLOT(0, 1, true, 0, 0, true, false, true, 4, 0, false);    MSIREN_X1W_FINAL_EPI(0, 1);              \
        MSIREN_X1W_SLOT(0, 2, true, 0, 1, true, false, false, 4, 0, false);   MSIREN_X1W_FINAL_EPI(0, 2);              \
        MSIREN_X1W_SLOT(0, 3, true, 0, 2, true, false, false, 4, 0, false);   MSIREN_X1W_FINAL_EPI(0, 3);              \
        MSIREN_X1W_SLOT(1, 0, true, 0, 3, true, false, false, 4, 0, false);   MSIREN_X1W_FINAL_EPI(1, 0);              \
        MSIREN_X1W_SLOT(1, 1, true, 1, 0, true, false, false, 4, 0, false);   MSIREN_X1W_FINAL_EPI(1, 1);              \
        MSIREN_X1W_SLOT(1, 2, true, 1, 1, true, false, false, 4, 0, false);   MSIREN_X1W_FINAL_EPI(1, 2);              \
        MSIREN_X1W_SLOT(1, 3, true, 1, 2, true, false, false, 4, 0, false);   MSIREN_X1W_FINAL_EPI(1, 3);              \
    } while (0)

// The same for a pass of TWO units (slots (0,0) (0,1) (1,0) (1,1); the second unit's slots fetch).  Barriers: (a) (0,0) [carries
// (l-1, 1, 1)] and (1,1) [carries (1,0)]; (b) unit 1's image is stored into during (0,0) of the next layer and read in (0,1).
#define MSIREN_X1W_LAYER2()                                                                                            \
    do {                                                                                                               \
        MSIREN_X1W_SLOT(0, 0, false, 1, 1, false, true, true, 2, 0, false);                                            \
        MSIREN_X1W_SLOT(0, 1, false, 0, 0, false, true, true, 2, 0, false);                                            \
        MSIREN_X1W_SLOT(1, 0, false, 0, 1, false, true, false, 2, 0, false);                                           \
        stage_mods(l + 1); /* (behind a slot that has waited for every weight load: none of ours is in flight) */      \
        MSIREN_X1W_SLOT(1, 1, false, 1, 0, false, true, true, 2, 0, false);                                            \
    } while (0)
#define MSIREN_X1W_FINAL_LAYER2()                                                                                      \
    do {                                                                                                               \
        MSIREN_X1W_SLOT(0, 0, true, 1, 1, false, true, true, 2, 0, false);    MSIREN_X1W_FINAL_EPI(0, 0);              \
        MSIREN_X1W_SLOT(0, 1, true, 0, 0, true, false, true, 2, 0, false);    MSIREN_X1W_FINAL_EPI(0, 1);              \
        MSIREN_X1W_SLOT(1, 0, true, 0, 1, true, false, false, 2, 0, false);   MSIREN_X1W_FINAL_EPI(1, 0);              \
        MSIREN_X1W_SLOT(1, 1, true, 1, 0, true, false, false, 2, 0, false);   MSIREN_X1W_FINAL_EPI(1, 1);              \
    } while (0)

    for (int pass = 0; (unsigned)cur_pass < npasses; ++pass) {
        // ---- the pass's four units (clamped into the batch; surplus units are computed and not stored) ----
        const int nb = cur_pass < sch.n4 ? 4 : 2;
        const int u0 = cur_pass < sch.n4 ? 4 * cur_pass : 4 * sch.n4 + 2 * (cur_pass - sch.n4);
        int patch[4], c0[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int unit = u0 + (u < nb ? u : 0);
            live[u] = u < nb && unit < total_units;
            unit = live[u] ? unit : total_units - 1;
            patch[u] = unit / p.units_per_patch;
            c0[u] = (unit - patch[u] * p.units_per_patch) * 32;
        }
        int nxt = 0;
        if (tid == 0) nxt = (int)((unsigned)atomicAdd(p.pass_counter, 1) - p.pass_base) + (int)gridDim.x;
        // modulation rows of layers 0 and 1: wave u stages unit u's (512 floats -> fp16, 8 per lane)
        auto stage_mods = [&](int l) {
            const float* src = p.mods + ((size_t)l * p.B + patch[0]) * 512;
            if (wave == 1) src = p.mods + ((size_t)l * p.B + patch[1]) * 512;
            if (wave == 2) src = p.mods + ((size_t)l * p.B + patch[2]) * 512;
            if (wave == 3) src = p.mods + ((size_t)l * p.B + patch[3]) * 512;
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(src + lane * 8);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(src + lane * 8 + 4);
            u32x4 hm;
            hm[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{m0[0], m0[1]}, hf2));
            hm[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{m0[2], m0[3]}, hf2));
            hm[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{m1[0], m1[1]}, hf2));
            hm[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{m1[2], m1[3]}, hf2));
            *reinterpret_cast<u32x4*>(modsS + ((l & 1) * 4 + wave) * 1024 + lane * 16) = hm;
        };
        stage_mods(0);
        if (L > 1) stage_mods(1);
        if (tid == 0) qslot[pass & 1] = nxt;
        __syncthreads();  // layer-0 modulation rows visible (and: every wave is out of the pass before)

        // ---- layer 0 (K = 2) from the per-weight-set table act0(W0 x_p + b0), x modulation, straight into the images: wave w
        //      writes k-steps 4 i + w (i = 0..3) of every unit
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int pc0 = c0[u] + n16, pc1 = c0[u] + 16 + n16;
            pc0 = pc0 < P ? pc0 : P - 1;
            pc1 = pc1 < P ? pc1 : P - 1;
            const f32x4* s0a = reinterpret_cast<const f32x4*>(p.s0t) + (size_t)q * P + pc0;
            const f32x4* s0b = reinterpret_cast<const f32x4*>(p.s0t) + (size_t)q * P + pc1;
            f32x4 raw[4][2][2];  // [i][column group][sub]
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int ks = 4 * i + wave;
                    raw[i][0][sub] = s0a[(size_t)(8 * ks + 4 * sub) * P];
                    raw[i][1][sub] = s0b[(size_t)(8 * ks + 4 * sub) * P];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ks = 4 * i + wave;
                hf4 m4[2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    m4[sub] = *reinterpret_cast<const hf4*>(modsS + (0 * 4 + u) * 1024 + (32 * ks + 16 * sub + 4 * q) * 2);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    u32x4 w4;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const f32x4 a = raw[i][g][sub];
                        w4[2 * sub] = x1_pack2<BF>(a[0] * (float)m4[sub][0], a[1] * (float)m4[sub][1]);
                        w4[2 * sub + 1] = x1_pack2<BF>(a[2] * (float)m4[sub][2], a[3] * (float)m4[sub][3]);
                    }
                    *reinterpret_cast<u32x4*>(actL + u * 32768 + (2 * ks + g) * 1024) = w4;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) part[u][0] = part[u][1] = 0.f;

        // ---- hidden layers ----
        int l = 1;
        if (nb == 4) {
            for (; l < L - 1; ++l) MSIREN_X1W_LAYER();
            MSIREN_X1W_FINAL_LAYER();  // l == L - 1
        } else {
            for (; l < L - 1; ++l) MSIREN_X1W_LAYER2();
            MSIREN_X1W_FINAL_LAYER2();
        }

        // ---- last_layer: sum over the lane's feature sub-groups, over the waves through LDS, sine, store ----
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s0v = sum_over_q(part[u][0]), s1v = sum_over_q(part[u][1]);
            if (q < 2) redT[(u * 4 + wave) * 32 + q * 16 + n16] = q == 0 ? s0v : s1v;
        }
        __syncthreads();
        if (tid < 128) {
            const int u = tid >> 5, c = tid & 31;
            const float s = (redT[(u * 4 + 0) * 32 + c] + redT[(u * 4 + 1) * 32 + c]) + (redT[(u * 4 + 2) * 32 + c] + redT[(u * 4 + 3) * 32 + c]);
            int pu = patch[0], cu = c0[0];
            bool lv = live[0];
            if (u == 1) { pu = patch[1]; cu = c0[1]; lv = live[1]; }
            if (u == 2) { pu = patch[2]; cu = c0[2]; lv = live[2]; }
            if (u == 3) { pu = patch[3]; cu = c0[3]; lv = live[3]; }
            const float o_ = sin_rev(s + p.bout);
            if (lv && cu + c < P) {
                p.out[(size_t)pu * P + cu + c] = o_;
                if constexpr (!BF) {
                    if (!(__builtin_fabsf(o_) <= 2.f) && p.status) *p.status = p.status_val;  // NaN: the fp16 domain was left (or the input was NaN)
                }
            }
        }
        cur_pass = __builtin_amdgcn_readfirstlane(qslot[pass & 1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the weights fetched for a pass that does not come)
    MSIREN_X1W_HOLD();
#undef MSIREN_X1W_HOLD
#undef MSIREN_X1W_LAYER
#undef MSIREN_X1W_LAYER2
#undef MSIREN_X1W_FINAL_LAYER2
#undef MSIREN_X1W_FINAL_LAYER
#undef MSIREN_X1W_FINAL_EPI
#undef MSIREN_X1W_F16
#undef MSIREN_X1W_F4
#undef MSIREN_X1W_F1
#undef MSIREN_X1W_SLOT
#undef MSIREN_X1W_M
#undef MSIREN_X1W_STEP
#undef MSIREN_X1W_EPI_BEGIN
#undef MSIREN_X1W_EPI_END
#undef MSIREN_X1W_OLD
#undef MSIREN_X1W_LDM
#undef MSIREN_X1W_KSTEP
#undef MSIREN_X1W_MFMA
#undef MSIREN_X1W_WAITK
#undef MSIREN_X1W_LOADK
#undef MSIREN_X1W_A
}

}  // namespace msiren
