// RECORD ONLY -- not part of the product build (profiles/r4/05_config5_weight_stationary_ab.txt: correct, 8-13 % slower than the
// register-resident kernel because its epilogue is not overlapped with MFMAs; the starting point for an overlapped version).
// It built against mri_inr_amd/csrc at the commit that added this file: msiren.hip packed its weight stream as
// block ((l - 1) * 2 + n) * 4 + wave = [16 k-steps][4 tiles][64 lanes][8] (output feature 256 n + 64 wave + 16 t + (lane & 15),
// input feature 32 s + 16 (j >> 2) + 4 (lane >> 4) + (j & 3)) and launched it with X1wLds::total(L) bytes of LDS, one workgroup per CU.
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
//   * simpler than its split-fp16 sibling on purpose: a slot's epilogue (sine, modulation + residual, pack, LDS stores) runs
//     BEHIND its MFMAs, compiler-scheduled, not interleaved with the next slot's -- the matrix pipe idles meanwhile, and
//     the kernel is still ahead because the LDS no longer throttles the MFMA phase.
#pragma once
#include <hip/hip_runtime.h>

#include "../../mri_inr_amd/csrc/siren_trunk_x1n.hip.h"  // TrunkX1Params, x1_pack2 / x1_unpack2, sum_over_q, vector types

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
    const unsigned npasses = (unsigned)(total_units + 3) >> 2;
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
        return reinterpret_cast<const unsigned char*>(p.wp) + ((size_t)((l - 1) * 2 + n) * 4 + wave) * 65536;
    };
#define MSIREN_X1W_A(S, T) (16 * (S) + 4 * (T))
#define MSIREN_X1W_LOADK(S, WB)                                                                                        \
    asm volatile("global_load_dwordx4 a[%2:%3], %0, %1 offset:0\n\t"                                                   \
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

    // One slot = (layer l, N-pass N, unit U); N and U compile-time.  FL: U == 0 waits for the fragments fetched during the slot
    // before, U == 3 fetches the next (layer, N-pass)'s behind the MFMAs that retire them.
#define MSIREN_X1W_KSTEP(N, U, S)                                                                                      \
    do {                                                                                                               \
        MSIREN_X1W_HOLD();                                                                                             \
        if ((U) == 0) MSIREN_X1W_WAITK(S);                                                                             \
        if ((S) < 15) {                                                                                                \
            Bf[((S) + 1) & 1][0] = *reinterpret_cast<const u32x4*>(img_ + (2 * ((S) + 1)) * 1024);                     \
            Bf[((S) + 1) & 1][1] = *reinterpret_cast<const u32x4*>(img_ + (2 * ((S) + 1) + 1) * 1024);                 \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        MSIREN_X1W_MFMA(acc[0][0], S, 0, Bf[(S) & 1][0]); MSIREN_X1W_MFMA(acc[0][1], S, 0, Bf[(S) & 1][1]);            \
        MSIREN_X1W_MFMA(acc[1][0], S, 1, Bf[(S) & 1][0]); MSIREN_X1W_MFMA(acc[1][1], S, 1, Bf[(S) & 1][1]);            \
        MSIREN_X1W_MFMA(acc[2][0], S, 2, Bf[(S) & 1][0]); MSIREN_X1W_MFMA(acc[2][1], S, 2, Bf[(S) & 1][1]);            \
        MSIREN_X1W_MFMA(acc[3][0], S, 3, Bf[(S) & 1][0]); MSIREN_X1W_MFMA(acc[3][1], S, 3, Bf[(S) & 1][1]);            \
        if ((U) == 3) MSIREN_X1W_LOADK(S, wnext_);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)

// LAST (compile-time): the final hidden layer -- its outputs only meet last_layer.weight (a dot product per coordinate), no
// image is written.  (As a run-time condition the compiler computed the dot product in every layer and selected.)
#define MSIREN_X1W_SLOT(N, U, LAST)                                                                                    \
    do {                                                                                                               \
        /* (opaque: as a constant the unit's offset is folded into dozens of loop-invariant address registers, one per      \
           64 KB-offset window and unit, and the register file overflows) */                                           \
        unsigned uoff_ = (U) * 32768u;                                                                                 \
        asm volatile("" : "+s"(uoff_));                                                                                \
        unsigned char* const img_ = actL + uoff_;                                                                      \
        /* what the unit's last slot fetches: (l, 1) behind (l, 0); (l + 1, 0) behind (l, 1); layer 1 of the next pass at the end */ \
        const unsigned char* const wnext_ = (N) == 0 ? wblock(l, 1) : wblock((LAST) ? 1 : l + 1, 0);                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
        __builtin_amdgcn_s_barrier(); /* the images written behind earlier slots are complete */                       \
        f32x4 acc[4][2];                                                                                               \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                \
            acc[t][0] = *reinterpret_cast<const f32x4*>(biasL + (l - 1) * 2048 + (N) * 1024 + t * 64);                 \
            acc[t][1] = acc[t][0];                                                                                     \
        }                                                                                                              \
        u32x4 Bf[2][2];                                                                                                \
        Bf[0][0] = *reinterpret_cast<const u32x4*>(img_);                                                              \
        Bf[0][1] = *reinterpret_cast<const u32x4*>(img_ + 1024);                                                       \
        asm volatile("s_nop 1" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1])); \
        MSIREN_X1W_KSTEP(N, U, 0); MSIREN_X1W_KSTEP(N, U, 1); MSIREN_X1W_KSTEP(N, U, 2); MSIREN_X1W_KSTEP(N, U, 3);    \
        MSIREN_X1W_KSTEP(N, U, 4); MSIREN_X1W_KSTEP(N, U, 5); MSIREN_X1W_KSTEP(N, U, 6); MSIREN_X1W_KSTEP(N, U, 7);    \
        MSIREN_X1W_KSTEP(N, U, 8); MSIREN_X1W_KSTEP(N, U, 9); MSIREN_X1W_KSTEP(N, U, 10); MSIREN_X1W_KSTEP(N, U, 11);  \
        MSIREN_X1W_KSTEP(N, U, 12); MSIREN_X1W_KSTEP(N, U, 13); MSIREN_X1W_KSTEP(N, U, 14); MSIREN_X1W_KSTEP(N, U, 15); \
        /* the last MFMAs (asm: the compiler pads nothing behind them) are still writing the accumulators */           \
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1])); \
        /* ---- epilogue: this wave's 64 features of unit U = k-steps 8 N + 2 wave + kk of the image ---- */           \
        const unsigned char* const mr_ = modsL + ((l & 1) * 4 + (U)) * 1024 + (N) * 512;                               \
        const float wi_ = winvT[l - 1];                                                                                \
        u32x4 nf_[2][2];                                                                                               \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                               \
        _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                \
            u32x4 old_ = {0u, 0u, 0u, 0u};                                                                             \
            if constexpr (RES) old_ = *reinterpret_cast<const u32x4*>(img_ + (2 * (8 * (N) + 2 * wave + kk) + g) * 1024); \
            _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) {                                                      \
                const int t = 2 * kk + sub;                                                                            \
                const hf4 m_ = *reinterpret_cast<const hf4*>(mr_ + t * 32);                                            \
                hf4 w_ = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};                                 \
                if (LAST) w_ = *reinterpret_cast<const hf4*>(woutL + (N) * 512 + t * 32);                              \
                float v_[4];                                                                                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
                    float r_ = acc[t][g][e];                                                                           \
                    if constexpr (!BF) r_ *= wi_;                                                                      \
                    const float s_ = activate<ACT>(r_, p.cg);                                                          \
                    if constexpr (RES) {                                                                               \
                        float x0_, x1_;                                                                                \
                        x1_unpack2<BF>(old_[2 * sub + (e >> 1)], x0_, x1_);                                            \
                        v_[e] = __builtin_fmaf(s_, (float)m_[e], (e & 1) ? x1_ : x0_);                                 \
                    } else {                                                                                           \
                        v_[e] = s_ * (float)m_[e];                                                                     \
                    }                                                                                                  \
                    if (LAST) part[U][g] = __builtin_fmaf(v_[e], (float)w_[e], part[U][g]);                            \
                }                                                                                                      \
                if (!(LAST)) {                                                                                         \
                    nf_[kk][g][2 * sub] = x1_pack2<BF>(v_[0], v_[1]);                                                  \
                    nf_[kk][g][2 * sub + 1] = x1_pack2<BF>(v_[2], v_[3]);                                              \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
        if (!(LAST)) {                                                                                                 \
            if ((N) == 0) {                                                                                            \
                _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                       \
                _Pragma("unroll") for (int g = 0; g < 2; ++g) held[U][kk][g] = nf_[kk][g];                             \
            } else {                                                                                                   \
                /* every wave is past its last read of the image: both passes' outputs go back in place */            \
                __builtin_amdgcn_s_barrier();                                                                          \
                _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                       \
                _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                        \
                    *reinterpret_cast<u32x4*>(img_ + (2 * (2 * wave + kk) + g) * 1024) = held[U][kk][g];               \
                    *reinterpret_cast<u32x4*>(img_ + (2 * (8 + 2 * wave + kk) + g) * 1024) = nf_[kk][g];               \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)

    for (int pass = 0; (unsigned)cur_pass < npasses; ++pass) {
        // ---- the pass's four units (clamped into the batch; surplus units are computed and not stored) ----
        int patch[4], c0[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int unit = cur_pass * 4 + u;
            live[u] = unit < total_units;
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
        for (; l < L - 1; ++l) {
            MSIREN_X1W_SLOT(0, 0, false);
            MSIREN_X1W_SLOT(0, 1, false);
            // the next layer's modulation rows (no asm load is in flight here: the compiler's own wait drains nothing of ours)
            stage_mods(l + 1);
            MSIREN_X1W_SLOT(0, 2, false);
            MSIREN_X1W_SLOT(0, 3, false);
            MSIREN_X1W_SLOT(1, 0, false);
            MSIREN_X1W_SLOT(1, 1, false);
            MSIREN_X1W_SLOT(1, 2, false);
            MSIREN_X1W_SLOT(1, 3, false);
        }
        {   // the final hidden layer (l == L - 1)
            MSIREN_X1W_SLOT(0, 0, true);
            MSIREN_X1W_SLOT(0, 1, true);
            MSIREN_X1W_SLOT(0, 2, true);
            MSIREN_X1W_SLOT(0, 3, true);
            MSIREN_X1W_SLOT(1, 0, true);
            MSIREN_X1W_SLOT(1, 1, true);
            MSIREN_X1W_SLOT(1, 2, true);
            MSIREN_X1W_SLOT(1, 3, true);
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
            if (lv && cu + c < P) p.out[(size_t)pu * P + cu + c] = sin_rev(s + p.bout);
        }
        cur_pass = __builtin_amdgcn_readfirstlane(qslot[pass & 1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the weights fetched for a pass that does not come)
    MSIREN_X1W_HOLD();
#undef MSIREN_X1W_HOLD
#undef MSIREN_X1W_SLOT
#undef MSIREN_X1W_KSTEP
#undef MSIREN_X1W_MFMA
#undef MSIREN_X1W_WAITK
#undef MSIREN_X1W_LOADK
#undef MSIREN_X1W_A
}

}  // namespace msiren
