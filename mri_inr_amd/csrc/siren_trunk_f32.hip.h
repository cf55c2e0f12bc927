// Fused modulated-SIREN trunk for gfx950 (MI355X), exact-fp32 path.
//
// One launch evaluates SirenNet.forward (reference: src/networks/modulated_siren.py:215-233) for
// every coordinate of every patch: layer 0 (K=2, VALU) -> L-1 hidden layers
// (v_mfma_f32_32x32x2_f32) -> last_layer dot product (N=1) -> sine; activations never leave the CU.
//
// Work decomposition
//   workgroup = 256 threads = 4 waves = one (patch b, chunk of 64 coordinates); grid = B * ceil(P/64).
//   LDS holds the activation tile X as [k/4][coord 0..63][k%4] fp32 (HP*256 B): 64 KB at H=256, so
//   two workgroups share a CU (one wave of each per SIMD): while one sits in its epilogue or at a
//   barrier the other's MFMAs keep the matrix pipe busy.
//   Each wave owns HP/4 output features (TT = HP/128 tiles of 32) for all 64 coordinates
//   (2 column tiles): acc[TT][2] x 16 registers.  Per k-step of 8:
//     A (weights)     : one global_load_dwordx4 per feature tile from the host-packed stream
//                       Wp[layer][wave][q][tt][lane][4]  (1 KiB, fully coalesced, L2-resident;
//                       every element is used by exactly one wave, so it never goes through LDS);
//     B (activations) : one ds_read_b128 per column tile, conflict-free;
//     4 steps x TT x 2 MFMAs (32x32x2): half-wave h supplies k = 8q + 4h + j at step j.
//   The D layout (row = (r&3) + 8(r>>2) + 4h, col = lane&31) puts four consecutive features
//   of one coordinate in registers 4g..4g+3, i.e. exactly one float4 of the X image:
//   the epilogue is  bias -> act -> *mod -> ds_write_b128  with no shuffles.
//
// Scaling folded on the host (weights_pack.hip: pack_trunk): all weights/biases are pre-multiplied by
// w0/(2*pi) (w0_initial for layer 0) so that the accumulator is the sine argument in REVOLUTIONS,
// the unit v_sin_f32 takes.  Morlet's Gaussian exp(-p^2/2) becomes exp2(cg * r^2).
#pragma once
#include <hip/hip_runtime.h>

namespace msiren {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TrunkParams {
    const float* grid;   // (P,2) coordinates (the state_dict's "grid" buffer)
    const float* l0;     // (HP,4): {w_row*c0, w_col*c0, b*c0, 0}
    const float* wp;     // packed hidden weights, see above
    const float* bias;   // (L-1, HP), scaled
    const float* wout;   // (HP), scaled
    const float* mods;   // (L, B, mod_stride)
    float* out;          // (B, P)
    float bout;          // scaled
    float cg0, cg;       // Morlet Gaussian constants for layer 0 / hidden layers
    int B, P, L, mod_stride, chunks;
    unsigned long long* stamps;  // diagnostic build only: [grid][32] s_memtime stamps
    const int* plan;     // optional (compact_flags_kernel): only patches b < plan[0] are evaluated
    // conditional instance only (siren_trunk_f32_cond_kernel): run iff *cond == cond_val, over `items` (patch, chunk)
    // work items; when it runs it raises *host_flag (a word in host memory, informational)
    const int* cond;
    int cond_val, items;
    int* host_flag;
};

// sin(2*pi*r) with the hardware sine, whose argument is in revolutions.  On gfx950 v_sin_f32 performs
// the fractional range reduction itself and exactly: measured max abs error 1.25e-7 over [-32, 32]
// (tools/sin_accuracy.hip) and exact results out to 2^20 revolutions (tools/sin_range_probe.hip), so no
// explicit v_fract / v_rndne is spent on it.
__device__ __forceinline__ float sin_rev(float r) { return __builtin_amdgcn_sinf(r); }

template <int ACT>
__device__ __forceinline__ float activate(float r, float cg) {
    if constexpr (ACT == 1) {
        return sin_rev(r) * __builtin_amdgcn_exp2f(cg * r * r);
    } else {
        return sin_rev(r);
    }
}

// last_layer's partial sum over four features, part + ((a0 m0 + a1 m1) + (a2 m2 + a3 m3)), with the multiply-adds SPELLED OUT:
// left to the compiler, a * b + c is fused or not depending on the shape of the surrounding code, and two kernels that
// must give the same bits (siren_trunk_f32_kernel and its conditional stand-in behind the f16x3 trunks) would differ.
__device__ __forceinline__ float dot4_acc(float part, const f32x4 a, const f32x4 m) {
    const float s01 = __builtin_fmaf(a[1], m[1], a[0] * m[0]);
    const float s23 = __builtin_fmaf(a[3], m[3], a[2] * m[2]);
    return part + (s01 + s23);
}

// DBG = 1 is a separate diagnostic instantiation (msiren_trunk_timeline): wave 0 of every workgroup
// stamps s_memtime at each phase boundary into p.stamps; the shipped kernel (DBG = 0) has no stamps.
// One work item = (patch, chunk of 64 coordinates) = `item`; the kernels below are the callers.
template <int HP, int ACT, int RES, int DBG>
__device__ __forceinline__ void siren_trunk_f32_item(const TrunkParams& p, const int item) {
    constexpr int TT = HP / 128;  // 32-feature tiles per wave
    constexpr int QN = HP / 8;    // k-blocks of 8 per layer (16*TT/2 MFMAs each)
    constexpr int KG = HP / 4;    // k-groups of 4 (rows of the X image)
    static_assert(HP % 128 == 0, "hidden width is padded to a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* X = reinterpret_cast<f32x4*>(lds);    // X[kg*64 + coord]              HP*256 B
    f32x4* P0 = X + KG * 64;                     // layer-0 rows {wx, wy, b, mod} HP*16 B

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int c32 = lane & 31;
    const int b = item / p.chunks;
    const int ch = item - b * p.chunks;
    if (p.plan && b >= p.plan[0]) return;  // workgroup-uniform, before any barrier
    const int L = p.L;
    int nstamp = 0;
    auto stamp = [&]() {
        if constexpr (DBG) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (tid == 0 && nstamp < 28) p.stamps[(size_t)item * 32 + 4 + nstamp] = t;
            ++nstamp;
        }
    };
    if constexpr (DBG) {
        if (tid == 0) {
            unsigned hwid, ldsa, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(ldsa));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            p.stamps[(size_t)item * 32 + 0] = hwid;
            p.stamps[(size_t)item * 32 + 1] = ldsa;
            p.stamps[(size_t)item * 32 + 2] = xcc;
            p.stamps[(size_t)item * 32 + 3] = __builtin_amdgcn_s_memrealtime();
        }
    }
    stamp();  // 0: start

    // ---------------- weight stream: a ring of 4 register stages, 3 k-blocks ahead --------------
    // Block s of the stream is k-block (s % QN) of hidden layer 1 + s / QN.  The ring runs across
    // layer boundaries: while a layer's epilogue executes, the next layer's first three blocks are
    // already in flight.  A lone wave on a SIMD (its partner workgroup in an epilogue or at a
    // barrier) issues 16*TT/2 MFMAs per block back to back, so the distance has to cover the L2
    // latency at the FULL matrix rate -- one block ahead (~1000 cycles) does not.
    const int nblk = (L - 1) * QN;
    const f32x4* wbase = reinterpret_cast<const f32x4*>(p.wp) + lane;
    auto loadA = [&](f32x4(&a)[TT], int s) {
        s = s < nblk ? s : nblk - 1;  // past the end: harmless re-load of the last block
        const int l1 = s / QN, q = s - l1 * QN;
        const f32x4* ptr = wbase + (((size_t)l1 * 4 + wave) * QN + q) * (TT * 64);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) a[tt] = ptr[tt * 64];
    };
    f32x4 a0[TT], a1[TT], a2[TT], a3[TT];
    if (L > 1) {
        loadA(a0, 0);
        loadA(a1, 1);
        loadA(a2, 2);
    }

    // ---------------- layer 0: K = 2, straight into the X image --------------------------------
    {
        // per-feature rows {w_row, w_col, bias, mod0} staged once, then broadcast-read from LDS
        const float* mod0 = p.mods + (size_t)b * p.mod_stride;
        const f32x4* l0 = reinterpret_cast<const f32x4*>(p.l0);
        for (int f = tid; f < HP; f += 256) {
            f32x4 w = l0[f];
            w[3] = mod0[f];
            P0[f] = w;
        }
        int pc = ch * 64 + lane;
        pc = pc < p.P ? pc : p.P - 1;
        const float2 xy = reinterpret_cast<const float2*>(p.grid)[pc];
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < KG / 4; ++i) {
            const int kg = wave * (KG / 4) + i;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 w = P0[4 * kg + e];
                const float r = __builtin_fmaf(xy.y, w.y, __builtin_fmaf(xy.x, w.x, w.z));
                v[e] = activate<ACT>(r, p.cg0) * w[3];
            }
            X[kg * 64 + lane] = v;
        }
    }
    __syncthreads();
    stamp();  // 1: layer 0 done

    // ---------------- hidden layers 1..L-1 on the matrix cores ---------------------------------
    const int fwave = wave * (32 * TT);  // first feature owned by this wave
    float part[2] = {0.f, 0.f};          // last_layer partial sums (used on the final hidden layer)

    for (int l = 1; l < L; ++l) {
        const float* bl = p.bias + (size_t)(l - 1) * HP;
        const float* ml = p.mods + ((size_t)l * p.B + b) * p.mod_stride;
        const bool last = (l == L - 1);

        // per-feature constants of this wave's rows: issued now, consumed in the epilogue.
        // On the final hidden layer the modulation is pre-multiplied by last_layer's weight:
        // out = sum_f act_f * (mod_f * wout_f)  (the residual variant needs them separately).
        f32x4 bias_r[TT][4], mod_r[TT][4];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int fo = fwave + 32 * tt + 8 * g + 4 * half;
                bias_r[tt][g] = *reinterpret_cast<const f32x4*>(bl + fo);
                mod_r[tt][g] = *reinterpret_cast<const f32x4*>(ml + fo);
            }
        if (!RES && last) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    mod_r[tt][g] *= *reinterpret_cast<const f32x4*>(p.wout + fwave + 32 * tt + 8 * g + 4 * half);
        }

        f32x16 acc[TT][2];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int jc = 0; jc < 2; ++jc)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tt][jc][r] = 0.f;

        const f32x4* xB = X + half * 64 + c32;  // + (2q)*64 + 32*jc
        auto loadB = [&](f32x4(&bb)[2], int q) {
            q = q < QN ? q : QN - 1;
            bb[0] = xB[(2 * q) * 64];
            bb[1] = xB[(2 * q) * 64 + 32];
        };
        auto mma = [&](const f32x4(&a)[TT], const f32x4(&bb)[2]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int jc = 0; jc < 2; ++jc)
                        acc[tt][jc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt][j], bb[jc][j], acc[tt][jc], 0, 0, 0);
        };

        f32x4 b0[2], b1[2];
        loadB(b0, 0);
        const int sb = (l - 1) * QN;
        // sched_barrier pins "issue the loads of later blocks, then this block's MFMAs": left alone,
        // hipcc sinks every load down to its first use and exposes the L2 latency per block.
#pragma nounroll
        for (int q = 0; q < QN; q += 4) {
            loadA(a3, sb + q + 3);
            loadB(b1, q + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            loadA(a0, sb + q + 4);
            loadB(b0, q + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            loadA(a1, sb + q + 5);
            loadB(b1, q + 3);
            __builtin_amdgcn_sched_barrier(0);
            mma(a2, b0);
            __builtin_amdgcn_sched_barrier(0);
            loadA(a2, sb + q + 6);
            loadB(b0, q + 4);
            __builtin_amdgcn_sched_barrier(0);
            mma(a3, b1);
            __builtin_amdgcn_sched_barrier(0);
        }

        stamp();  // K loop done (this wave)
        __syncthreads();  // every wave has finished reading X: rows may now be overwritten
        stamp();  // barrier passed

        const int kgw = wave * (8 * TT);
        if (!last) {
            // epilogue: bias -> activation -> modulation, back into this wave's rows of the X image
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int jc = 0; jc < 2; ++jc) {
                        const int xi = (kgw + 8 * tt + 2 * g + half) * 64 + 32 * jc + c32;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float r = acc[tt][jc][4 * g + e] + bias_r[tt][g][e];
                            v[e] = activate<ACT>(r, p.cg) * mod_r[tt][g][e];
                        }
                        if constexpr (RES) v += X[xi];
                        X[xi] = v;
                    }
            __syncthreads();
            stamp();  // epilogue + barrier done
        } else {
            // final hidden layer: its output feeds last_layer's dot product straight from registers
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 wo;
                    if constexpr (RES) wo = *reinterpret_cast<const f32x4*>(p.wout + fwave + 32 * tt + 8 * g + 4 * half);
#pragma unroll
                    for (int jc = 0; jc < 2; ++jc) {
                        if constexpr (RES) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float r = acc[tt][jc][4 * g + e] + bias_r[tt][g][e];
                                v[e] = activate<ACT>(r, p.cg) * mod_r[tt][g][e];
                            }
                            v += X[(kgw + 8 * tt + 2 * g + half) * 64 + 32 * jc + c32];
                            v *= wo;
                            part[jc] += (v[0] + v[1]) + (v[2] + v[3]);
                        } else {
                            f32x4 av;
#pragma unroll
                            for (int e = 0; e < 4; ++e) av[e] = activate<ACT>(acc[tt][jc][4 * g + e] + bias_r[tt][g][e], p.cg);
                            part[jc] = dot4_acc(part[jc], av, mod_r[tt][g]);  // (mod_r holds modulation x last_layer.weight here)
                        }
                    }
                }
        }
    }

    // ---------------- last_layer: dot over H features, always sine -----------------------------
    float* red = lds;  // [4][64], X is dead (or, for L == 1, read below before the barrier)
    if (L == 1) {
        // no hidden layer ran: take the dot product from the X image (coord = lane, kgs split by wave)
        float s = 0.f;
        for (int i = 0; i < KG / 4; ++i) {
            const int kg = wave * (KG / 4) + i;
            const f32x4 v = X[kg * 64 + lane];
            const f32x4 wo = *reinterpret_cast<const f32x4*>(p.wout + 4 * kg);
            s += v[0] * wo[0] + v[1] * wo[1] + v[2] * wo[2] + v[3] * wo[3];
        }
        __syncthreads();
        red[wave * 64 + lane] = s;
    } else {
        part[0] += __shfl_xor(part[0], 32);
        part[1] += __shfl_xor(part[1], 32);
        // (the barrier after the K loop already separates the last X reads from these writes,
        //  except for RES, whose epilogue reads X: add one)
        if constexpr (RES) __syncthreads();
        if (half == 0) {
            red[wave * 64 + c32] = part[0];
            red[wave * 64 + 32 + c32] = part[1];
        }
    }
    __syncthreads();
    if (tid < 64) {
        const float s = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid] + p.bout;
        const int pc = ch * 64 + tid;
        if (pc < p.P) p.out[(size_t)b * p.P + pc] = sin_rev(s);
    }
    stamp();  // end
    if constexpr (DBG) {
        if (tid == 0) p.stamps[(size_t)item * 32 + 31] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int HP, int ACT, int RES, int DBG = 0>
__global__ __launch_bounds__(256, (HP <= 256 ? 2 : 1)) void siren_trunk_f32_kernel(TrunkParams p) {
    // (as a conditional launch behind a single-product fp16 trunk, H = 512: leave unless that launch wrote its number to *cond)
    if (p.cond) {
        if (__builtin_amdgcn_readfirstlane(*p.cond) != p.cond_val) return;
        if (blockIdx.x == 0 && threadIdx.x == 0 && p.host_flag) *p.host_flag = 1;
    }
    siren_trunk_f32_item<HP, ACT, RES, DBG>(p, (int)blockIdx.x);
}

// The exact-fp32 trunk (H = 256) as a CONDITIONAL launch behind a split-fp16 trunk launch on the same stream: the f16x3
// kernels write their launch's number to *cond when a scaled modulation does not fit their fp16 operands
// (f16_out_of_range); every workgroup of this kernel reads the word first and leaves if it does not hold that number -- the
// normal case, a couple of microseconds for the launch.  Otherwise its (few, persistent) workgroups evaluate all `items`
// = (patch, chunk of 32 coordinates) into the same output buffer: the buffer always ends up holding what the reference's
// fp32 arithmetic computes (modulated_siren.py:215-233), whatever the modulations.
//
// Small on purpose -- 32 KB of LDS, <= 96 registers -- so that it is dispatched BESIDE a persistent register-resident trunk
// of the handle's other stream (which leaves 34 KB / 112 registers per CU; the 64-coordinate kernel above needs 68 KB and
// would sit in front of its stream until that trunk has left every CU: measured -12 % on the two-stream pipeline).  Speed
// when it does run is secondary; what matters is that it produces THE BITS OF siren_trunk_f32_kernel<256, ACT, 0>: the same
// v_mfma_f32_32x32x2_f32 on the same k pairs in the same order per accumulator, the same layer-0 FMAs, the same epilogue
// expressions, the same order of the last_layer sum (tests/test_gpu_ws.py: np.array_equal against the fp32 model).
template <int ACT>
__global__ __launch_bounds__(256, 5) void siren_trunk_f32_cond_kernel(TrunkParams p) {
    if (__builtin_amdgcn_readfirstlane(*p.cond) != p.cond_val) return;
    constexpr int HP = 256, TT = 2, QN = HP / 8, KG = HP / 4;
    __shared__ f32x4 X[KG * 32];  // X[kg * 32 + coord]: features 4 kg .. 4 kg + 3 of the chunk's 32 coordinates
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int c32 = lane & 31;
    const int L = p.L;
    if (blockIdx.x == 0 && tid == 0 && p.host_flag) *p.host_flag = 1;
    const int cpp = (p.P + 31) / 32;  // chunks per patch
    const int fwave = wave * 64, kgw = wave * 16;
    const f32x4* wbase = reinterpret_cast<const f32x4*>(p.wp) + lane;
    for (int item = (int)blockIdx.x; item < p.items; item += (int)gridDim.x) {
        const int b = item / cpp, ch = item - b * cpp;
        if (p.plan && b >= p.plan[0]) break;  // workgroup-uniform; items are in patch order
        int pc = ch * 32 + c32;
        pc = pc < p.P ? pc : p.P - 1;
        const float2 xy = reinterpret_cast<const float2*>(p.grid)[pc];
        {   // layer 0 (K = 2): this wave's 16 rows of the image, two rows (one per half-wave) at a time
            const f32x4* l0 = reinterpret_cast<const f32x4*>(p.l0);
            const float* mod0 = p.mods + (size_t)b * p.mod_stride;
            for (int i = 0; i < 8; ++i) {
                const int kg = kgw + 2 * i + half;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 w = l0[4 * kg + e];
                    const float r = __builtin_fmaf(xy.y, w.y, __builtin_fmaf(xy.x, w.x, w.z));
                    v[e] = activate<ACT>(r, p.cg0) * mod0[4 * kg + e];
                }
                X[kg * 32 + c32] = v;
            }
        }
        __syncthreads();
        float part = 0.f;
        for (int l = 1; l < L; ++l) {
            const float* bl = p.bias + (size_t)(l - 1) * HP;
            const float* ml = p.mods + ((size_t)l * p.B + b) * p.mod_stride;
            const bool last = (l == L - 1);
            f32x16 acc[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
            for (int q = 0; q < QN; ++q) {
                const f32x4* ptr = wbase + (((size_t)(l - 1) * 4 + wave) * QN + q) * (TT * 64);
                const f32x4 a0 = ptr[0], a1 = ptr[64];
                const f32x4 bb = X[(2 * q + half) * 32 + c32];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[j], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bb[j], acc[1], 0, 0, 0);
                }
            }
            __syncthreads();  // every wave has finished reading X: rows may now be overwritten
            // (two separate branches with the expressions of siren_trunk_f32_item, so that the compiler contracts -- or does not
            //  contract -- the same multiply-adds in both kernels: the results must be the same BITS)
            if (!last) {
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int fo = fwave + 32 * tt + 8 * g + 4 * half;
                        const f32x4 bias_r = *reinterpret_cast<const f32x4*>(bl + fo);
                        const f32x4 mod_r = *reinterpret_cast<const f32x4*>(ml + fo);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float r = acc[tt][4 * g + e] + bias_r[e];
                            v[e] = activate<ACT>(r, p.cg) * mod_r[e];
                        }
                        X[(kgw + 8 * tt + 2 * g + half) * 32 + c32] = v;
                    }
            } else {
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int fo = fwave + 32 * tt + 8 * g + 4 * half;
                        const f32x4 bias_r = *reinterpret_cast<const f32x4*>(bl + fo);
                        f32x4 mod_r = *reinterpret_cast<const f32x4*>(ml + fo);
                        mod_r *= *reinterpret_cast<const f32x4*>(p.wout + fo);
                        f32x4 av;
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[e] = activate<ACT>(acc[tt][4 * g + e] + bias_r[e], p.cg);
                        part = dot4_acc(part, av, mod_r);
                    }
            }
            if (!last) __syncthreads();
        }
        // last_layer: dot over the 256 features, always sine
        float* red = reinterpret_cast<float*>(X);  // [4][32]; X is dead (the barrier behind the last K loop)
        if (L == 1) {
            float s = 0.f;
            for (int i = 0; i < 8; ++i) {
                const int kg = kgw + 2 * i + half;
                const f32x4 v = X[kg * 32 + c32];
                const f32x4 wo = *reinterpret_cast<const f32x4*>(p.wout + 4 * kg);
                s += v[0] * wo[0] + v[1] * wo[1] + v[2] * wo[2] + v[3] * wo[3];
            }
            part = s;
            __syncthreads();
        }
        part += __shfl_xor(part, 32);
        if (half == 0) red[wave * 32 + c32] = part;
        __syncthreads();
        if (tid < 32) {
            const float s = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid] + p.bout;
            const int po = ch * 32 + tid;
            if (po < p.P) p.out[(size_t)b * p.P + po] = sin_rev(s);
        }
        __syncthreads();  // the next item reuses the image
    }
}

}  // namespace msiren
