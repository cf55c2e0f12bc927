// Types and helpers shared by the 16-bit-operand trunks (split-fp16 "f16x3" n / h / w kernels, single-product x1 kernel):
// launch parameters, the LDS layout of the register-resident kernels, the fp16 domain guard, fragment packing.
//
// Arithmetic of the split-fp16 trunks.  Every hidden-layer operand is split into two fp16 numbers, v = hi + lo
// (hi = f16(v), lo = f16(v - hi); 22 significant bits), and the product is evaluated as
//     W*x  ~=  W_lo*x_hi + W_hi*x_lo + W_hi*x_hi        (the lo*lo term, 2^-22 relative, is dropped)
// with three fp16 MFMAs accumulating in fp32.  Weights are split once on the host after scaling by a power of two;
// activations are split in the epilogue.  gfx950's MFMA keeps fp16 subnormal inputs (tools/f16_probe.hip).  Measured
// against the fp64 oracle this arithmetic is indistinguishable from true fp32 (tests/test_gpu_parity.py).
// (Round 1's kernel on 32x32x16 tiles, which these definitions were written for, is kept as a record under
// tools/experiments/ of commit 27d6e80, pruned in round 6; the library ships the 16x16x32 kernels.)
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
    const float* s0t;         // (64, P, 4): layer-0 activations act0(W0 x_p + b0) before modulation, feature-group major
    const _Float16* wp;       // [(L-1)*8 chunks][16 k-steps][2: hi,lo][64 lanes][8]
    const float* bias;        // (L-1, 256) in revolutions
    const float* wout;        // (256) * w0/2pi
    const float* mods;        // (L, B, 256)
    float* out;               // (B, P)
    float winv[16];           // per hidden layer: exact inverse of the power-of-two weight scale
    float bout, cg0, cg;
    int B, P, L, units_per_patch, total_units;
    int unit_base;            // first unit of this launch (16x16x32 kernels; a launch covers [unit_base, unit_base + total_units))
    const int* plan;          // optional (compact_flags_kernel): the unit count is plan[1] (<= total_units)
    int* pass_counter;        // work queue (never reset: the host passes the value it holds at launch)
    unsigned pass_base;        // value of *pass_counter when this launch starts (arithmetic is modulo 2^32)
    unsigned long long* stamps; // diagnostic instantiation only: [grid][8 passes][8] s_memtime + realtime
    int* status;              // domain guard: the stream's flag word (device memory; may be null) -- a launch that meets a scaled
    int status_val;           // modulation that does not fit fp16 writes its own number there (read by the conditional fp32 launch behind it)
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
    static __host__ __device__ constexpr int queue(int L) { return mods(L) + 4 * L * 1024; }  // 2 ints: next pass id
    static __host__ __device__ constexpr int winv(int L) { return queue(L) + 16; }  // per-layer inverse weight scales
    static __host__ __device__ constexpr int total(int L) { return winv(L) + 64; }
};

// Domain of the split-fp16 arithmetic on the activation side: x' = a * (m * 2^-a_next) is rounded to fp16 (hi) with |a| <= 1,
// so a scaled modulation beyond fp16's largest finite value (or a NaN / inf) would give inf / NaN silently.  Checked where the
// modulation rows are staged (a handful of compares per unit); the flag is a word in host memory the library reads at its
// next synchronisation (MSIREN_E_RANGE, or the exact-fp32 re-run of a host-pointer call).
__device__ __forceinline__ bool f16_out_of_range(const f32x4 m) {
    return !(__builtin_fabsf(m[0]) <= 65504.f) || !(__builtin_fabsf(m[1]) <= 65504.f) || !(__builtin_fabsf(m[2]) <= 65504.f) ||
           !(__builtin_fabsf(m[3]) <= 65504.f);
}

__device__ __forceinline__ h8 pack_h8(fp16x2 a, fp16x2 b, fp16x2 c, fp16x2 d) {
    u32x4 u;
    u[0] = __builtin_bit_cast(unsigned, a);
    u[1] = __builtin_bit_cast(unsigned, b);
    u[2] = __builtin_bit_cast(unsigned, c);
    u[3] = __builtin_bit_cast(unsigned, d);
    return __builtin_bit_cast(h8, u);
}

// Register-file placement.  The kernel keeps 256 registers of activations (this layer's and the next
// layer's B operands) live for a whole layer; they only fit if they sit in the ACCUMULATOR half of the
// unified 512-register file, which MFMA can read B from directly.  Left alone hipcc keeps builtin-MFMA
// operands in arch VGPRs (and then spills ~300 of them), so the placement is pinned here: every B
// fragment passes through an "a"-constrained asm once, when produced; the MFMAs themselves are the
// builtin, compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (A and the accumulator in arch VGPRs).
__device__ __forceinline__ h8 to_acc_file(h8 v) {
    h8 r;
    asm("; activation fragment -> AGPR" : "=a"(r) : "0"(v));
    return r;
}
// D = A*B (first k-step of a tile: C = 0) and D += A*B.  A (weights) and the accumulator in arch VGPRs,
// B (activations) in AGPRs: 2 x 128 activation registers fill the accumulator half.  Builtins: hipcc
// inserts whatever hazard padding the operands need.
__device__ __forceinline__ void mfma_f16_first(f32x16& d, const h8& a, const h8& b) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ void mfma_f16_acc(f32x16& d, const h8& a, const h8& b) {
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
}

// v (fp32) -> hi, lo (fp16, round toward zero; lo absorbs hi's truncation error exactly)
__device__ __forceinline__ void split4(const f32x4 v, fp16x2& h01, fp16x2& h23, fp16x2& l01, fp16x2& l23) {
    h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
    h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
    l01 = __builtin_amdgcn_cvt_pkrtz(v[0] - (float)h01[0], v[1] - (float)h01[1]);
    l23 = __builtin_amdgcn_cvt_pkrtz(v[2] - (float)h23[0], v[3] - (float)h23[1]);
}

}  // namespace msiren
