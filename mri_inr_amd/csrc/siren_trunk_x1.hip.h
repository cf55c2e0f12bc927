// Types and helpers of the single-product 16-bit trunk for wide/deep models (the kernel: siren_trunk_x1n.hip.h):
// H = 512, bf16 (or fp16) operands, fp32 accumulation, optional residual connections.
//
// This is BASELINE config 5 ("deep residual variant: 10 layers, hidden 512, latent 128, bf16 MFMA").
// The reference's residual model lives on a branch that is not in the container (README.md:27-29),
// so the residual semantics are this build's own and PARITY IS UNPINNED against the reference:
//     x_{l+1} = x_l + mod_l * act(W_l x_l + b_l)   for l >= 1      (layer 0 and last_layer unchanged)
// (oracle/siren_oracle.py: siren_forward(residual=True)); the tolerance is the 16-bit format's, not 1e-4.
//
// (Round 1's kernel on 32x32x16 tiles, which this header used to hold, was kept as a record under tools/experiments/ until round 6 (commit 27d6e80 has it); same-box
// A/B against its successor: profiles/r4/04_config5_x1n_vs_x1_ab.txt.)
#pragma once
#include <hip/hip_runtime.h>

#include "siren_trunk_f16_common.hip.h"

namespace msiren {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct TrunkX1Params {
    const float* s0t;         // (128, P, 4): layer-0 activations act0(W0 x_p + b0) before modulation, feature-group major
    const unsigned short* wp; // [(L-1)*16 chunks][16 k-steps][2 sub-tiles][64 lanes][8] bf16 / fp16 bit patterns
    const float* bias32;      // (L-1, 512) in revolutions x the layer's weight scale, fp32 (the MFMAs' C operand)
    const _Float16* wout;     // (512) * w0/2pi, fp16
    const float* mods;        // (L, B, 512)
    float* out;               // (B, P)
    float winv[64];           // per hidden layer: exact inverse of the power-of-two weight scale (1 for bf16)
    float bout, cg0, cg;
    int B, P, L, units_per_patch, total_units;
    const int* plan;          // optional (compact_flags_kernel): the unit count is plan[1] (<= total_units)
    int* pass_counter;        // work queue (never reset: the host passes the value it holds at launch)
    unsigned pass_base;
    // fp16 operands only (BF = 0): a launch that stores a non-finite output writes its own number to *status -- an activation,
    // a modulation or a residual sum beyond fp16's 65 504 became inf, and inf is NaN one sine later, where the reference's fp32
    // stays finite; the conditional exact-fp32 launch behind it (siren_trunk_f32_kernel with p.cond) then redoes the batch
    int* status;
    int status_val;
};

constexpr int X1_CHUNK_BYTES = 32768;

__device__ __forceinline__ u32x4 x1_to_acc_file(u32x4 v) {
    u32x4 r;
    asm("; activation fragment -> AGPR" : "=a"(r) : "0"(v));
    return r;
}

template <int BF>
__device__ __forceinline__ unsigned x1_pack2(float a, float b) {  // round to nearest even, packed
    const f32x2 v = {a, b};
    if constexpr (BF) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hf2));
}

template <int BF>
__device__ __forceinline__ void x1_unpack2(unsigned u, float& a, float& b) {
    if constexpr (BF) {
        a = __builtin_bit_cast(float, u << 16);
        b = __builtin_bit_cast(float, u & 0xffff0000u);
    } else {
        const hf2 h = __builtin_bit_cast(hf2, u);
        a = (float)h[0];
        b = (float)h[1];
    }
}

}  // namespace msiren
