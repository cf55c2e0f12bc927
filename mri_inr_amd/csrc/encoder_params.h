// Weights of FixedEncoder as the encoder kernels take them (encoder_modulator.hip.h, encoder_modulator_f16x3.hip.h).
// Reference: src/networks/encoding/siren_encoder.py:503-512.
#pragma once
#include <hip/hip_runtime.h>

namespace msiren {

struct EncoderParams {
    const float* c1w;  // (16, 9)              conv 3x3 s2 p1, 1 -> 16
    const float* c1b;  // (16)
    const float* c2w;  // (144, 32) transposed conv 3x3 s2 p1, 16 -> 32; k = c*9 + ky*3 + kx
    const float* c2b;  // (32)
    const float* c3w;  // (2048, 64) transposed conv 8x8, 32 -> 64;      k = c*64 + y*8 + x
    const float* c3b;  // (64)
    const float* fcw;  // (64, Z) transposed   Linear(64, Z)
    const float* fcb;  // (Z)
    int Z;
    const void* c2f16;  // conv2 as MFMA A fragments, split fp16: [2 channel tiles][5 k-steps][hi|lo][64 lanes][8 x f16] (encoder_modulator_f16x3.hip.h)
    float c2_winv;      // exact inverse of conv2's power-of-two weight scale
    const int* plan;   // optional (compact_flags_kernel): workgroup j handles tile plan[2 + j], j < plan[0]
};

// tile t of the call
__device__ __forceinline__ const float* enc_tile(const EncoderParams&, const float* tiles, int t) { return tiles + (size_t)t * 1024; }

__device__ __forceinline__ float leaky02(float x) { return x >= 0.f ? x : 0.2f * x; }

}  // namespace msiren
