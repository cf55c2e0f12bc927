// Device code of the encoder_modulator_f16x3.hip.h instances libmsiren launches (declared extern in trunk_instances.h).
#include "encoder_modulator_f16x3.hip.h"
namespace msiren {
template __global__ void latent_mods_f16x3_kernel<2, 2, 2, 3>(EmTailParams);
template __global__ void latent_mods_f16x3_kernel<2, 2, 4, 3>(EmTailParams);
template __global__ void latent_mods_f16x3_kernel<2, 2, 8, 3>(EmTailParams);
template __global__ void latent_mods_f16x3_kernel<2, 2, 4, 1>(EmTailParams);
template __global__ void latent_mods_f16x3_kernel<2, 2, 4, 2>(EmTailParams);
template __global__ void latent_mods_f16x3_kernel<4, 1, 4, 3>(EmTailParams);
template __global__ void latent_mods_f16x3_kernel<4, 1, 8, 3>(EmTailParams);
template __global__ void latent_mods_f16x3_kernel<4, 1, 4, 1>(EmTailParams);
template __global__ void latent_mods_f16x3_kernel<4, 1, 4, 2>(EmTailParams);
template __global__ void encoder_conv_f16x3_kernel<1>(EncoderParams, const float*, em_u4*, float*);
}  // namespace msiren
