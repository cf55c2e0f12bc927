// Kernel instantiations of the trunk translation units (k_*.hip), declared `extern` for the host units: a host file parses the
// kernel headers for their parameter structs, LDS layouts and schedules, but the device code of every trunk instance is
// generated once, in its own translation unit (make -j: six compilers side by side instead of one 4-minute run).
// GENERATED together with k_*.hip by the list in this file's history; keep the two in step (the link fails otherwise).
#pragma once
#include "encoder_modulator_f16x3.hip.h"
#include "siren_trunk_f16x3h.hip.h"
#include "siren_trunk_f16x3n.hip.h"
#include "siren_trunk_f16x3w.hip.h"
#include "siren_trunk_f32.hip.h"
#include "siren_trunk_x1n.hip.h"
#include "siren_trunk_x1w.hip.h"
namespace msiren {
extern template __global__ void siren_trunk_f32_kernel<128, 0, 0>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<128, 0, 1>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<128, 1, 0>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<128, 1, 1>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<256, 0, 0>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<256, 0, 1>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<256, 1, 0>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<256, 1, 1>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<384, 0, 0>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<384, 0, 1>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<384, 1, 0>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<384, 1, 1>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<512, 0, 0>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<512, 0, 1>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<512, 1, 0>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<512, 1, 1>(TrunkParams);
extern template __global__ void siren_trunk_f32_kernel<256, 0, 0, 1>(TrunkParams);
extern template __global__ void siren_trunk_f32_cond_kernel<0>(TrunkParams);
extern template __global__ void siren_trunk_f32_cond_kernel<1>(TrunkParams);
extern template __global__ void siren_trunk_f16x3n_kernel<0, 3, 5>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3n_kernel<0, 3, 0>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3n_kernel<0, 4, 5>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3n_kernel<0, 4, 0>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3n_kernel<1, 3, 5>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3n_kernel<1, 3, 0>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3n_kernel<1, 4, 5>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3n_kernel<1, 4, 0>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3n_kernel<0, 4, 5, 1>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3h_kernel<0, 3, 5>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3h_kernel<0, 4, 5>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3h_kernel<1, 3, 5>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3h_kernel<1, 4, 5>(TrunkF16Params);
extern template __global__ void siren_trunk_f16x3w_kernel<0, 4>(TrunkWsParams);
extern template __global__ void siren_trunk_f16x3w_kernel<1, 4>(TrunkWsParams);
extern template __global__ void siren_trunk_f16x3w_kernel<0, 4, 1>(TrunkWsParams);
extern template __global__ void siren_trunk_x1w_kernel<0, 0, 0>(TrunkX1Params);
extern template __global__ void siren_trunk_x1n_kernel<0, 0, 0, 3>(TrunkX1Params);
extern template __global__ void siren_trunk_x1w_kernel<0, 0, 1>(TrunkX1Params);
extern template __global__ void siren_trunk_x1n_kernel<0, 0, 1, 3>(TrunkX1Params);
extern template __global__ void siren_trunk_x1w_kernel<0, 1, 0>(TrunkX1Params);
extern template __global__ void siren_trunk_x1n_kernel<0, 1, 0, 3>(TrunkX1Params);
extern template __global__ void siren_trunk_x1w_kernel<0, 1, 1>(TrunkX1Params);
extern template __global__ void siren_trunk_x1n_kernel<0, 1, 1, 3>(TrunkX1Params);
extern template __global__ void siren_trunk_x1w_kernel<1, 0, 0>(TrunkX1Params);
extern template __global__ void siren_trunk_x1n_kernel<1, 0, 0, 3>(TrunkX1Params);
extern template __global__ void siren_trunk_x1w_kernel<1, 0, 1>(TrunkX1Params);
extern template __global__ void siren_trunk_x1n_kernel<1, 0, 1, 3>(TrunkX1Params);
extern template __global__ void siren_trunk_x1w_kernel<1, 1, 0>(TrunkX1Params);
extern template __global__ void siren_trunk_x1n_kernel<1, 1, 0, 3>(TrunkX1Params);
extern template __global__ void siren_trunk_x1w_kernel<1, 1, 1>(TrunkX1Params);
extern template __global__ void siren_trunk_x1n_kernel<1, 1, 1, 3>(TrunkX1Params);
extern template __global__ void latent_mods_f16x3_kernel<2, 2, 2, 3>(EmTailParams);
extern template __global__ void latent_mods_f16x3_kernel<2, 2, 4, 3>(EmTailParams);
extern template __global__ void latent_mods_f16x3_kernel<2, 2, 8, 3>(EmTailParams);
extern template __global__ void latent_mods_f16x3_kernel<2, 2, 4, 1>(EmTailParams);
extern template __global__ void latent_mods_f16x3_kernel<2, 2, 4, 2>(EmTailParams);
extern template __global__ void latent_mods_f16x3_kernel<4, 1, 4, 3>(EmTailParams);
extern template __global__ void latent_mods_f16x3_kernel<4, 1, 8, 3>(EmTailParams);
extern template __global__ void latent_mods_f16x3_kernel<4, 1, 4, 1>(EmTailParams);
extern template __global__ void latent_mods_f16x3_kernel<4, 1, 4, 2>(EmTailParams);
extern template __global__ void encoder_conv_f16x3_kernel<1>(EncoderParams, const float*, em_u4*, float*);
}  // namespace msiren
