// Device code of the siren_trunk_f32.hip.h instances libmsiren launches (declared extern in trunk_instances.h).
#include "siren_trunk_f32.hip.h"
namespace msiren {
template __global__ void siren_trunk_f32_kernel<128, 0, 0>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<128, 0, 1>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<128, 1, 0>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<128, 1, 1>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<256, 0, 0>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<256, 0, 1>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<256, 1, 0>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<256, 1, 1>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<384, 0, 0>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<384, 0, 1>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<384, 1, 0>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<384, 1, 1>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<512, 0, 0>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<512, 0, 1>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<512, 1, 0>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<512, 1, 1>(TrunkParams);
template __global__ void siren_trunk_f32_kernel<256, 0, 0, 1>(TrunkParams);
template __global__ void siren_trunk_f32_cond_kernel<0>(TrunkParams);
template __global__ void siren_trunk_f32_cond_kernel<1>(TrunkParams);
}  // namespace msiren
