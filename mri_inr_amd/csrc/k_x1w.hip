// Device code of the siren_trunk_x1w.hip.h instances libmsiren launches (declared extern in trunk_instances.h).
#include "siren_trunk_x1w.hip.h"
namespace msiren {
template __global__ void siren_trunk_x1w_kernel<0, 0, 0>(TrunkX1Params);
template __global__ void siren_trunk_x1w_kernel<0, 0, 1>(TrunkX1Params);
template __global__ void siren_trunk_x1w_kernel<0, 1, 0>(TrunkX1Params);
template __global__ void siren_trunk_x1w_kernel<0, 1, 1>(TrunkX1Params);
template __global__ void siren_trunk_x1w_kernel<1, 0, 0>(TrunkX1Params);
template __global__ void siren_trunk_x1w_kernel<1, 0, 1>(TrunkX1Params);
template __global__ void siren_trunk_x1w_kernel<1, 1, 0>(TrunkX1Params);
template __global__ void siren_trunk_x1w_kernel<1, 1, 1>(TrunkX1Params);
}  // namespace msiren
