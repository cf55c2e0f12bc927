// Device code of the siren_trunk_x1n.hip.h instances libmsiren launches (declared extern in trunk_instances.h).
#include "siren_trunk_x1n.hip.h"
namespace msiren {
template __global__ void siren_trunk_x1n_kernel<0, 0, 0, 3>(TrunkX1Params);
template __global__ void siren_trunk_x1n_kernel<0, 0, 1, 3>(TrunkX1Params);
template __global__ void siren_trunk_x1n_kernel<0, 1, 0, 3>(TrunkX1Params);
template __global__ void siren_trunk_x1n_kernel<0, 1, 1, 3>(TrunkX1Params);
template __global__ void siren_trunk_x1n_kernel<1, 0, 0, 3>(TrunkX1Params);
template __global__ void siren_trunk_x1n_kernel<1, 0, 1, 3>(TrunkX1Params);
template __global__ void siren_trunk_x1n_kernel<1, 1, 0, 3>(TrunkX1Params);
template __global__ void siren_trunk_x1n_kernel<1, 1, 1, 3>(TrunkX1Params);
}  // namespace msiren
