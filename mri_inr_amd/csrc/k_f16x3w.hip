// Device code of the siren_trunk_f16x3w.hip.h instances libmsiren launches (declared extern in trunk_instances.h).
#include "siren_trunk_f16x3w.hip.h"
namespace msiren {
template __global__ void siren_trunk_f16x3w_kernel<0, 4>(TrunkWsParams);
template __global__ void siren_trunk_f16x3w_kernel<1, 4>(TrunkWsParams);
template __global__ void siren_trunk_f16x3w_kernel<0, 4, 1>(TrunkWsParams);
}  // namespace msiren
