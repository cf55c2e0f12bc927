// Device code of the siren_trunk_f16x3h.hip.h instances libmsiren launches (declared extern in trunk_instances.h).
#include "siren_trunk_f16x3h.hip.h"
namespace msiren {
template __global__ void siren_trunk_f16x3h_kernel<0, 3, 5>(TrunkF16Params);
template __global__ void siren_trunk_f16x3h_kernel<0, 4, 5>(TrunkF16Params);
template __global__ void siren_trunk_f16x3h_kernel<1, 3, 5>(TrunkF16Params);
template __global__ void siren_trunk_f16x3h_kernel<1, 4, 5>(TrunkF16Params);
}  // namespace msiren
