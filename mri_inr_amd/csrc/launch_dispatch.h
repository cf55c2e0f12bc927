// Pieces of launch_dispatch.hip that diagnostics.hip needs as well (include the trunk kernel headers first).
#pragma once
#include "host_ctx.h"

namespace mh {
msiren::TrunkParams make_trunk_params(msiren_ctx* h, const float* mods, int stride, int64_t B, float* out_dev);
}
