// placeholder, replaced below
#include <hip/hip_runtime.h>
#include "vrc_params.h"
namespace vrc { hipError_t launch_raycast_jump(const RaycastParams &, hipStream_t) { return hipErrorNotSupported; } }
