// hdsm_consts.h — host-side construction of the config-level constants (hdsm::Consts).
#pragma once
#include "../../include/hdsm.h"
#include "hdsm_types.h"

namespace hdsm {
// Counterpart of Agent::InitializePlannerParameters + Agent::CreateGurobiModel (AC:2169-2188, AC:2071-2153):
// everything that depends only on the ROS parameters is computed once here, in fp64, on the host.
// Returns HDSM_OK or HDSM_ERR_BAD_ARG (and a message through `err`, which may be null).
int build_consts(const hdsm_params* prm, Consts* out, const char** err);
}  // namespace hdsm
