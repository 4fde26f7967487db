// eb_env_step_t1.hip — the TASK_STRAIGHT instantiations of the env step's kernels (eb_env_step_body.h): a translation unit of their own,
// compiled next to eb_env_step.hip's (build time: one unit took 95 s).
#include "eb_env_step_body.h"

namespace eb {
template hipError_t launch_env_step_task<TASK_STRAIGHT>(const EnvStepArgs&, int, bool, int, size_t, int, hipStream_t);
}  // namespace eb
