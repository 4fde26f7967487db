// eb_env_step_t2.hip — the TASK_RIGHT instantiations of the env step's kernels (eb_env_step_body.h): a translation unit of their own,
// compiled next to eb_env_step.hip's (build time: one unit took 95 s).
#include "eb_env_step_body.h"

namespace eb {
template hipError_t launch_env_step_task<TASK_RIGHT>(const EnvStepArgs&, int, bool, int, size_t, int, hipStream_t);
}  // namespace eb
