// sfmstep_generic.hip -- builds of the fused SFM / HSFM step kernel (sfmstep_kernel.h, k_sfm_step<SOC, HEADED, PEQ, MAXT, OCC, ROWS_CT, LEAN>):
// one world per block (rows > 64) and the run-time-loop build for everything the lean builds do not cover (walls + robot row, per-agent parameters, goal lists in memory, peek / out-of-place modes).
// One translation unit per group of builds so that they compile in parallel; crowdstep.hip picks the build (select_variant).
// Reference path: update_humans_parallel, /root/reference/social_gym/src/forces_parallel.py:185-284.  gfx950 only.
#include "sfmstep_kernel.h"

namespace cstep {

kfn sfm_builds_generic(const Variant& v, int type)
{
    CS_V(1024, 1, 0, 0) CS_V(64, 3, 0, 0)
    return nullptr;
}

} // namespace cstep
