// sfmstep_robotx.hip -- builds of the fused SFM / HSFM step kernel (sfmstep_kernel.h, k_sfm_step<SOC, HEADED, PEQ, MAXT, OCC, ROWS_CT, LEAN>):
// 5 / 10 / 50 humans + a visible robot (6, 11, 51 rows).
// One translation unit per group of builds so that they compile in parallel; crowdstep.hip picks the build (select_variant).
// Reference path: update_humans_parallel, /root/reference/social_gym/src/forces_parallel.py:185-284.  gfx950 only.
#include "sfmstep_kernel.h"

namespace cstep {

kfn sfm_builds_robotx(const Variant& v, int type)
{
    CS_V(64, 4, 6, 3) CS_V(64, 4, 11, 3) CS_V(64, 3, 51, 3)
    return nullptr;
}

} // namespace cstep
