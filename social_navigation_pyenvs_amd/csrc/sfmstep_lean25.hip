// sfmstep_lean25.hip -- builds of the fused SFM / HSFM step kernel (sfmstep_kernel.h, k_sfm_step<SOC, HEADED, PEQ, MAXT, OCC, ROWS_CT, LEAN>):
// 25 rows per world, plain crowd batch (BASELINE.json configs[2]: the headline).
// One translation unit per group of builds so that they compile in parallel; crowdstep.hip picks the build (select_variant).
// Reference path: update_humans_parallel, /root/reference/social_gym/src/forces_parallel.py:185-284.  gfx950 only.
#include "sfmstep_kernel.h"

namespace cstep {

kfn sfm_builds_lean25(const Variant& v, int type)
{
    CS_V(64, 1, 25, 1) CS_V(64, 4, 25, 1)
    return nullptr;
}

} // namespace cstep
