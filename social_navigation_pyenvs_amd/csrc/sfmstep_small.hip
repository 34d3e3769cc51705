// sfmstep_small.hip -- builds of the fused SFM / HSFM step kernel (sfmstep_kernel.h, k_sfm_step<SOC, HEADED, PEQ, MAXT, OCC, ROWS_CT, LEAN>):
// 10 and 20 rows per world, plain crowd batch (10: BASELINE.json configs[1] when CROWDSTEP_ROW16=0).
// One translation unit per group of builds so that they compile in parallel; crowdstep.hip picks the build (select_variant).
// Reference path: update_humans_parallel, /root/reference/social_gym/src/forces_parallel.py:185-284.  gfx950 only.
#include "sfmstep_kernel.h"

namespace cstep {

kfn sfm_builds_small(const Variant& v, int type)
{
    CS_V(64, 4, 10, 1) CS_V(64, 1, 20, 1) CS_V(64, 4, 20, 1)
    return nullptr;
}

} // namespace cstep
