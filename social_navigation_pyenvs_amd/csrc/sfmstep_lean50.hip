// sfmstep_lean50.hip -- builds of the fused SFM / HSFM step kernel (sfmstep_kernel.h, k_sfm_step<SOC, HEADED, PEQ, MAXT, OCC, ROWS_CT, LEAN>):
// 50 rows per world without / with walls (BASELINE.json configs[4]).
// One translation unit per group of builds so that they compile in parallel; crowdstep.hip picks the build (select_variant).
// Reference path: update_humans_parallel, /root/reference/social_gym/src/forces_parallel.py:185-284.  gfx950 only.
#include "sfmstep_kernel.h"

namespace cstep {

kfn sfm_builds_lean50(const Variant& v, int type)
{
    CS_V(64, 3, 50, 1) CS_V(64, 3, 50, 2)
    return nullptr;
}

} // namespace cstep
