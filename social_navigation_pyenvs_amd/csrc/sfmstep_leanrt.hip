// sfmstep_leanrt.hip -- builds of the fused SFM / HSFM step kernel (sfmstep_kernel.h, k_sfm_step<SOC, HEADED, PEQ, MAXT, OCC, ROWS_CT, LEAN>):
// the lean builds with a run-time row count (any row count without a compile-time build).
// One translation unit per group of builds so that they compile in parallel; crowdstep.hip picks the build (select_variant).
// Reference path: update_humans_parallel, /root/reference/social_gym/src/forces_parallel.py:185-284.  gfx950 only.
#include "sfmstep_kernel.h"

namespace cstep {

kfn sfm_builds_leanrt(const Variant& v, int type)
{
    CS_V(64, 3, 0, 1) CS_V(64, 3, 0, 2) CS_V(64, 3, 0, 3) CS_V(64, 3, 0, 5)
    return nullptr;
}

} // namespace cstep
