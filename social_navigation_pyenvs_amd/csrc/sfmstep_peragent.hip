// sfmstep_peragent.hip -- builds of the fused SFM / HSFM step kernel (sfmstep_kernel.h, k_sfm_step<SOC, HEADED, PEQ, MAXT, OCC, ROWS_CT, LEAN>):
// 25 rows per world with PER-AGENT parameters (all_params_equal = False; forces_parallel.py:43-84, :261) for the Helbing / Guo laws: the
// pair-once loop with both directions of a pair evaluated by the lane that visits it, the partners' parameter rows held in registers.
// One translation unit per group of builds so that they compile in parallel; crowdstep.hip picks the build (select_variant).
// Reference path: update_humans_parallel, /root/reference/social_gym/src/forces_parallel.py:185-284.  gfx950 only.
#include "sfmstep_kernel.h"

namespace cstep {

kfn sfm_builds_peragent(const Variant& v, int type)
{
    if (!(v.maxt == 64 && v.occ == 1 && v.rows_ct == 25 && v.lean == 0 && !v.peq)) return nullptr;
    switch (type) {   // (Moussaid, types 2 / 5 / 8, keeps the all-partners loop of the generic build)
        case 0: return (kfn)k_sfm_step<0, 0, false, 64, 1, 25, 0>;
        case 1: return (kfn)k_sfm_step<1, 0, false, 64, 1, 25, 0>;
        case 3: return (kfn)k_sfm_step<0, 1, false, 64, 1, 25, 0>;
        case 4: return (kfn)k_sfm_step<1, 1, false, 64, 1, 25, 0>;
        case 6: return (kfn)k_sfm_step<0, 2, false, 64, 1, 25, 0>;
        case 7: return (kfn)k_sfm_step<1, 2, false, 64, 1, 25, 0>;
    }
    return nullptr;
}

} // namespace cstep
