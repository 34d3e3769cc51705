// sfmstep_imit.hip -- builds of the fused SFM / HSFM step kernel (sfmstep_kernel.h, k_sfm_step<SOC, HEADED, PEQ, MAXT, OCC, ROWS_CT, LEAN>):
// LEAN = 4 -- a VISIBLE robot that follows a human motion model of its own (imitation learning): the substep loop of
// SocialNavGym.imitation_learning_step (/root/reference/social_gym/social_nav_gym.py:259-263), n x { update_robot ; update_humans }, as ONE
// launch: the robot is the last row of every world, its single-agent model (robot_model.h) runs at the head of every substep.
// 25 humans + robot (26 rows) and any other row count.  gfx950 only.
#include "sfmstep_kernel.h"

namespace cstep {

kfn sfm_builds_imit(const Variant& v, int type)
{
    CS_V(64, 1, 26, 4) CS_V(64, 3, 0, 4)
    return nullptr;
}

} // namespace cstep
