// respawnx.h -- where the c-th human respawned in ONE substep lands (the parallel-traffic rule, /root/reference/social_gym/src/motion_model_manager.py:407-422).
// The reference walks the flagged humans of a world in index order, each behind everybody else:  x_0 = max(max_x + 2 max_r, bound) and, the
// (c-1)-th being the rightmost human by then,  x_c = max(x_{c-1} + 2 max_r, bound) = x_0 + c * 2 max_r  -- accumulated in float64 (Python floats;
// for an ORCA crowd the float32 of RVO2 only sees the result, :416 setAgentPosition).  Accumulated in float32 the c-th human lands c roundings of
// an ~88 m coordinate away from that (1e-4 m at c = 30: the worst figure of round 5's "4096-human traffic world" parity group).  Here the sum is
// formed in double -- exact for float32 operands: 24-bit terms, at most 2^13 of them -- and rounded to float32 ONCE.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace csimpl {

__device__ __forceinline__ float respawn_x(float max_x, float max_r, float bound_x, int c)
{
    const double step = 2.0 * (double)max_r;
    const double x0 = fmax((double)max_x + step, (double)bound_x);
    return (float)(x0 + (double)c * step);
}

} // namespace csimpl
