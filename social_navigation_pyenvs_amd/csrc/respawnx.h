// respawnx.h -- where the c-th human respawned in ONE substep lands (the parallel-traffic rule, /root/reference/social_gym/src/motion_model_manager.py:407-422).
// The reference walks the flagged humans of a world in index order, each behind everybody else:  x_0 = max(max_x + 2 max_r, bound) and, the
// (c-1)-th being the rightmost human by then,  x_c = max(x_{c-1} + 2 max_r, bound) = x_0 + c * 2 max_r  -- accumulated in float64 (Python floats;
// for an ORCA crowd the float32 of RVO2 only sees the result, :416 setAgentPosition).  Accumulated in float32 the c-th human lands c roundings of
// an ~88 m coordinate away from that (1e-4 m at c = 30: the worst figure of round 5's "4096-human traffic world" parity group).  Here the sum is
// formed exactly and rounded to float32 ONCE.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace csimpl {

#ifndef CS_RESPAWNX
#define CS_RESPAWNX 1
#endif

// FAST0: a uniform early way out for c == 0 (the usual case: one human of a world respawns in a substep -- a single rounding as it is).  The DPP row
// kernel instantiates the straight form: with the branch compiled in, its launch was 2 % slower whether anybody respawns or not (same-box A/B).
template <bool FAST0 = true>
__device__ __forceinline__ float respawn_x(float max_x, float max_r, float bound_x, int c)
{
#if CS_RESPAWNX == 0
    // in double: exact for float32 operands (24-bit terms, at most 2^13 of them)
    const double step = 2.0 * (double)max_r;
    const double x0 = fmax((double)max_x + step, (double)bound_x);
    return (float)(x0 + (double)c * step);
#else
    // the same sum in float32 pairs (hi + lo, error-free transformations: Knuth's TwoSum, an FMA for the product's error): the hot kernels carry
    // no float64 arithmetic (the DPP row kernel lost 0.7 us per launch of 10.4 with the double form compiled in beside its substep loop, in a branch
    // cfg2 never takes: same-box A/B, tools/ab_vs_prev.sh; HISTORY.md)
#pragma clang fp contract(off)
    const float step = 2.0f * max_r;                          // exact
    if constexpr (FAST0) { if (c == 0) return fmaxf(max_x + step, bound_x); }
    float s = max_x + step;                                   // TwoSum(max_x, step) -> s + e
    float bb = s - max_x;
    float e = (max_x - (s - bb)) + (step - bb);
    if (!(s > bound_x)) { s = bound_x; e = 0.0f; }            // x_0 = max(max_x + 2 max_r, bound)
    const float cf = (float)c;                                // exact (c < 2^24)
    const float p = cf * step;                                // TwoProduct(c, step) -> p + pe
    const float pe = __builtin_fmaf(cf, step, -p);
    const float t = s + p;                                    // TwoSum(s, p) -> t + te
    const float tb = t - s;
    const float te = (s - (t - tb)) + (p - tb);
    return t + ((e + pe) + te);                               // one rounding of the exact sum (the low words are far below ulp(t) / 2 apart from ties)
#endif
}

} // namespace csimpl
