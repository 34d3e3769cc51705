// robotstep.h -- robot.step(action, dt) of the Gym loop (social_nav_gym.py:240-243 -> robot_agent.py:119-136) for the kernels that move the
// robot by its action beside a crowd whose model is not a force model (ORCA, social momentum): the robot's motion does not depend on the
// crowd's model.  Holonomic ActionXY (vx, vy): p += a dt, v = a.  Unicycle ActionRot (v, r) (CS_ROBOT_UNICYCLE): p += v (cos, sin)(yaw + r) dt,
// yaw = (yaw + r) % 2 pi (python's modulo: in [0, 2 pi)), velocity = v (cos, sin)(yaw) -- the yaw turns by r at EVERY call, i.e. per substep.
// One lane per world takes this path: the library's sinf / cosf, no contraction of the sums (the holonomic sums are the restatement's, bit for bit).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "crowdstep.h"

namespace csimpl {

__device__ __forceinline__ void robot_action_step(int flags, float& x, float& y, float& yaw, float& vx, float& vy, float a0, float a1, float dt)
{
#pragma clang fp contract(off)
    if (flags & CS_ROBOT_UNICYCLE) {
        const float h = yaw + a1;
        x += cosf(h) * a0 * dt;
        y += sinf(h) * a0 * dt;
        // python's % (2 pi): h - floor(h / 2 pi) * 2 pi with a two-term 2 pi (stepcommon.h mod_two_pi: no float64 instruction in a step kernel)
        const float k = floorf(h * 0.15915494309189535f);
        float t = fmaf(-k, 6.2831855f, h);
        t = fmaf(k, 1.7484555e-7f, t);
        if (t < 0.0f) t += 6.2831855f;
        if (t >= 6.2831855f) t -= 6.2831855f;
        yaw = t;
        vx = cosf(t) * a0;
        vy = sinf(t) * a0;
    } else {
        x += a0 * dt;
        y += a1 * dt;
        vx = a0;
        vy = a1;
    }
}

} // namespace csimpl
