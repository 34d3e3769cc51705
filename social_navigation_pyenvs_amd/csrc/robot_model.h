// robot_model.h -- the robot under a HUMAN motion model (imitation learning), the arithmetic shared by its two homes:
//   robot_model.hip   k_robot_model_step: one wavefront per world (cs_robot_model_step, the robot half of an alternating loop, and
//                     the robot's substeps behind the crowd's snapshots of an invisible-robot imitation block)
//   sfmstep_kernel.h  k_sfm_step<..., LEAN = 4>: the robot as the last row of the crowd's own fused launch (a VISIBLE robot:
//                     robot and crowd act on each other in every substep, so the robot's update sits inside the substep loop)
// Restates MotionModelManager.update_robot(t, dt) for the nine SFM / HSFM titles with the SINGLE-AGENT force functions
//   /root/reference/social_gym/src/motion_model_manager.py:591-629, 72-86;  forces.py:9-16, 27-53, 153-218, 279-290.
// Both homes must give the SAME bits (tests/test_imitation.py compares the fused launch with the alternating launches), so every
// function here runs with floating-point contraction OFF: each operation is one IEEE operation (or one libm call) whatever code
// surrounds it, and the per-human terms are summed in index order by both.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace rmodel {

constexpr float PI_F = 3.14159265358979323846f;
constexpr float TWO_PI_F = 6.28318530717958647692f;

// utils.py:7-13
__device__ __forceinline__ float bound_angle(float a)
{
    if (a >= TWO_PI_F) a = fmodf(a, TWO_PI_F);
    if (a <= -TWO_PI_F) a = fmodf(a, TWO_PI_F);
    if (a > PI_F) a -= TWO_PI_F;
    if (a < -PI_F) a += TWO_PI_F;
    return a;
}

// the robot's dynamic state and what its model remembers between substeps (agent.desired_force)
struct RState {
    float px, py, yaw, vx, vy, bvx, bvy, om;
    float radius, mass, gx, gy, vd;
    float fdx, fdy;
};

// The term human (hx, hy, hvx, hvy) adds to the robot's social force (forces.py:153-218, consider_robot = False).  P: the robot's
// 20 parameters (agent.py:269 slots); rij = robot radius + robot margin + human radius + human margin; soc = type % 3.
__device__ __forceinline__ void pair_term(int soc, const float* P, float px, float py, float vx, float vy, float hx, float hy, float hvx,
                                          float hvy, float rij, float& tx_out, float& ty_out)
{
#pragma clang fp contract(off)
    // (single-instruction rsq / exp as in the crowd kernel: <= 1 ulp each, two orders of magnitude inside the parity bar)
    const float dx = px - hx, dy = py - hy;
    const float d2h = fmaxf(dx * dx + dy * dy, 1e-30f);    // (coincident robot and human: finite, as in the crowd kernel)
    const float dinv = __builtin_amdgcn_rsqf(d2h);
    const float dn = d2h * dinv;
    const float nx = dx * dinv, ny = dy * dinv;
    const float rd = rij - dn;
    // body-contact overlap from a Newton-refined distance: one ulp of dist is 1.4e-5 of the k1 / k2 force (stepcommon.h dist_refined)
    const float comp = fmaxf(0.0f, rij - fmaf(fmaf(-dn, dn, d2h), 0.5f * dinv, dn));
    if (soc == 2) {
        const float ivx = P[12] * (vx - hvx) - nx, ivy = P[12] * (vy - hvy) - ny;
        const float inorm = sqrtf(ivx * ivx + ivy * ivy);
        const float ix = ivx / inorm, iy = ivy / inorm;
        const float th = bound_angle(atan2f(ny, nx) - atan2f(iy, ix) + PI_F);
        const float k = (th > 0.0f) ? 1.0f : ((th < 0.0f) ? -1.0f : 0.0f);
        const float hxv = -iy, hyv = ix;
        const float F = P[13] * inorm;
        const float dvh = (hvx - vx) * hxv + (hvy - vy) * hyv;
        const float e0 = P[9] * expf(-dn / F);
        const float t1 = P[15] * F * th, t2 = P[14] * F * th;
        const float e1 = expf(-(t1 * t1)), e2 = k * expf(-(t2 * t2));
        tx_out = -(e0 * (e1 * ix + e2 * hxv) + P[10] * comp * ix + P[11] * comp * dvh * hxv);
        ty_out = -(e0 * (e1 * iy + e2 * hyv) + P[10] * comp * iy + P[11] * comp * dvh * hyv);
    } else {
        const float tx = -ny, ty = nx;
        const float dv = (hvx - vx) * tx + (hvy - vy) * ty;
        const float fn = P[1] * __expf(rd / P[3]) + P[10] * comp;
        float ft = P[11] * comp * dv;
        if (soc == 1) ft += P[5] * __expf(rd / P[7]);
        tx_out = fn * nx + ft * tx;
        ty_out = fn * ny + ft * ty;
    }
}

// headed_agent_update_linear_velocity (motion_model_manager.py:143-145) at the head of compute_robot_forces: the heading's sine /
// cosine (kept for the body-frame projection) and, for a headed robot, the linear velocity from the body velocity
__device__ __forceinline__ void refresh_velocity(RState& r, bool headed, float& sn, float& cs)
{
#pragma clang fp contract(off)
    sincosf(r.yaw, &sn, &cs);
    if (headed) { r.vx = cs * r.bvx - sn * r.bvy; r.vy = sn * r.bvx + cs * r.bvy; }
}

// obstacle force of the single-agent functions (forces.py:27-53): one closest point per polygon (the LAST nearest segment point,
// `<=`, obstacle.py:53-66), Helbing averaged over the polygons, Guo a plain sum.  ob: [O][Smax][2][2] NaN-padded, or null.
__device__ __forceinline__ void obstacle_force(const float* ob, int O, int Smax, int soc, const float* P, float px, float py, float vx, float vy,
                                               float rme, float& fox, float& foy)
{
#pragma clang fp contract(off)
    fox = 0.0f; foy = 0.0f;
    if (ob == nullptr || O <= 0) return;
    for (int o = 0; o < O; ++o) {
        float bx = 0.0f, by = 0.0f, bd = 10000.0f;
        for (int sg = 0; sg < Smax; ++sg) {
            const float* q = ob + ((long)o * Smax + sg) * 4;
            const float ax = q[0], ay = q[1], ex = q[2], ey = q[3];
            if (isnan(ax) || isnan(ay) || isnan(ex) || isnan(ey)) continue;
            const float sx = ex - ax, sy = ey - ay;
            const float len = sqrtf(sx * sx + sy * sy);
            float t = ((px - ax) * sx + (py - ay) * sy) / (len * len);
            t = fminf(fmaxf(0.0f, t), 1.0f);
            const float hx = ax + t * sx, hy = ay + t * sy;
            const float d = sqrtf((hx - px) * (hx - px) + (hy - py) * (hy - py));
            if (d <= bd) { bx = hx; by = hy; bd = d; }
        }
        const float dx = px - bx, dy = py - by;
        const float dn = sqrtf(dx * dx + dy * dy);
        const float nx = dx / dn, ny = dy / dn, tx = -ny, ty = nx;
        const float dv = -(vx * tx + vy * ty);
        const float rd = rme - dn;
        const float comp = fmaxf(0.0f, rd);
        const float fn = P[2] * expf(rd / P[4]) + P[10] * comp;
        float ft;
        if (soc == 1) ft = (-P[6] * expf(rd / P[8]) - P[11] * comp) * dv;
        else ft = -P[11] * comp * dv;
        fox += fn * nx + ft * tx;
        foy += fn * ny + ft * ty;
    }
    if (soc != 1) { fox /= (float)O; foy /= (float)O; }   // Guo's single-agent obstacle force is not averaged
}

// the right-hand side of the robot's RK45 solve (motion_model_manager.py:661-687) from the forces of compute_robot_forces: the
// desired force (kept within one radius of the goal), then ydot = [v, F / m] or [R bv, omega, F_body / m, torque / I]
template <int NS>
__device__ __forceinline__ void derivative(RState& r, int type, const float* P, float fsx, float fsy, float fox, float foy, float sn, float cs,
                                           float (&out)[NS])
{
#pragma clang fp contract(off)
    const bool torque_new = type >= 6;
    {
        const float ddx = r.gx - r.px, ddy = r.gy - r.py;
        const float dist = sqrtf(ddx * ddx + ddy * ddy);
        if (dist > r.radius) {
            r.fdx = r.mass * (ddx / dist * r.vd - r.vx) / P[0];
            r.fdy = r.mass * (ddy / dist * r.vd - r.vy) / P[0];
        }
    }
    const float fdx = r.fdx, fdy = r.fdy;
    if constexpr (NS == 4) {
        out[0] = r.vx; out[1] = r.vy;
        out[2] = (fdx + fox + fsx) / r.mass; out[3] = (fdy + foy + fsy) / r.mass;
    } else {
        const float inertia = 0.5f * r.mass * r.radius * r.radius;
        const float tx = torque_new ? fdx + fox + fsx : fdx, ty = torque_new ? fdy + foy + fsy : fdy;
        const float tn = sqrtf(tx * tx + ty * ty);
        const float k_theta = inertia * P[19] * tn;
        const float k_omega = inertia * (1.0f + P[18]) * sqrtf(P[19] * tn / P[18]);
        const float torque = -k_theta * bound_angle(r.yaw - atan2f(ty, tx)) - k_omega * r.om;
        const float g0 = (fdx + fox + fsx) * cs + (fdy + foy + fsy) * sn;
        const float g1 = P[16] * ((fox + fsx) * -sn + (foy + fsy) * cs) - P[17] * r.bvy;
        out[0] = cs * r.bvx - sn * r.bvy; out[1] = sn * r.bvx + cs * r.bvy; out[2] = r.om;
        out[3] = g0 / r.mass; out[4] = g1 / r.mass; out[5] = torque / inertia;
    }
}

// desired force (forces.py:9-16; within one radius of the goal the previous one is kept), total force, torque (forces.py:279-290)
// and the Euler update (motion_model_manager.py:72-86: position first, then velocity, speed clamp; headed: yaw, body velocity,
// angular velocity, linear velocity from the NEW yaw).  (fsx, fsy): the summed social force; (fox, foy): the obstacle force.
__device__ __forceinline__ void integrate(RState& r, int type, const float* P, float fsx, float fsy, float fox, float foy, float sn, float cs,
                                          float dt, int just_velocities)
{
#pragma clang fp contract(off)
    const bool headed = type >= 3, torque_new = type >= 6;
    {
        const float ddx = r.gx - r.px, ddy = r.gy - r.py;
        const float dist = sqrtf(ddx * ddx + ddy * ddy);
        if (dist > r.radius) {
            r.fdx = r.mass * (ddx / dist * r.vd - r.vx) / P[0];
            r.fdy = r.mass * (ddy / dist * r.vd - r.vy) / P[0];
        }
    }
    const float fdx = r.fdx, fdy = r.fdy;
    float npx, npy, nyaw = r.yaw, nvx, nvy, nbx = r.bvx, nby = r.bvy, nom = r.om;
    if (!headed) {
        const float gfx = fdx + fox + fsx, gfy = fdy + foy + fsy;
        npx = just_velocities ? r.px : r.px + r.vx * dt; npy = just_velocities ? r.py : r.py + r.vy * dt;
        nvx = r.vx + gfx / r.mass * dt; nvy = r.vy + gfy / r.mass * dt;
        const float sp = sqrtf(nvx * nvx + nvy * nvy);
        if (sp > r.vd) { nvx = nvx / sp * r.vd; nvy = nvy / sp * r.vd; }
    } else {
        const float inertia = 0.5f * r.mass * r.radius * r.radius;
        const float tx = torque_new ? fdx + fox + fsx : fdx, ty = torque_new ? fdy + foy + fsy : fdy;
        const float tn = sqrtf(tx * tx + ty * ty);
        const float k_theta = inertia * P[19] * tn;
        const float k_omega = inertia * (1.0f + P[18]) * sqrtf(P[19] * tn / P[18]);
        const float torque = -k_theta * bound_angle(r.yaw - atan2f(ty, tx)) - k_omega * r.om;
        // global_force = [ (fd + fo + fs) . R[:,0] ,  ko * (fo + fs) . R[:,1] - kd * body_velocity[1] ]
        const float g0 = (fdx + fox + fsx) * cs + (fdy + foy + fsy) * sn;
        const float g1 = P[16] * ((fox + fsx) * -sn + (foy + fsy) * cs) - P[17] * r.bvy;
        npx = just_velocities ? r.px : r.px + r.vx * dt; npy = just_velocities ? r.py : r.py + r.vy * dt;
        nyaw = just_velocities ? r.yaw : bound_angle(r.yaw + r.om * dt);
        nbx = r.bvx + g0 / r.mass * dt; nby = r.bvy + g1 / r.mass * dt;
        nom = r.om + torque / inertia * dt;
        const float sp = sqrtf(nbx * nbx + nby * nby);
        if (sp > r.vd) { nbx = nbx / sp * r.vd; nby = nby / sp * r.vd; }
        float s2, c2;
        sincosf(nyaw, &s2, &c2);
        nvx = c2 * nbx - s2 * nby; nvy = s2 * nbx + c2 * nby;
    }
    r.px = npx; r.py = npy; r.yaw = nyaw; r.vx = nvx; r.vy = nvy; r.bvx = nbx; r.bvy = nby; r.om = nom;
}

} // namespace rmodel
