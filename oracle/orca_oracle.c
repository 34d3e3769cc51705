/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.   *** PARITY UNPINNED ***
 *
 * CPU restatement of the ORCA branch of the reference's pedestrian update.  The arithmetic of that
 * branch does not live in /root/reference: it is the third-party RVO2 library (C++), reached through
 * the Cython wrapper Python-RVO2 (`import rvo2`, github.com/sybrenstuvel/Python-RVO2 wrapping RVO2
 * Library v2.0.x, Apache-2.0).  The reference neither vendors nor pins it (requirements.txt:1-10 omits
 * it; README.md:80 links it) and it is not installed here, so no golden vector can be produced and
 * this file cannot be checked against the real library: it restates the published algorithm
 * (van den Berg, Guy, Lin, Manocha, "Reciprocal n-body collision avoidance", ISRR 2009; RVO2 v2.0.2
 * Agent::computeNeighbors / computeNewVelocity / linearProgram1-3 / update) in float32 as RVO2 does,
 * and is anchored on the reference's own call sites:
 *   motion_model_manager.py:14        ORCA_DEFAULTS neighborDist=10, maxNeighbors=10, timeHorizon=5, timeHorizonObst=5
 *   motion_model_manager.py:237-246   simulator / agent creation (radius + 0.01, maxSpeed = desired_speed)
 *   motion_model_manager.py:385-394   setTimeStep(dt); doStep(); read back v, p; update_goals_orca
 *   motion_model_manager.py:125-133   goal rotation (strict <, :66-70) and preferred velocity
 *   motion_model_manager.py:105-114   set_state_orca (robot agent overwritten after each step)
 *   motion_model_manager.py:407-422   respawn (:416 setAgentPosition)
 * Tests pin it with analytic known answers and with a brute-force f64 solver of the same
 * half-plane programme (tests/test_orca_oracle.py).
 *
 * Differences from RVO2 that cannot change a result: neighbours are found by brute force in index
 * order instead of a kd-tree (same set; only the order of exactly tied distances could differ, for
 * N <= 10 = MAX_LEAF_SIZE even that is identical); static obstacles (ORCA obstacle lines) are not
 * modelled (every Gym scenario has walls == [], social_nav_sim.py:296,355,428).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RVO_EPSILON 0.00001f
#define ORCA_MAX_NEIGHBORS 32

typedef struct { float px, py, dx, dy; } orca_line;

static inline float det2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

/* RVO2 linearProgram1 */
static int lp1(const orca_line* L, int lineNo, float radius, float ox, float oy, int dirOpt, float* rx, float* ry)
{
    const float dot = L[lineNo].px * L[lineNo].dx + L[lineNo].py * L[lineNo].dy;
    const float disc = dot * dot + radius * radius - (L[lineNo].px * L[lineNo].px + L[lineNo].py * L[lineNo].py);
    if (disc < 0.0f) return 0;
    const float sq = sqrtf(disc);
    float tL = -dot - sq, tR = -dot + sq;
    for (int i = 0; i < lineNo; ++i) {
        const float den = det2(L[lineNo].dx, L[lineNo].dy, L[i].dx, L[i].dy);
        const float num = det2(L[i].dx, L[i].dy, L[lineNo].px - L[i].px, L[lineNo].py - L[i].py);
        if (fabsf(den) <= RVO_EPSILON) {
            if (num < 0.0f) return 0;
            continue;
        }
        const float t = num / den;
        if (den >= 0.0f) tR = fminf(tR, t); else tL = fmaxf(tL, t);
        if (tL > tR) return 0;
    }
    float t;
    if (dirOpt) {
        t = (ox * L[lineNo].dx + oy * L[lineNo].dy > 0.0f) ? tR : tL;
    } else {
        t = L[lineNo].dx * (ox - L[lineNo].px) + L[lineNo].dy * (oy - L[lineNo].py);
        if (t < tL) t = tL; else if (t > tR) t = tR;
    }
    *rx = L[lineNo].px + t * L[lineNo].dx;
    *ry = L[lineNo].py + t * L[lineNo].dy;
    return 1;
}

/* RVO2 linearProgram2 */
static int lp2(const orca_line* L, int nl, float radius, float ox, float oy, int dirOpt, float* rx, float* ry)
{
    if (dirOpt) { *rx = ox * radius; *ry = oy * radius; }
    else if (ox * ox + oy * oy > radius * radius) {
        const float nrm = sqrtf(ox * ox + oy * oy);
        *rx = ox / nrm * radius; *ry = oy / nrm * radius;
    } else { *rx = ox; *ry = oy; }
    for (int i = 0; i < nl; ++i) {
        if (det2(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry) > 0.0f) {
            const float tx = *rx, ty = *ry;
            if (!lp1(L, i, radius, ox, oy, dirOpt, rx, ry)) { *rx = tx; *ry = ty; return i; }
        }
    }
    return nl;
}

/* RVO2 linearProgram3 (numObstLines = 0) */
static void lp3(const orca_line* L, int nl, int begin, float radius, float* rx, float* ry)
{
    float distance = 0.0f;
    orca_line proj[ORCA_MAX_NEIGHBORS];
    for (int i = begin; i < nl; ++i) {
        if (det2(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry) > distance) {
            int np = 0;
            for (int j = 0; j < i; ++j) {
                orca_line ln;
                const float d = det2(L[i].dx, L[i].dy, L[j].dx, L[j].dy);
                if (fabsf(d) <= RVO_EPSILON) {
                    if (L[i].dx * L[j].dx + L[i].dy * L[j].dy > 0.0f) continue;
                    ln.px = 0.5f * (L[i].px + L[j].px); ln.py = 0.5f * (L[i].py + L[j].py);
                } else {
                    const float s = det2(L[j].dx, L[j].dy, L[i].px - L[j].px, L[i].py - L[j].py) / d;
                    ln.px = L[i].px + s * L[i].dx; ln.py = L[i].py + s * L[i].dy;
                }
                const float ex = L[j].dx - L[i].dx, ey = L[j].dy - L[i].dy;
                const float en = sqrtf(ex * ex + ey * ey);
                ln.dx = ex / en; ln.dy = ey / en;
                proj[np++] = ln;
            }
            const float tx = *rx, ty = *ry;
            if (lp2(proj, np, radius, -L[i].dy, L[i].dx, 1, rx, ry) < np) { *rx = tx; *ry = ty; }
            distance = det2(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry);
        }
    }
}

/*
 * One RVO2 doStep for one world of `na` agents (Jacobi: every agent reads the old state).
 *   pos, vel, pref [na][2]; radius, maxspeed [na] (radius already includes the +0.01 (+safety)).
 *   out_vel [na][2]; positions are advanced by the caller.
 * lines_out (optional): [na][max_nb] lines for inspection, nlines_out [na].
 */
void orc_orca_new_velocities(int na, const float* pos, const float* vel, const float* pref, const float* radius,
                             const float* maxspeed, float neighbor_dist, int max_nb, float time_horizon,
                             float time_step, float* out_vel, orca_line* lines_out, int* nlines_out)
{
    if (max_nb > ORCA_MAX_NEIGHBORS) max_nb = ORCA_MAX_NEIGHBORS;
    for (int a = 0; a < na; ++a) {
        /* Agent::computeNeighbors + insertAgentNeighbor, brute force in index order */
        float nd[ORCA_MAX_NEIGHBORS]; int ni[ORCA_MAX_NEIGHBORS]; int cnt = 0;
        float rangeSq = neighbor_dist * neighbor_dist;
        if (max_nb > 0) {
            for (int b = 0; b < na; ++b) {
                if (b == a) continue;
                const float ddx = pos[2 * a] - pos[2 * b], ddy = pos[2 * a + 1] - pos[2 * b + 1];
                const float dsq = ddx * ddx + ddy * ddy;
                if (dsq < rangeSq) {
                    if (cnt < max_nb) ++cnt;
                    int i = cnt - 1;
                    while (i != 0 && dsq < nd[i - 1]) { nd[i] = nd[i - 1]; ni[i] = ni[i - 1]; --i; }
                    nd[i] = dsq; ni[i] = b;
                    if (cnt == max_nb) rangeSq = nd[cnt - 1];
                }
            }
        }
        /* Agent::computeNewVelocity, agent lines */
        orca_line L[ORCA_MAX_NEIGHBORS];
        const float invT = 1.0f / time_horizon;
        const float vx = vel[2 * a], vy = vel[2 * a + 1];
        for (int k = 0; k < cnt; ++k) {
            const int b = ni[k];
            const float rpx = pos[2 * b] - pos[2 * a], rpy = pos[2 * b + 1] - pos[2 * a + 1];
            const float rvx = vx - vel[2 * b], rvy = vy - vel[2 * b + 1];
            const float distSq = rpx * rpx + rpy * rpy;
            const float R = radius[a] + radius[b];
            const float RSq = R * R;
            float dx, dy, ux, uy;
            if (distSq > RSq) {
                const float wx = rvx - invT * rpx, wy = rvy - invT * rpy;
                const float wLenSq = wx * wx + wy * wy;
                const float dot1 = wx * rpx + wy * rpy;
                if (dot1 < 0.0f && dot1 * dot1 > RSq * wLenSq) {
                    const float wLen = sqrtf(wLenSq);
                    const float uwx = wx / wLen, uwy = wy / wLen;
                    dx = uwy; dy = -uwx;
                    const float s = R * invT - wLen;
                    ux = s * uwx; uy = s * uwy;
                } else {
                    const float leg = sqrtf(distSq - RSq);
                    if (det2(rpx, rpy, wx, wy) > 0.0f) {
                        dx = (rpx * leg - rpy * R) / distSq; dy = (rpx * R + rpy * leg) / distSq;
                    } else {
                        dx = -(rpx * leg + rpy * R) / distSq; dy = -(-rpx * R + rpy * leg) / distSq;
                    }
                    const float dot2 = rvx * dx + rvy * dy;
                    ux = dot2 * dx - rvx; uy = dot2 * dy - rvy;
                }
            } else {
                const float invDt = 1.0f / time_step;
                const float wx = rvx - invDt * rpx, wy = rvy - invDt * rpy;
                const float wLen = sqrtf(wx * wx + wy * wy);
                const float uwx = wx / wLen, uwy = wy / wLen;
                dx = uwy; dy = -uwx;
                const float s = R * invDt - wLen;
                ux = s * uwx; uy = s * uwy;
            }
            L[k].px = vx + 0.5f * ux; L[k].py = vy + 0.5f * uy; L[k].dx = dx; L[k].dy = dy;
        }
        float rx, ry;
        const int failed = lp2(L, cnt, maxspeed[a], pref[2 * a], pref[2 * a + 1], 0, &rx, &ry);
        if (failed < cnt) lp3(L, cnt, failed, maxspeed[a], &rx, &ry);
        out_vel[2 * a] = rx; out_vel[2 * a + 1] = ry;
        if (lines_out) {
            memcpy(lines_out + (size_t)a * max_nb, L, sizeof(orca_line) * cnt);
            nlines_out[a] = cnt;
        }
    }
}

/*
 * n_substeps of the reference's ORCA branch for one world, on the [rows][13] rows the rest of the
 * path uses (columns 5:7 hold the preferred velocity RVO2 keeps between steps):
 *   doStep (all rows incl. a visible robot) ; humans: v,p <- sim ; goal rotation (strict <) ;
 *   prefVel ; robot row <- true robot state ; respawn.
 * The Gym loop moves the robot first (robot.step) but the simulator's robot agent is only
 * overwritten AFTER doStep (motion_model_manager.py:389), so humans see the robot one substep late.
 * margin[rows]: what is added to the radius (0.01 or 0.01 + safety_space).
 */
void orc_orca_step_block(float* S, float* goals, int G, int rows, int robot_visible, const float* margin,
                         float* robot, const float* action, float dt, int n_substeps, float neighbor_dist,
                         int max_nb, float time_horizon, int respawn, float bound_x, float bound_y)
{
    const int n = rows - (robot_visible ? 1 : 0);
    float* pos = (float*)malloc(sizeof(float) * rows * 9);
    float *vel = pos + 2 * rows, *pref = vel + 2 * rows, *rad = pref + 2 * rows, *vmax = rad + rows, *nv = vmax + rows;
    float* nvv = (float*)malloc(sizeof(float) * rows * 2);
    (void)nv;
    for (int s = 0; s < n_substeps; ++s) {
        if (robot && action) { robot[0] += action[0] * dt; robot[1] += action[1] * dt; robot[3] = action[0]; robot[4] = action[1]; }
        for (int i = 0; i < rows; ++i) {
            const float* r = S + 13 * i;
            pos[2 * i] = r[0]; pos[2 * i + 1] = r[1]; vel[2 * i] = r[3]; vel[2 * i + 1] = r[4];
            pref[2 * i] = r[5]; pref[2 * i + 1] = r[6]; rad[i] = r[8] + margin[i]; vmax[i] = r[12];
        }
        orc_orca_new_velocities(rows, pos, vel, pref, rad, vmax, neighbor_dist, max_nb, time_horizon, dt, nvv, NULL, NULL);
        for (int i = 0; i < n; ++i) {
            float* r = S + 13 * i;
            float* gi = goals + (size_t)i * G * 2;
            r[3] = nvv[2 * i]; r[4] = nvv[2 * i + 1];
            r[0] += r[3] * dt; r[1] += r[4] * dt;
            /* update_goals: strict <, list rotation over the non-NaN prefix */
            float ddx = gi[0] - r[0], ddy = gi[1] - r[1];
            if (sqrtf(ddx * ddx + ddy * ddy) < r[8]) {
                int k = G;
                for (int g = 0; g < G; ++g) if (isnan(gi[2 * g])) { k = g; break; }
                const float r0 = gi[0], r1 = gi[1];
                for (int g = 0; g + 1 < k; ++g) { gi[2 * g] = gi[2 * g + 2]; gi[2 * g + 1] = gi[2 * g + 3]; }
                if (k > 0) { gi[2 * (k - 1)] = r0; gi[2 * (k - 1) + 1] = r1; }
            }
            r[10] = gi[0]; r[11] = gi[1];
            ddx = gi[0] - r[0]; ddy = gi[1] - r[1];
            const float nrm = sqrtf(ddx * ddx + ddy * ddy);
            if (nrm > r[12]) { r[5] = ddx / nrm; r[6] = ddy / nrm; } else { r[5] = ddx; r[6] = ddy; }
        }
        if (robot_visible && robot) { /* set_state_orca(robot): position, velocity (pref vel unused) */
            float* r = S + 13 * n;
            r[0] = robot[0]; r[1] = robot[1]; r[3] = robot[3]; r[4] = robot[4];
        }
        if (respawn) {
            for (int i = 0; i < n; ++i) {
                float* r = S + 13 * i;
                float* gi = goals + (size_t)i * G * 2;
                const float ddx = r[0] - gi[0], ddy = r[1] - gi[1];
                if (sqrtf(ddx * ddx + ddy * ddy) < 3.0f) {
                    /* h.radius + h.safety_space: set_safety_space() only resizes the RVO agents for ORCA
                       (mmm.py:154-158), the humans' safety_space attribute stays 0 -> plain radii */
                    float mx = S[0], mr = S[8];
                    for (int j = 1; j < n; ++j) {
                        if (S[13 * j] > mx) mx = S[13 * j];
                        if (S[13 * j + 8] > mr) mr = S[13 * j + 8];
                    }
                    if (robot_visible && robot) {
                        if (robot[0] > mx) mx = robot[0];
                        if (robot[8] > mr) mr = robot[8];
                    }
                    const float x = mx + mr * 2.0f;
                    r[0] = x > bound_x ? x : bound_x;
                    if (r[1] >= 0) r[1] = r[1] < bound_y ? r[1] : bound_y; else r[1] = r[1] > -bound_y ? r[1] : -bound_y;
                    const float gx = gi[0], gy = r[1];
                    for (int g = 0; g < G; ++g) { gi[2 * g] = gx; gi[2 * g + 1] = gy; }
                    r[10] = gx; r[11] = gy;
                }
            }
        }
    }
    free(pos);
    free(nvv);
}

void orc_orca_step_block_batched(int W, float* S, float* goals, int G, int rows, int robot_visible,
                                 const float* margin, float* robot, const float* action, float dt, int n_substeps,
                                 float neighbor_dist, int max_nb, float time_horizon, int respawn, float bound_x,
                                 float bound_y, int threads)
{
    const int n = rows - (robot_visible ? 1 : 0);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
#endif
    for (int w = 0; w < W; ++w)
        orc_orca_step_block(S + (size_t)w * rows * 13, goals + (size_t)w * n * G * 2, G, rows, robot_visible,
                            margin + (size_t)w * rows, robot ? robot + (size_t)w * 13 : NULL,
                            action ? action + (size_t)w * 2 : NULL, dt, n_substeps, neighbor_dist, max_nb,
                            time_horizon, respawn, bound_x, bound_y);
}
