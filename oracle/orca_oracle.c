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
 * Agent::computeNeighbors / computeNewVelocity / linearProgram1-3 / update) in float32 as RVO2 does
 * (a vector divided by a scalar is multiplied by the scalar's reciprocal, as RVO2's Vector2::operator/ does),
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
 * N <= 10 = MAX_LEAF_SIZE even that is identical).
 * Static obstacles (motion_model_manager.py:244-246 addObstacle / processObstacles): the obstacle
 * ORCA lines of Agent::computeNewVelocity and linearProgram3 with numObstLines are restated below.  processObstacles() =
 * KdTree::buildObstacleTree SPLITS every edge that straddles the line of a node's splitting edge into collinear pieces: that part is
 * restated on the host (oracle/crowd_oracle.py split_obstacles; the package's own: rvo2.split_obstacles_kdtree), so the vertex records this
 * file is given already hold the pieces; the obstacle neighbours are then found by brute force over them (the kd-tree is a search
 * structure: same set; only the order of exactly tied distances could differ).  No Gym scenario has walls (social_nav_sim.py:296,355,428).
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
#define ORCA_MAX_OBST 32
#define ORCA_MAX_LINES (ORCA_MAX_NEIGHBORS + ORCA_MAX_OBST)

typedef struct { float px, py, dx, dy; } orca_line;

/* Hooks of the PROBE instantiation (oracle/orca_oracle_probe.c: the same statements with every division, square root and two-term
 * product sum perturbed by a few ulps and every decision recorded -- a classification aid for tests/orca_fast_parity.py).  Here they
 * are the plain float32 operations and comparisons, so this file's arithmetic is exactly what it was without them. */
#ifndef O_HOOKS
#define O_DIV(a, b) ((a) / (b))
#define O_SQRT(x) sqrtf(x)
#define O_DET(ax, ay, bx, by) ((ax) * (by) - (ay) * (bx))
#define O_DOT(ax, ay, bx, by) ((ax) * (bx) + (ay) * (by))
#define O_MAD(p, t, d) ((p) + (t) * (d))
#define O_GT(kind, a, b) ((a) > (b))
#define O_LT(kind, a, b) ((a) < (b))
#define O_GE(kind, a, b) ((a) >= (b))
#define O_LE(kind, a, b) ((a) <= (b))
#define O_NOTE(kind, flag)
#define O_AGENT_BEGIN(a)
#endif
/* decision kinds (the probe's trace) */
enum { OD_NB_RANGE = 1, OD_NB_ORDER, OD_COLLIDING, OD_CIRCLE_SIDE, OD_CIRCLE_CONE, OD_WHICH_LEG, OD_LP2_CLIP, OD_LP2_VIOLATED, OD_LP1_DISC,
       OD_LP1_PARALLEL, OD_LP1_PARALLEL_SIDE, OD_LP1_DEN_SIGN, OD_LP1_EMPTY, OD_LP1_DIROPT, OD_LP1_CLAMP_L, OD_LP1_CLAMP_R, OD_LP3_VIOLATED,
       OD_LP3_PARALLEL, OD_LP3_SAME_DIR, OD_LP2_FAILED, OD_LP3_LP2_FAILED, OD_KINDS };

static inline float det2(float ax, float ay, float bx, float by) { return O_DET(ax, ay, bx, by); }

/* RVO2 linearProgram1 */
static int lp1(const orca_line* L, int lineNo, float radius, float ox, float oy, int dirOpt, float* rx, float* ry)
{
    const float dot = O_DOT(L[lineNo].px, L[lineNo].py, L[lineNo].dx, L[lineNo].dy);
    const float disc = dot * dot + radius * radius - O_DOT(L[lineNo].px, L[lineNo].py, L[lineNo].px, L[lineNo].py);
    if (O_LT(OD_LP1_DISC, disc, 0.0f)) return 0;
    const float sq = O_SQRT(disc);
    float tL = -dot - sq, tR = -dot + sq;
    for (int i = 0; i < lineNo; ++i) {
        const float den = det2(L[lineNo].dx, L[lineNo].dy, L[i].dx, L[i].dy);
        const float num = det2(L[i].dx, L[i].dy, L[lineNo].px - L[i].px, L[lineNo].py - L[i].py);
        if (O_LE(OD_LP1_PARALLEL, fabsf(den), RVO_EPSILON)) {
            if (O_LT(OD_LP1_PARALLEL_SIDE, num, 0.0f)) return 0;
            continue;
        }
        const float t = O_DIV(num, den);
        if (O_GE(OD_LP1_DEN_SIGN, den, 0.0f)) tR = fminf(tR, t); else tL = fmaxf(tL, t);
        if (O_GT(OD_LP1_EMPTY, tL, tR)) return 0;
    }
    float t;
    if (dirOpt) {
        t = O_GT(OD_LP1_DIROPT, O_DOT(ox, oy, L[lineNo].dx, L[lineNo].dy), 0.0f) ? tR : tL;
    } else {
        t = O_DOT(L[lineNo].dx, L[lineNo].dy, ox - L[lineNo].px, oy - L[lineNo].py);
        if (O_LT(OD_LP1_CLAMP_L, t, tL)) t = tL; else if (O_GT(OD_LP1_CLAMP_R, t, tR)) t = tR;
    }
    *rx = O_MAD(L[lineNo].px, t, L[lineNo].dx);
    *ry = O_MAD(L[lineNo].py, t, L[lineNo].dy);
    return 1;
}

/* RVO2 linearProgram2 */
static int lp2(const orca_line* L, int nl, float radius, float ox, float oy, int dirOpt, float* rx, float* ry)
{
    if (dirOpt) { *rx = ox * radius; *ry = oy * radius; }
    else if (O_GT(OD_LP2_CLIP, O_DOT(ox, oy, ox, oy), radius * radius)) {
        const float nrm = O_SQRT(O_DOT(ox, oy, ox, oy));
        const float inv = O_DIV(1.0f, nrm);       /* RVO2's Vector2 / float multiplies by the reciprocal (Vector2.h) */
        *rx = ox * inv * radius; *ry = oy * inv * radius;
    } else { *rx = ox; *ry = oy; }
    for (int i = 0; i < nl; ++i) {
        if (O_GT(OD_LP2_VIOLATED, det2(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry), 0.0f)) {
            const float tx = *rx, ty = *ry;
            if (!lp1(L, i, radius, ox, oy, dirOpt, rx, ry)) { *rx = tx; *ry = ty; return i; }
        }
    }
    return nl;
}

/* RVO2 linearProgram3: the first numObst lines (static obstacles) are hard constraints, copied unprojected */
static void lp3(const orca_line* L, int nl, int numObst, int begin, float radius, float* rx, float* ry)
{
    float distance = 0.0f;
    orca_line proj[ORCA_MAX_LINES];
    for (int i = begin; i < nl; ++i) {
        if (O_GT(OD_LP3_VIOLATED, det2(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry), distance)) {
            int np = 0;
            for (int j = 0; j < numObst; ++j) proj[np++] = L[j];
            for (int j = numObst; j < i; ++j) {
                orca_line ln;
                const float d = det2(L[i].dx, L[i].dy, L[j].dx, L[j].dy);
                if (O_LE(OD_LP3_PARALLEL, fabsf(d), RVO_EPSILON)) {
                    if (O_GT(OD_LP3_SAME_DIR, O_DOT(L[i].dx, L[i].dy, L[j].dx, L[j].dy), 0.0f)) continue;
                    ln.px = 0.5f * (L[i].px + L[j].px); ln.py = 0.5f * (L[i].py + L[j].py);
                } else {
                    const float s = O_DIV(det2(L[j].dx, L[j].dy, L[i].px - L[j].px, L[i].py - L[j].py), d);
                    ln.px = O_MAD(L[i].px, s, L[i].dx); ln.py = O_MAD(L[i].py, s, L[i].dy);
                }
                const float ex = L[j].dx - L[i].dx, ey = L[j].dy - L[i].dy;
                const float en = O_SQRT(O_DOT(ex, ey, ex, ey));
                const float inv = O_DIV(1.0f, en);
                ln.dx = ex * inv; ln.dy = ey * inv;
                proj[np++] = ln;
            }
            const float tx = *rx, ty = *ry;
            const int lp2_failed = lp2(proj, np, radius, -L[i].dy, L[i].dx, 1, rx, ry) < np;
            O_NOTE(OD_LP3_LP2_FAILED, lp2_failed);
            if (lp2_failed) { *rx = tx; *ry = ty; }
            distance = det2(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry);
        }
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * Static obstacles.  One record per polygon vertex, as RVOSimulator::addObstacle builds them:
 *   point, unitDir = normalize(next.point - point), isConvex = leftOf(prev, this, next) >= 0 (2 vertices: convex),
 *   next / prev vertex indices.  The edge of vertex v runs from v to next(v); agents stay on its right side
 *   (polygons are given counter-clockwise, obstacle.py:11-12).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { float px, py, ux, uy, convex, next, prev, pad; } orca_vertex;

static inline float absSq2(float x, float y) { return x * x + y * y; }

static float dist_sq_point_segment(float ax, float ay, float bx, float by, float cx, float cy)
{
    const float r = ((cx - ax) * (bx - ax) + (cy - ay) * (by - ay)) / absSq2(bx - ax, by - ay);
    if (r < 0.0f) return absSq2(cx - ax, cy - ay);
    if (r > 1.0f) return absSq2(cx - bx, cy - by);
    return absSq2(cx - (ax + r * (bx - ax)), cy - (ay + r * (by - ay)));
}

/* Agent::computeNeighbors (obstacle part) + insertObstacleNeighbor, brute force over the edges in index order:
 * edges the agent is on the right of and closer than rangeSq, sorted by distance (strict <, ties keep order). */
static int obstacle_neighbors(const orca_vertex* V, int nv, float px, float py, float rangeSq, int* oi)
{
    float od[ORCA_MAX_OBST];
    int cnt = 0;
    for (int v = 0; v < nv; ++v) {
        const orca_vertex* o1 = V + v;
        const orca_vertex* o2 = V + (int)o1->next;
        const float agentLeftOfLine = det2(o1->px - px, o1->py - py, o2->px - o1->px, o2->py - o1->py);
        const float distSqLine = agentLeftOfLine * agentLeftOfLine / absSq2(o2->px - o1->px, o2->py - o1->py);
        if (distSqLine < rangeSq && agentLeftOfLine < 0.0f) {
            const float distSq = dist_sq_point_segment(o1->px, o1->py, o2->px, o2->py, px, py);
            if (distSq < rangeSq && cnt < ORCA_MAX_OBST) {
                int i = cnt++;
                while (i != 0 && distSq < od[i - 1]) { od[i] = od[i - 1]; oi[i] = oi[i - 1]; --i; }
                od[i] = distSq; oi[i] = v;
            }
        }
    }
    return cnt;
}

/* Agent::computeNewVelocity, "Create obstacle ORCA lines" */
static int obstacle_lines(const orca_vertex* V, const int* oi, int no, float px, float py, float vx, float vy,
                          float radius, float invT, orca_line* L)
{
    int nl = 0;
    for (int k = 0; k < no; ++k) {
        const orca_vertex* o1 = V + oi[k];
        const orca_vertex* o2 = V + (int)o1->next;
        const float r1x = o1->px - px, r1y = o1->py - py, r2x = o2->px - px, r2y = o2->py - py;
        /* already covered by a previous obstacle line? */
        int covered = 0;
        for (int j = 0; j < nl; ++j) {
            if (det2(invT * r1x - L[j].px, invT * r1y - L[j].py, L[j].dx, L[j].dy) - invT * radius >= -RVO_EPSILON &&
                det2(invT * r2x - L[j].px, invT * r2y - L[j].py, L[j].dx, L[j].dy) - invT * radius >= -RVO_EPSILON) {
                covered = 1;
                break;
            }
        }
        if (covered) continue;
        const float distSq1 = absSq2(r1x, r1y), distSq2 = absSq2(r2x, r2y), radiusSq = radius * radius;
        const float ovx = o2->px - o1->px, ovy = o2->py - o1->py;
        const float s = (-r1x * ovx + -r1y * ovy) / absSq2(ovx, ovy);
        const float distSqLine = absSq2(-r1x - s * ovx, -r1y - s * ovy);
        orca_line ln;
        if (s < 0.0f && distSq1 <= radiusSq) {            /* collision with left vertex; ignore if non-convex */
            if (o1->convex != 0.0f) {
                const float inv = 1.0f / sqrtf(absSq2(-r1y, r1x));
                ln.px = 0.0f; ln.py = 0.0f; ln.dx = -r1y * inv; ln.dy = r1x * inv;
                L[nl++] = ln;
            }
            continue;
        } else if (s > 1.0f && distSq2 <= radiusSq) {     /* collision with right vertex */
            if (o2->convex != 0.0f && det2(r2x, r2y, o2->ux, o2->uy) >= 0.0f) {
                const float inv = 1.0f / sqrtf(absSq2(-r2y, r2x));
                ln.px = 0.0f; ln.py = 0.0f; ln.dx = -r2y * inv; ln.dy = r2x * inv;
                L[nl++] = ln;
            }
            continue;
        } else if (s >= 0.0f && s < 1.0f && distSqLine <= radiusSq) { /* collision with the segment */
            ln.px = 0.0f; ln.py = 0.0f; ln.dx = -o1->ux; ln.dy = -o1->uy;
            L[nl++] = ln;
            continue;
        }
        /* no collision: legs */
        float llx, lly, rlx, rly;
        const orca_vertex* a1 = o1;
        const orca_vertex* a2 = o2;
        if (s < 0.0f && distSqLine <= radiusSq) {         /* obliquely viewed: left vertex defines the velocity obstacle */
            if (o1->convex == 0.0f) continue;
            a2 = o1;
            const float leg1 = sqrtf(distSq1 - radiusSq);
            const float inv1 = 1.0f / distSq1;
            llx = (r1x * leg1 - r1y * radius) * inv1; lly = (r1x * radius + r1y * leg1) * inv1;
            rlx = (r1x * leg1 + r1y * radius) * inv1; rly = (-r1x * radius + r1y * leg1) * inv1;
        } else if (s > 1.0f && distSqLine <= radiusSq) {  /* right vertex defines it */
            if (o2->convex == 0.0f) continue;
            a1 = o2;
            const float leg2 = sqrtf(distSq2 - radiusSq);
            const float inv2 = 1.0f / distSq2;
            llx = (r2x * leg2 - r2y * radius) * inv2; lly = (r2x * radius + r2y * leg2) * inv2;
            rlx = (r2x * leg2 + r2y * radius) * inv2; rly = (-r2x * radius + r2y * leg2) * inv2;
        } else {                                          /* usual situation */
            if (o1->convex != 0.0f) {
                const float leg1 = sqrtf(distSq1 - radiusSq);
                const float inv1 = 1.0f / distSq1;
                llx = (r1x * leg1 - r1y * radius) * inv1; lly = (r1x * radius + r1y * leg1) * inv1;
            } else { llx = -o1->ux; lly = -o1->uy; }     /* left leg extends the cut-off line */
            if (o2->convex != 0.0f) {
                const float leg2 = sqrtf(distSq2 - radiusSq);
                const float inv2 = 1.0f / distSq2;
                rlx = (r2x * leg2 + r2y * radius) * inv2; rly = (-r2x * radius + r2y * leg2) * inv2;
            } else { rlx = o1->ux; rly = o1->uy; }
        }
        /* legs never point into a neighbouring edge of a convex vertex: take that edge's cut-off line instead */
        const orca_vertex* leftNb = V + (int)a1->prev;
        int leftForeign = 0, rightForeign = 0;
        if (a1->convex != 0.0f && det2(llx, lly, -leftNb->ux, -leftNb->uy) >= 0.0f) { llx = -leftNb->ux; lly = -leftNb->uy; leftForeign = 1; }
        if (a2->convex != 0.0f && det2(rlx, rly, a2->ux, a2->uy) <= 0.0f) { rlx = a2->ux; rly = a2->uy; rightForeign = 1; }
        /* cut-off centres */
        const float lcx = invT * (a1->px - px), lcy = invT * (a1->py - py);
        const float rcx = invT * (a2->px - px), rcy = invT * (a2->py - py);
        const float cvx = rcx - lcx, cvy = rcy - lcy;
        const int same = (a1 == a2);
        /* project the current velocity on the velocity obstacle */
        const float t = same ? 0.5f : ((vx - lcx) * cvx + (vy - lcy) * cvy) / absSq2(cvx, cvy);
        const float tL = (vx - lcx) * llx + (vy - lcy) * lly;
        const float tR = (vx - rcx) * rlx + (vy - rcy) * rly;
        if ((t < 0.0f && tL < 0.0f) || (same && tL < 0.0f && tR < 0.0f)) { /* left cut-off circle */
            const float wx = vx - lcx, wy = vy - lcy, wn = sqrtf(absSq2(wx, wy));
            const float inv = 1.0f / wn;
            const float ux = wx * inv, uy = wy * inv;
            ln.dx = uy; ln.dy = -ux; ln.px = lcx + radius * invT * ux; ln.py = lcy + radius * invT * uy;
            L[nl++] = ln;
            continue;
        } else if (t > 1.0f && tR < 0.0f) {                                  /* right cut-off circle */
            const float wx = vx - rcx, wy = vy - rcy, wn = sqrtf(absSq2(wx, wy));
            const float inv = 1.0f / wn;
            const float ux = wx * inv, uy = wy * inv;
            ln.dx = uy; ln.dy = -ux; ln.px = rcx + radius * invT * ux; ln.py = rcy + radius * invT * uy;
            L[nl++] = ln;
            continue;
        }
        /* left leg, right leg or cut-off line, whichever is closest to the velocity */
        const float dCut = (t < 0.0f || t > 1.0f || same) ? INFINITY : absSq2(vx - (lcx + t * cvx), vy - (lcy + t * cvy));
        const float dLeft = (tL < 0.0f) ? INFINITY : absSq2(vx - (lcx + tL * llx), vy - (lcy + tL * lly));
        const float dRight = (tR < 0.0f) ? INFINITY : absSq2(vx - (rcx + tR * rlx), vy - (rcy + tR * rly));
        if (dCut <= dLeft && dCut <= dRight) {            /* cut-off line */
            ln.dx = -a1->ux; ln.dy = -a1->uy;
            ln.px = lcx + radius * invT * -ln.dy; ln.py = lcy + radius * invT * ln.dx;
            L[nl++] = ln;
        } else if (dLeft <= dRight) {                     /* left leg */
            if (leftForeign) continue;
            ln.dx = llx; ln.dy = lly;
            ln.px = lcx + radius * invT * -ln.dy; ln.py = lcy + radius * invT * ln.dx;
            L[nl++] = ln;
        } else {                                          /* right leg */
            if (rightForeign) continue;
            ln.dx = -rlx; ln.dy = -rly;
            ln.px = rcx + radius * invT * -ln.dy; ln.py = rcy + radius * invT * ln.dx;
            L[nl++] = ln;
        }
    }
    return nl;
}

/*
 * One RVO2 doStep for one world of `na` agents (Jacobi: every agent reads the old state).
 *   pos, vel, pref [na][2]; radius, maxspeed [na] (radius already includes the +0.01 (+safety)).
 *   out_vel [na][2]; positions are advanced by the caller.
 * lines_out (optional): [na][max_nb] lines for inspection, nlines_out [na].
 */
/* with static obstacles: verts [nv] orca_vertex records (or nv = 0); lines_out rows hold max_nb + ORCA_MAX_OBST lines,
 * nobst_out [na] = how many of them are obstacle lines (they come first) */
/* agent_params (optional): [na][4] = neighborDist, maxNeighbors, timeHorizon, timeHorizonObst of every agent -- RVO2 keeps the four per
 * agent (RVOSimulator::addAgent(position, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed, velocity), the
 * signature the reference calls at motion_model_manager.py:241); NULL = the scalar arguments for everyone (what the reference passes:
 * ORCA_DEFAULTS).  lines_out rows are sized by the scalar max_nb, which must then be the largest per-agent value. */
void orc_orca_new_velocities_pa(int na, const float* pos, const float* vel, const float* pref, const float* radius,
                                const float* maxspeed, float neighbor_dist_all, int max_nb_all, float time_horizon_all,
                                float time_horizon_obst_all, float time_step, const float* verts, int nv, float* out_vel,
                                orca_line* lines_out, int* nlines_out, int* nobst_out, const float* agent_params)
{
    const orca_vertex* V = (const orca_vertex*)verts;
    if (max_nb_all > ORCA_MAX_NEIGHBORS) max_nb_all = ORCA_MAX_NEIGHBORS;
    for (int a = 0; a < na; ++a) {
        O_AGENT_BEGIN(a);
        float neighbor_dist = neighbor_dist_all, time_horizon = time_horizon_all, time_horizon_obst = time_horizon_obst_all;
        int max_nb = max_nb_all;
        if (agent_params) {
            const float* ap = agent_params + 4 * (size_t)a;
            neighbor_dist = ap[0]; max_nb = (int)ap[1]; time_horizon = ap[2]; time_horizon_obst = ap[3];
            if (max_nb > max_nb_all) max_nb = max_nb_all;
        }
        orca_line L[ORCA_MAX_LINES];
        int numObst = 0;
        if (nv > 0) { /* rangeSq = sqr(timeHorizonObst * maxSpeed + radius) */
            int oi[ORCA_MAX_OBST];
            const float rng = time_horizon_obst * maxspeed[a] + radius[a];
            const int no = obstacle_neighbors(V, nv, pos[2 * a], pos[2 * a + 1], rng * rng, oi);
            numObst = obstacle_lines(V, oi, no, pos[2 * a], pos[2 * a + 1], vel[2 * a], vel[2 * a + 1], radius[a],
                                     1.0f / time_horizon_obst, L);
        }
        /* Agent::computeNeighbors + insertAgentNeighbor, brute force in index order */
        float nd[ORCA_MAX_NEIGHBORS]; int ni[ORCA_MAX_NEIGHBORS]; int cnt = 0;
        float rangeSq = neighbor_dist * neighbor_dist;
        if (max_nb > 0) {
            for (int b = 0; b < na; ++b) {
                if (b == a) continue;
                const float ddx = pos[2 * a] - pos[2 * b], ddy = pos[2 * a + 1] - pos[2 * b + 1];
                const float dsq = O_DOT(ddx, ddy, ddx, ddy);
                if (O_LT(OD_NB_RANGE, dsq, rangeSq)) {
                    if (cnt < max_nb) ++cnt;
                    int i = cnt - 1;
                    while (i != 0 && O_LT(OD_NB_ORDER, dsq, nd[i - 1])) { nd[i] = nd[i - 1]; ni[i] = ni[i - 1]; --i; }
                    nd[i] = dsq; ni[i] = b;
                    if (cnt == max_nb) rangeSq = nd[cnt - 1];
                }
            }
        }
        /* Agent::computeNewVelocity, agent lines (appended behind the obstacle lines) */
        const float invT = O_DIV(1.0f, time_horizon);
        const float vx = vel[2 * a], vy = vel[2 * a + 1];
        for (int k = 0; k < cnt; ++k) {
            const int b = ni[k];
            const float rpx = pos[2 * b] - pos[2 * a], rpy = pos[2 * b + 1] - pos[2 * a + 1];
            const float rvx = vx - vel[2 * b], rvy = vy - vel[2 * b + 1];
            const float distSq = O_DOT(rpx, rpy, rpx, rpy);
            const float R = radius[a] + radius[b];
            const float RSq = R * R;
            float dx, dy, ux, uy;
            if (O_GT(OD_COLLIDING, distSq, RSq)) {
                const float wx = rvx - invT * rpx, wy = rvy - invT * rpy;
                const float wLenSq = O_DOT(wx, wy, wx, wy);
                const float dot1 = O_DOT(wx, wy, rpx, rpy);
                if (O_LT(OD_CIRCLE_SIDE, dot1, 0.0f) && O_GT(OD_CIRCLE_CONE, dot1 * dot1, RSq * wLenSq)) {
                    const float wLen = O_SQRT(wLenSq);
                    const float inv = O_DIV(1.0f, wLen);
                    const float uwx = wx * inv, uwy = wy * inv;
                    dx = uwy; dy = -uwx;
                    const float s = R * invT - wLen;
                    ux = s * uwx; uy = s * uwy;
                } else {
                    const float leg = O_SQRT(distSq - RSq);
                    if (O_GT(OD_WHICH_LEG, det2(rpx, rpy, wx, wy), 0.0f)) {
                        const float inv = O_DIV(1.0f, distSq);
                        dx = (rpx * leg - rpy * R) * inv; dy = (rpx * R + rpy * leg) * inv;
                    } else {
                        const float inv = O_DIV(1.0f, distSq);
                        dx = -(rpx * leg + rpy * R) * inv; dy = -(-rpx * R + rpy * leg) * inv;
                    }
                    const float dot2 = O_DOT(rvx, rvy, dx, dy);
                    ux = dot2 * dx - rvx; uy = dot2 * dy - rvy;
                }
            } else {
                const float invDt = O_DIV(1.0f, time_step);
                const float wx = rvx - invDt * rpx, wy = rvy - invDt * rpy;
                const float wLen = O_SQRT(O_DOT(wx, wy, wx, wy));
                const float inv = O_DIV(1.0f, wLen);
                const float uwx = wx * inv, uwy = wy * inv;
                dx = uwy; dy = -uwx;
                const float s = R * invDt - wLen;
                ux = s * uwx; uy = s * uwy;
            }
            L[numObst + k].px = vx + 0.5f * ux; L[numObst + k].py = vy + 0.5f * uy; L[numObst + k].dx = dx; L[numObst + k].dy = dy;
        }
        const int total = numObst + cnt;
        float rx, ry;
        const int failed = lp2(L, total, maxspeed[a], pref[2 * a], pref[2 * a + 1], 0, &rx, &ry);
        O_NOTE(OD_LP2_FAILED, failed < total);
        if (failed < total) lp3(L, total, numObst, failed, maxspeed[a], &rx, &ry);
        out_vel[2 * a] = rx; out_vel[2 * a + 1] = ry;
        if (lines_out) {
            const int stride = nv > 0 ? max_nb + ORCA_MAX_OBST : max_nb;
            memcpy(lines_out + (size_t)a * stride, L, sizeof(orca_line) * total);
            nlines_out[a] = total;
            if (nobst_out) nobst_out[a] = numObst;
        }
    }
}

void orc_orca_new_velocities_obst(int na, const float* pos, const float* vel, const float* pref, const float* radius,
                                  const float* maxspeed, float neighbor_dist, int max_nb, float time_horizon,
                                  float time_horizon_obst, float time_step, const float* verts, int nv, float* out_vel,
                                  orca_line* lines_out, int* nlines_out, int* nobst_out)
{
    orc_orca_new_velocities_pa(na, pos, vel, pref, radius, maxspeed, neighbor_dist, max_nb, time_horizon, time_horizon_obst, time_step,
                               verts, nv, out_vel, lines_out, nlines_out, nobst_out, NULL);
}

void orc_orca_new_velocities(int na, const float* pos, const float* vel, const float* pref, const float* radius,
                             const float* maxspeed, float neighbor_dist, int max_nb, float time_horizon,
                             float time_step, float* out_vel, orca_line* lines_out, int* nlines_out)
{
    orc_orca_new_velocities_obst(na, pos, vel, pref, radius, maxspeed, neighbor_dist, max_nb, time_horizon, 5.0f, time_step,
                                 NULL, 0, out_vel, lines_out, nlines_out, NULL);
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
void orc_orca_step_block_pa(float* S, float* goals, int G, int rows, int robot_visible, const float* margin,
                            float* robot, const float* action, float dt, int n_substeps, float neighbor_dist,
                            int max_nb, float time_horizon, int respawn, float bound_x, float bound_y,
                            float time_horizon_obst, const float* verts, int nv, const float* agent_params /* [rows][4] or NULL */)
{
    const int n = rows - (robot_visible ? 1 : 0);
    float* pos = (float*)malloc(sizeof(float) * rows * 9);
    float *vel = pos + 2 * rows, *pref = vel + 2 * rows, *rad = pref + 2 * rows, *vmax = rad + rows;
    float* nvv = (float*)malloc(sizeof(float) * rows * 2);
    for (int s = 0; s < n_substeps; ++s) {
        if (robot && action) { robot[0] += action[0] * dt; robot[1] += action[1] * dt; robot[3] = action[0]; robot[4] = action[1]; }
        for (int i = 0; i < rows; ++i) {
            const float* r = S + 13 * i;
            pos[2 * i] = r[0]; pos[2 * i + 1] = r[1]; vel[2 * i] = r[3]; vel[2 * i + 1] = r[4];
            pref[2 * i] = r[5]; pref[2 * i + 1] = r[6]; rad[i] = r[8] + margin[i]; vmax[i] = r[12];
        }
        orc_orca_new_velocities_pa(rows, pos, vel, pref, rad, vmax, neighbor_dist, max_nb, time_horizon, time_horizon_obst, dt,
                                   verts, nv, nvv, NULL, NULL, NULL, agent_params);
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
            /* motion_model_manager.py:407-422, Python glue in float64: the flagged humans in index order, each behind everybody else -- the c-th
             * lands at max(max_x + 2 max_r, bound) + c * 2 max_r (the (c-1)-th is the rightmost by then); RVO2's float32 sees the result once
             * (:416 setAgentPosition).  The sum is formed in double and rounded once.
             * h.radius + h.safety_space: set_safety_space() only resizes the RVO agents for ORCA (mmm.py:154-158), the humans' safety_space
             * attribute stays 0 -> plain radii */
            double mx = S[0], mr = S[8];
            for (int j = 1; j < n; ++j) {
                if (S[13 * j] > mx) mx = S[13 * j];
                if (S[13 * j + 8] > mr) mr = S[13 * j + 8];
            }
            if (robot_visible && robot) {
                if (robot[0] > mx) mx = robot[0];
                if (robot[8] > mr) mr = robot[8];
            }
            int c = 0;
            for (int i = 0; i < n; ++i) {
                float* r = S + 13 * i;
                float* gi = goals + (size_t)i * G * 2;
                const float ddx = r[0] - gi[0], ddy = r[1] - gi[1];
                if (sqrtf(ddx * ddx + ddy * ddy) < 3.0f) {
                    double x0 = mx + mr * 2.0;
                    if (!(x0 > (double)bound_x)) x0 = (double)bound_x;
                    r[0] = (float)(x0 + (double)c * (mr * 2.0));
                    ++c;
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

void orc_orca_step_block_obst(float* S, float* goals, int G, int rows, int robot_visible, const float* margin,
                              float* robot, const float* action, float dt, int n_substeps, float neighbor_dist,
                              int max_nb, float time_horizon, int respawn, float bound_x, float bound_y,
                              float time_horizon_obst, const float* verts, int nv)
{
    orc_orca_step_block_pa(S, goals, G, rows, robot_visible, margin, robot, action, dt, n_substeps, neighbor_dist, max_nb, time_horizon,
                           respawn, bound_x, bound_y, time_horizon_obst, verts, nv, NULL);
}

void orc_orca_step_block(float* S, float* goals, int G, int rows, int robot_visible, const float* margin,
                         float* robot, const float* action, float dt, int n_substeps, float neighbor_dist,
                         int max_nb, float time_horizon, int respawn, float bound_x, float bound_y)
{
    orc_orca_step_block_obst(S, goals, G, rows, robot_visible, margin, robot, action, dt, n_substeps, neighbor_dist, max_nb,
                             time_horizon, respawn, bound_x, bound_y, 5.0f, NULL, 0);
}

void orc_orca_step_block_batched_pa(int W, float* S, float* goals, int G, int rows, int robot_visible,
                                    const float* margin, float* robot, const float* action, float dt, int n_substeps,
                                    float neighbor_dist, int max_nb, float time_horizon, int respawn, float bound_x,
                                    float bound_y, int threads, float time_horizon_obst, const float* verts, int nv,
                                    const float* agent_params /* [W][rows][4] or NULL */)
{
    const int n = rows - (robot_visible ? 1 : 0);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
#endif
    for (int w = 0; w < W; ++w)
        orc_orca_step_block_pa(S + (size_t)w * rows * 13, goals + (size_t)w * n * G * 2, G, rows, robot_visible,
                               margin + (size_t)w * rows, robot ? robot + (size_t)w * 13 : NULL,
                               action ? action + (size_t)w * 2 : NULL, dt, n_substeps, neighbor_dist, max_nb,
                               time_horizon, respawn, bound_x, bound_y, time_horizon_obst, verts, nv,
                               agent_params ? agent_params + (size_t)w * rows * 4 : NULL);
}

void orc_orca_step_block_batched_obst(int W, float* S, float* goals, int G, int rows, int robot_visible,
                                      const float* margin, float* robot, const float* action, float dt, int n_substeps,
                                      float neighbor_dist, int max_nb, float time_horizon, int respawn, float bound_x,
                                      float bound_y, int threads, float time_horizon_obst, const float* verts, int nv)
{
    orc_orca_step_block_batched_pa(W, S, goals, G, rows, robot_visible, margin, robot, action, dt, n_substeps, neighbor_dist, max_nb,
                                   time_horizon, respawn, bound_x, bound_y, threads, time_horizon_obst, verts, nv, NULL);
}

void orc_orca_step_block_batched(int W, float* S, float* goals, int G, int rows, int robot_visible,
                                 const float* margin, float* robot, const float* action, float dt, int n_substeps,
                                 float neighbor_dist, int max_nb, float time_horizon, int respawn, float bound_x,
                                 float bound_y, int threads)
{
    orc_orca_step_block_batched_obst(W, S, goals, G, rows, robot_visible, margin, robot, action, dt, n_substeps, neighbor_dist,
                                     max_nb, time_horizon, respawn, bound_x, bound_y, threads, 5.0f, NULL, 0);
}
