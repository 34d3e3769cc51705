// orca.hip -- ORCA (RVO2) crowd step for gfx950: the ORCA branch of MotionModelManager.update_humans
//   /root/reference/social_gym/src/motion_model_manager.py:385-394  (setTimeStep; doStep; read back; update_goals_orca)
//   :125-133 (goal rotation + preferred velocity), :105-114 (robot agent overwritten after the step),
//   :407-422 (respawn), :14 ORCA_DEFAULTS, :237-246 (agent radius = radius + 0.01 [+ safety space]).
// The arithmetic of doStep lives in the third-party RVO2 library (un-vendored, un-pinned, absent in the
// build image): it is restated here from the published algorithm -- Agent::computeNeighbors (brute force
// in index order instead of the kd-tree; same neighbour set), computeNewVelocity (ORCA half-planes),
// linearProgram1/2/3, update -- in float32 like RVO2.  PARITY UNPINNED (see oracle/orca_oracle.c).
// Static obstacles: addObstacle's vertex records come in cs_worlds.d_orca_vertices; the obstacle
// neighbours (brute force over the edges), the obstacle ORCA lines and linearProgram3 with hard obstacle lines run on the
// generic LDS-column path below.
//
// Mapping: lane = agent row, floor(64/rows) worlds per wavefront (one world per block of 256 / 512 lanes above 64 rows), the rows' (x, y, vx, vy, radius) in LDS,
// every lane's neighbour list / ORCA lines / LP3 projection lines in per-lane LDS columns (the 2-D
// linear programme is divergent by nature; lanes walk their own constraints).  n_substeps are fused in
// one launch.  IEEE divide / sqrt and no FMA contraction, to follow the CPU restatement op for op.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "orca_sortnet.h"
#include "robotstep.h"
#include "respawnx.h"

#pragma clang fp contract(off)

#ifdef CS_STAMPS
#define OSTAMP(k)                                                                            \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        unsigned long long t__;                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if (g_ost) { g_ost[k] += t__ - g_ost_last; }                                         \
        g_ost_last = t__;                                                                    \
    } while (0)
#else
#define OSTAMP(k) do { } while (0)
#endif

namespace {

using csimpl::fail;

constexpr float RVO_EPSILON = 0.00001f;
constexpr int KMAX = 16; // max_neighbors supported (ORCA_DEFAULTS uses 10)
constexpr int KOBST = 16; // obstacle edges kept per agent (the nearest ones), static-obstacle worlds only

struct OArgs {
    int W, n, rows, G, flags, nsub, wpb, K;
    float dt, neighbor_dist, time_horizon, bx, by;
    float* S; long as, fs;
    float* goals;
    const float* margin;
    float* robot;
    const float* action;
    float* peek_out;
    const int* world_flags;
    unsigned long long* stamps;
    // static obstacles: [nv][8] vertex records px, py, unitDir.x, unitDir.y, isConvex, next, prev, 0
    // (RVOSimulator::addObstacle), shared by all worlds; KO = obstacle lines kept per agent
    const float* verts;
    int nv, KO;
    float time_horizon_obst;
    int lp3_static;        // diagnostic A/B switch: linearProgram3 as the statically unrolled walk (lp3_fast10) instead of lp3_rows
    // RVO2 keeps neighborDist, maxNeighbors, timeHorizon, timeHorizonObst PER AGENT (RVOSimulator::addAgent, the call at
    // motion_model_manager.py:241): [W][rows][4] or null = the four scalars for everyone (what the reference passes: ORCA_DEFAULTS).
    // K above is then the largest maxNeighbors (the LDS columns are laid out for it); such worlds take the generic solve.
    const float* agent_params;
    int young_from;        // blocks from this index on are the YOUNGER wavefront of their SIMD (a grid of exactly two per SIMD; sfmstep_kernel.h), INT_MAX: no split
};

__device__ __forceinline__ float det2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }
// IEEE divide / sqrt are kept on purpose: with v_rcp_f32 / v_sqrt_f32 the kernel ran 1.3x faster but left the
// 1e-5 band around the C restatement (the LP's branch decisions amplify ulp-level differences).
//
// ieee_div / ieee_sqrt: the correctly rounded quotient / root by the very FMA sequences hipcc emits for `a / b` and
// sqrtf(x) on gfx950 -- v_rcp_f32, one Newton step on the reciprocal, the quotient refined twice (the last refinement is
// what v_div_fmas_f32 does); v_sqrt_f32 corrected by the two one-ulp residual tests -- WITHOUT the range handling around
// them (v_div_scale_f32 x2 + v_div_fixup_f32; the 2^32 pre-scale of sqrt), which only acts on operands whose exponents lie
// outside [-96, 96] or so: the linear programmes work on velocities, unit directions and their determinants (|x| in
// 1e-30 .. 1e8; quotients of parallel lines, the one place a denominator can vanish, are discarded by a select).  8 VALU
// instructions instead of 11, 9 instead of 13.  tests/test_gpu_orca.py::test_ieee_div_sqrt_sequences_are_correctly_rounded
// compares both with the compiler's operators on 2^31 operand pairs drawn from that range, bit for bit.
__device__ __forceinline__ float ieee_div(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    const float e0 = fmaf(-b, r, 1.0f);
    r = fmaf(e0, r, r);
    float q = a * r;
    const float e1 = fmaf(-b, q, a);
    q = fmaf(e1, r, q);
    const float e2 = fmaf(-b, q, a);
    return fmaf(e2, r, q);
}
__device__ __forceinline__ float ieee_sqrt(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-dn, s, x);
    const float t = (rd <= 0.0f) ? dn : s;      // (0 >= residual): the root was rounded up, step down
    const float ru = fmaf(-up, s, x);
    const float res = (ru > 0.0f) ? up : t;     // the root was rounded down, step up
    return (x == 0.0f) ? x : res;               // +-0 stay (s - 1 ulp of 0 is not a number to step to)
}

// ---- the arithmetic the register-resident build is instantiated with (template parameter FM of everything below) -------------
// FM = 0: "exact": IEEE divide / square root, no FMA contraction -- the restatement's operations on the restatement's operands,
//         bit-identical to oracle/orca_oracle.c (the bit-identity reference; cs_worlds.orca_math = CS_ORCA_MATH_EXACT, the default).
// FM = 1: "fast": v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 (1 ulp each) instead of the 8- and 9-instruction correctly rounded sequences.
// FM = 2: "fast + fma": and the 2 x 2 determinants, dot products and point + t * direction as mul + fma (one rounding less).
// north_star asks for 1e-5 on positions / velocities per step, not for bits, and the restatement cannot be pinned on rvo2 anyway;
// tests/test_gpu_orca_fast.py measures the fast builds per substep from re-synchronised state against the exact restatement and
// accounts for every agent-substep beyond 1e-5.
template <int FM> __device__ __forceinline__ float odiv(float a, float b)
{
    if constexpr (FM != 0) return a * __builtin_amdgcn_rcpf(b); else return ieee_div(a, b);
}
template <int FM> __device__ __forceinline__ float orcp(float b)
{
    if constexpr (FM != 0) return __builtin_amdgcn_rcpf(b); else return ieee_div(1.0f, b);
}
template <int FM> __device__ __forceinline__ float osqrt(float x)
{
    if constexpr (FM != 0) return __builtin_amdgcn_sqrtf(x); else return ieee_sqrt(x);
}
// 1 / sqrt(x): RVO2 normalises a vector by the reciprocal of its length (Vector2 / float)
template <int FM> __device__ __forceinline__ float orsqrt(float x)
{
    if constexpr (FM != 0) return __builtin_amdgcn_rsqf(x); else return ieee_div(1.0f, ieee_sqrt(x));
}
// a.x * b.y - a.y * b.x
template <int FM> __device__ __forceinline__ float odet(float ax, float ay, float bx, float by)
{
    if constexpr (FM >= 2) return fmaf(ax, by, -(ay * bx)); else return ax * by - ay * bx;
}
// a.x * b.x + a.y * b.y
template <int FM> __device__ __forceinline__ float odot(float ax, float ay, float bx, float by)
{
    if constexpr (FM >= 2) return fmaf(ax, bx, ay * by); else return ax * bx + ay * by;
}
// p + t * d
template <int FM> __device__ __forceinline__ float omad(float t, float d, float p)
{
    if constexpr (FM >= 2) return fmaf(t, d, p); else return p + t * d;
}

// per-lane column views into LDS: element i of lane tid lives at base[i * T + tid]
struct Lines {
    float4* p;
    int T, tid;
    __device__ __forceinline__ float4 get(int i) const { return p[i * T + tid]; }
    __device__ __forceinline__ void set(int i, float4 v) const { p[i * T + tid] = v; }
};

// RVO2 linearProgram1
__device__ bool lp1(const Lines& L, int lineNo, float radius, float ox, float oy, bool dirOpt, float& rx, float& ry)
{
    const float4 ln = L.get(lineNo); // x,y = point ; z,w = direction
    const float dot = ln.x * ln.z + ln.y * ln.w;
    const float disc = dot * dot + radius * radius - (ln.x * ln.x + ln.y * ln.y);
    if (disc < 0.0f) return false;
    const float sq = sqrtf(disc);
    float tL = -dot - sq, tR = -dot + sq;
    for (int i = 0; i < lineNo; ++i) {
        const float4 li = L.get(i);
        const float den = det2(ln.z, ln.w, li.z, li.w);
        const float num = det2(li.z, li.w, ln.x - li.x, ln.y - li.y);
        if (fabsf(den) <= RVO_EPSILON) {
            if (num < 0.0f) return false;
            continue;
        }
        const float t = num / den;
        if (den >= 0.0f) tR = fminf(tR, t); else tL = fmaxf(tL, t);
        if (tL > tR) return false;
    }
    float t;
    if (dirOpt) {
        t = (ox * ln.z + oy * ln.w > 0.0f) ? tR : tL;
    } else {
        t = ln.z * (ox - ln.x) + ln.w * (oy - ln.y);
        if (t < tL) t = tL; else if (t > tR) t = tR;
    }
    rx = ln.x + t * ln.z;
    ry = ln.y + t * ln.w;
    return true;
}

// RVO2 linearProgram2
__device__ int lp2(const Lines& L, int nl, float radius, float ox, float oy, bool dirOpt, float& rx, float& ry)
{
    if (dirOpt) { rx = ox * radius; ry = oy * radius; }
    else if (ox * ox + oy * oy > radius * radius) {
        const float nrm = sqrtf(ox * ox + oy * oy);
        const float inv = 1.0f / nrm;             // RVO2's Vector2 / float multiplies by the reciprocal (Vector2.h)
        rx = ox * inv * radius; ry = oy * inv * radius;
    } else { rx = ox; ry = oy; }
    for (int i = 0; i < nl; ++i) {
        const float4 li = L.get(i);
        if (det2(li.z, li.w, li.x - rx, li.y - ry) > 0.0f) {
            const float tx = rx, ty = ry;
            if (!lp1(L, i, radius, ox, oy, dirOpt, rx, ry)) { rx = tx; ry = ty; return i; }
        }
    }
    return nl;
}

// RVO2 linearProgram3: the first numObst lines (static obstacles) are hard constraints, copied unprojected
__device__ void lp3(const Lines& L, const Lines& P, int nl, int numObst, int begin, float radius, float& rx, float& ry)
{
    float distance = 0.0f;
    for (int i = begin; i < nl; ++i) {
        const float4 li = L.get(i);
        if (det2(li.z, li.w, li.x - rx, li.y - ry) > distance) {
            int np = 0;
            for (int j = 0; j < numObst; ++j) P.set(np++, L.get(j));
            for (int j = numObst; j < i; ++j) {
                const float4 lj = L.get(j);
                float4 ln;
                const float d = det2(li.z, li.w, lj.z, lj.w);
                if (fabsf(d) <= RVO_EPSILON) {
                    if (li.z * lj.z + li.w * lj.w > 0.0f) continue;
                    ln.x = 0.5f * (li.x + lj.x); ln.y = 0.5f * (li.y + lj.y);
                } else {
                    const float s = det2(lj.z, lj.w, li.x - lj.x, li.y - lj.y) / d;
                    ln.x = li.x + s * li.z; ln.y = li.y + s * li.w;
                }
                const float ex = lj.z - li.z, ey = lj.w - li.w;
                const float en = sqrtf(ex * ex + ey * ey);
                const float inv = 1.0f / en;
                ln.z = ex * inv; ln.w = ey * inv;
                P.set(np++, ln);
            }
            const float tx = rx, ty = ry;
            if (lp2(P, np, radius, -li.w, li.z, true, rx, ry) < np) { rx = tx; ry = ty; }
            distance = det2(li.z, li.w, li.x - rx, li.y - ry);
        }
    }
}


// ---- static obstacles: RVO2 Agent::computeNeighbors (obstacle part), insertObstacleNeighbor and the "Create obstacle
// ORCA lines" block of Agent::computeNewVelocity, op for op as oracle/orca_oracle.c restates them ------------------------
struct Vtx { float px, py, ux, uy, convex; int next, prev; };
__device__ __forceinline__ Vtx load_vtx(const float* V, int i)
{
    const float4 a = *reinterpret_cast<const float4*>(V + (long)i * 8);
    const float4 b = *reinterpret_cast<const float4*>(V + (long)i * 8 + 4);
    return Vtx{a.x, a.y, a.z, a.w, b.x, (int)b.y, (int)b.z};
}
__device__ __forceinline__ float absSq2(float x, float y) { return x * x + y * y; }
__device__ __forceinline__ float dist_sq_point_segment(float ax, float ay, float bx, float by, float cx, float cy)
{
    const float r = ((cx - ax) * (bx - ax) + (cy - ay) * (by - ay)) / absSq2(bx - ax, by - ay);
    if (r < 0.0f) return absSq2(cx - ax, cy - ay);
    if (r > 1.0f) return absSq2(cx - bx, cy - by);
    return absSq2(cx - (ax + r * (bx - ax)), cy - (ay + r * (by - ay)));
}

// nearest obstacle edges the agent is on the right of, sorted by distance, in the per-lane LDS columns od / oi
__device__ int obstacle_neighbors(const float* V, int nv, int KO, float px, float py, float rangeSq, float* od, int* oi, int T, int tid)
{
    int cnt = 0;
    for (int v = 0; v < nv; ++v) {
        const Vtx o1 = load_vtx(V, v);
        const Vtx o2 = load_vtx(V, o1.next);
        const float agentLeftOfLine = det2(o1.px - px, o1.py - py, o2.px - o1.px, o2.py - o1.py);
        const float distSqLine = agentLeftOfLine * agentLeftOfLine / absSq2(o2.px - o1.px, o2.py - o1.py);
        if (distSqLine < rangeSq && agentLeftOfLine < 0.0f) {
            const float distSq = dist_sq_point_segment(o1.px, o1.py, o2.px, o2.py, px, py);
            if (distSq < rangeSq) {
                int i;
                if (cnt < KO) i = cnt++;
                else if (distSq < od[(KO - 1) * T + tid]) i = KO - 1; // full: the farthest edge drops out
                else continue;
                while (i != 0 && distSq < od[(i - 1) * T + tid]) {
                    od[i * T + tid] = od[(i - 1) * T + tid];
                    oi[i * T + tid] = oi[(i - 1) * T + tid];
                    --i;
                }
                od[i * T + tid] = distSq;
                oi[i * T + tid] = v;
            }
        }
    }
    return cnt;
}

__device__ int obstacle_lines(const float* V, const int* oi, int no, int T, int tid, float px, float py, float vx, float vy,
                              float radius, float invT, const Lines& L)
{
    int nl = 0;
    for (int k = 0; k < no; ++k) {
        const Vtx o1 = load_vtx(V, oi[k * T + tid]);
        const Vtx o2 = load_vtx(V, o1.next);
        const float r1x = o1.px - px, r1y = o1.py - py, r2x = o2.px - px, r2y = o2.py - py;
        bool covered = false;
        for (int j = 0; j < nl; ++j) {
            const float4 lj = L.get(j);
            if (det2(invT * r1x - lj.x, invT * r1y - lj.y, lj.z, lj.w) - invT * radius >= -RVO_EPSILON &&
                det2(invT * r2x - lj.x, invT * r2y - lj.y, lj.z, lj.w) - invT * radius >= -RVO_EPSILON) {
                covered = true;
                break;
            }
        }
        if (covered) continue;
        const float distSq1 = absSq2(r1x, r1y), distSq2 = absSq2(r2x, r2y), radiusSq = radius * radius;
        const float ovx = o2.px - o1.px, ovy = o2.py - o1.py;
        const float s = (-r1x * ovx + -r1y * ovy) / absSq2(ovx, ovy);
        const float distSqLine = absSq2(-r1x - s * ovx, -r1y - s * ovy);
        if (s < 0.0f && distSq1 <= radiusSq) {            // collision with left vertex; ignore if non-convex
            if (o1.convex != 0.0f) {
                const float inv = 1.0f / sqrtf(absSq2(-r1y, r1x));
                L.set(nl++, make_float4(0.0f, 0.0f, -r1y * inv, r1x * inv));
            }
            continue;
        } else if (s > 1.0f && distSq2 <= radiusSq) {     // collision with right vertex
            if (o2.convex != 0.0f && det2(r2x, r2y, o2.ux, o2.uy) >= 0.0f) {
                const float inv = 1.0f / sqrtf(absSq2(-r2y, r2x));
                L.set(nl++, make_float4(0.0f, 0.0f, -r2y * inv, r2x * inv));
            }
            continue;
        } else if (s >= 0.0f && s < 1.0f && distSqLine <= radiusSq) { // collision with the segment
            L.set(nl++, make_float4(0.0f, 0.0f, -o1.ux, -o1.uy));
            continue;
        }
        float llx, lly, rlx, rly;
        Vtx a1 = o1, a2 = o2;
        bool same = false;
        if (s < 0.0f && distSqLine <= radiusSq) {         // obliquely viewed: the left vertex defines the velocity obstacle
            if (o1.convex == 0.0f) continue;
            a2 = o1; same = true;
            const float leg1 = sqrtf(distSq1 - radiusSq);
            const float inv1 = 1.0f / distSq1;
            llx = (r1x * leg1 - r1y * radius) * inv1; lly = (r1x * radius + r1y * leg1) * inv1;
            rlx = (r1x * leg1 + r1y * radius) * inv1; rly = (-r1x * radius + r1y * leg1) * inv1;
        } else if (s > 1.0f && distSqLine <= radiusSq) {  // the right vertex defines it
            if (o2.convex == 0.0f) continue;
            a1 = o2; same = true;
            const float leg2 = sqrtf(distSq2 - radiusSq);
            const float inv2 = 1.0f / distSq2;
            llx = (r2x * leg2 - r2y * radius) * inv2; lly = (r2x * radius + r2y * leg2) * inv2;
            rlx = (r2x * leg2 + r2y * radius) * inv2; rly = (-r2x * radius + r2y * leg2) * inv2;
        } else {                                          // usual situation
            if (o1.convex != 0.0f) {
                const float leg1 = sqrtf(distSq1 - radiusSq);
                const float inv1 = 1.0f / distSq1;
                llx = (r1x * leg1 - r1y * radius) * inv1; lly = (r1x * radius + r1y * leg1) * inv1;
            } else { llx = -o1.ux; lly = -o1.uy; }
            if (o2.convex != 0.0f) {
                const float leg2 = sqrtf(distSq2 - radiusSq);
                const float inv2 = 1.0f / distSq2;
                rlx = (r2x * leg2 + r2y * radius) * inv2; rly = (-r2x * radius + r2y * leg2) * inv2;
            } else { rlx = o1.ux; rly = o1.uy; }
        }
        const Vtx leftNb = load_vtx(V, a1.prev);
        bool leftForeign = false, rightForeign = false;
        if (a1.convex != 0.0f && det2(llx, lly, -leftNb.ux, -leftNb.uy) >= 0.0f) { llx = -leftNb.ux; lly = -leftNb.uy; leftForeign = true; }
        if (a2.convex != 0.0f && det2(rlx, rly, a2.ux, a2.uy) <= 0.0f) { rlx = a2.ux; rly = a2.uy; rightForeign = true; }
        const float lcx = invT * (a1.px - px), lcy = invT * (a1.py - py);
        const float rcx = invT * (a2.px - px), rcy = invT * (a2.py - py);
        const float cvx = rcx - lcx, cvy = rcy - lcy;
        const float t = same ? 0.5f : ((vx - lcx) * cvx + (vy - lcy) * cvy) / absSq2(cvx, cvy);
        const float tL = (vx - lcx) * llx + (vy - lcy) * lly;
        const float tR = (vx - rcx) * rlx + (vy - rcy) * rly;
        if ((t < 0.0f && tL < 0.0f) || (same && tL < 0.0f && tR < 0.0f)) { // left cut-off circle
            const float wx = vx - lcx, wy = vy - lcy, wn = sqrtf(absSq2(wx, wy));
            const float inv = 1.0f / wn;
            const float ux = wx * inv, uy = wy * inv;
            L.set(nl++, make_float4(lcx + radius * invT * ux, lcy + radius * invT * uy, uy, -ux));
            continue;
        } else if (t > 1.0f && tR < 0.0f) {                                  // right cut-off circle
            const float wx = vx - rcx, wy = vy - rcy, wn = sqrtf(absSq2(wx, wy));
            const float inv = 1.0f / wn;
            const float ux = wx * inv, uy = wy * inv;
            L.set(nl++, make_float4(rcx + radius * invT * ux, rcy + radius * invT * uy, uy, -ux));
            continue;
        }
        const float dCut = (t < 0.0f || t > 1.0f || same) ? INFINITY : absSq2(vx - (lcx + t * cvx), vy - (lcy + t * cvy));
        const float dLeft = (tL < 0.0f) ? INFINITY : absSq2(vx - (lcx + tL * llx), vy - (lcy + tL * lly));
        const float dRight = (tR < 0.0f) ? INFINITY : absSq2(vx - (rcx + tR * rlx), vy - (rcy + tR * rly));
        float dx, dy, bx, by;
        if (dCut <= dLeft && dCut <= dRight) { dx = -a1.ux; dy = -a1.uy; bx = lcx; by = lcy; }            // cut-off line
        else if (dLeft <= dRight) { if (leftForeign) continue; dx = llx; dy = lly; bx = lcx; by = lcy; }   // left leg
        else { if (rightForeign) continue; dx = -rlx; dy = -rly; bx = rcx; by = rcy; }                     // right leg
        L.set(nl++, make_float4(bx + radius * invT * -dy, by + radius * invT * dx, dx, dy));
    }
    return nl;
}

// One ORCA half-plane (RVO2 Agent::computeNewVelocity, agent part): q = (x, y, vx, vy) of the neighbour,
// R = combined radius.  Returns (point.x, point.y, direction.x, direction.y).
// One agent-agent ORCA half-plane (RVO2 Agent::computeNewVelocity).  RVO2's three cases -- projection on the cut-off circle,
// on a leg, and the collision case (cut-off circle of the time step) -- each take one square root and one reciprocal of
// different arguments; the arguments are selected first, so a lane pays one IEEE sqrt and one IEEE divide whatever cases the
// wavefront mixes, with the very same operations on the very same operands as the case it is in.  invDt = 1 / timeStep.
// (FM: the decisions -- collision, cut-off circle or leg, which leg -- are taken on the same uncontracted quantities in every build;
//  only the root and the reciprocal differ.)
template <int FM = 0>
__device__ __forceinline__ float4 orca_line(float px, float py, float vx, float vy, const float4 q, float R, float invT, float invDt)
{
    const float rpx = q.x - px, rpy = q.y - py;
    const float rvx = vx - q.z, rvy = vy - q.w;
    const float distSq = rpx * rpx + rpy * rpy;
    const float RSq = R * R;
    const bool coll = !(distSq > RSq);
    const float kT = coll ? invDt : invT;
    const float wx = rvx - kT * rpx, wy = rvy - kT * rpy;
    const float wLenSq = wx * wx + wy * wy;
    const float dot1 = wx * rpx + wy * rpy;
    const bool circ = coll | ((dot1 < 0.0f) & (dot1 * dot1 > RSq * wLenSq));   // (no short circuit: three compares, no branch per line)
    const float root = osqrt<FM>(circ ? wLenSq : distSq - RSq);   // |w|  or  the leg length
    const float den = circ ? root : distSq;
    float inv;
    if constexpr (FM != 0) inv = __builtin_amdgcn_rcpf(den);           // (v_rcp_f32(+0) = +inf)
    else inv = (den == 0.0f) ? INFINITY : ieee_div(1.0f, den);          // (w == 0 exactly: 1 / +0 as the operator gives it)
    // cut-off circle (of the time horizon, or of the time step when the discs overlap)
    const float uwx = wx * inv, uwy = wy * inv;
    const float sc = R * kT - root;
    // legs
    const bool left = det2(rpx, rpy, wx, wy) > 0.0f;
    // RVO2: left leg (rp.x * leg - rp.y * R, rp.x * R + rp.y * leg) / distSq, right leg -(rp.x * leg + rp.y * R, -rp.x * R + rp.y * leg) / distSq:
    // the right leg is the left one with -leg for leg, bit for bit (negation commutes with every rounding here)
    const float sroot = left ? root : -root;
    const float lx = (rpx * sroot - rpy * R) * inv;
    const float ly = (rpx * R + rpy * sroot) * inv;
    const float dot2 = rvx * lx + rvy * ly;
    const float dx = circ ? uwy : lx, dy = circ ? -uwx : ly;
    const float ux = circ ? sc * uwx : dot2 * lx - rvx, uy = circ ? sc * uwy : dot2 * ly - rvy;
    return make_float4(vx + 0.5f * ux, vy + 0.5f * uy, dx, dy);
}

// linearProgram3 on ten register-resident lines (no static obstacles): the same operations in the same order as lp3() /
// lp2() / lp1() above.  The walk over the violated lines i is a wave-uniform runtime loop (line i is read from an LDS copy); the
// projected lines j < i and linearProgram2 / linearProgram1 over them are statically unrolled and guarded by the uniform j < i.
// RVO2 drops a projected line whose source line is parallel to line i and points the same way: here it keeps its slot and a
// cleared bit in `pvalid`, and every loop skips it, which visits the surviving lines in the same order.
template <int FM>
__device__ __forceinline__ void lp3_fast10(const float4 (&Lr)[10], const Lines& L, int cnt, int failed, float vmax, float& rx, float& ry)
{
    float distance = 0.0f;
#pragma nounroll
    for (int i = 0; i < 10; ++i) {
        const float4 li = L.get(i);   // line i by its run-time index: from the LDS copy (a register array indexed at run time goes to scratch memory, ~10x the latency)
        const bool act = (i >= failed) && (i < cnt) && (odet<FM>(li.z, li.w, li.x - rx, li.y - ry) > distance);
        if (__builtin_amdgcn_ballot_w64(act) == 0) continue;
        float4 Pr[9];
        unsigned pvalid = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j < i) {
                const float4 lj = Lr[j];
                const float d = odet<FM>(li.z, li.w, lj.z, lj.w);
                const bool par = fabsf(d) <= RVO_EPSILON;
                const bool same = odot<FM>(li.z, li.w, lj.z, lj.w) > 0.0f;
                const float s = odiv<FM>(odet<FM>(lj.z, lj.w, li.x - lj.x, li.y - lj.y), d);
                float4 ln;
                ln.x = par ? 0.5f * (li.x + lj.x) : omad<FM>(s, li.z, li.x);
                ln.y = par ? 0.5f * (li.y + lj.y) : omad<FM>(s, li.w, li.y);
                const float ex = lj.z - li.z, ey = lj.w - li.w;
                const float inv = orsqrt<FM>(odot<FM>(ex, ey, ex, ey));
                ln.z = ex * inv; ln.w = ey * inv;
                Pr[j] = ln;
                pvalid |= (par && same) ? 0u : (1u << j);
            }
        }
        // linearProgram2(projLines, radius, (-dir.y, dir.x), directionOpt = true)
        const float ox = -li.w, oy = li.z;
        float qx = ox * vmax, qy = oy * vmax;
        bool fail2 = false;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k < i) {
                const float4 lk = Pr[k];
                const bool viol = act && !fail2 && ((pvalid >> k) & 1u) && (odet<FM>(lk.z, lk.w, lk.x - qx, lk.y - qy) > 0.0f);
                if (__builtin_amdgcn_ballot_w64(viol) != 0) { // linearProgram1(projLines, k, ...)
                    const float dot = odot<FM>(lk.x, lk.y, lk.z, lk.w);
                    const float disc = dot * dot + vmax * vmax - odot<FM>(lk.x, lk.y, lk.x, lk.y);
                    bool ok = !(disc < 0.0f);
                    const float sq = osqrt<FM>(fmaxf(disc, 0.0f));
                    float tL = -dot - sq, tR = -dot + sq;
#pragma unroll
                    for (int m = 0; m < k; ++m) {   // branch-free: the divisions of all m are independent and overlap
                        const float4 lm = Pr[m];
                        const float den = odet<FM>(lk.z, lk.w, lm.z, lm.w);
                        const float num = odet<FM>(lm.z, lm.w, lk.x - lm.x, lk.y - lm.y);
                        const float t = odiv<FM>(num, den);
                        const bool live = ((pvalid >> m) & 1u) != 0;
                        const bool par = fabsf(den) <= RVO_EPSILON;
                        const bool upd = live && !par && ok;             // (after a failure RVO2 has already returned)
                        const float nR = (den >= 0.0f) ? fminf(tR, t) : tR, nL = (den >= 0.0f) ? tL : fmaxf(tL, t);
                        tR = upd ? nR : tR; tL = upd ? nL : tL;
                        ok = ok && !(live && par && num < 0.0f) && !(upd && tL > tR);
                    }
                    const float t = (odot<FM>(ox, oy, lk.z, lk.w) > 0.0f) ? tR : tL;
                    const bool set = viol && ok;
                    qx = set ? omad<FM>(t, lk.z, lk.x) : qx;
                    qy = set ? omad<FM>(t, lk.w, lk.y) : qy;
                    fail2 = fail2 || (viol && !ok);
                }
            }
        }
        const bool take = act && !fail2;           // a failed linearProgram2 leaves the result where it was
        rx = take ? qx : rx;
        ry = take ? qy : ry;
        distance = act ? odet<FM>(li.z, li.w, li.x - rx, li.y - ry) : distance;
    }
}

// ---- linearProgram3, one 16-lane ROW per (agent, violated line) ----------------------------------------------------------
// Measured on the restatement in the dense phase of the cfg4 crossing: 35 % of the agents have an infeasible programme in a
// substep, such an agent walks 1.2 of the ten LP3 levels, projects 7 lines and calls linearProgram1 3.6 times with 7 inner
// iterations in all -- but a wavefront of 50 agents executes the UNION of its lanes' paths: 7.5 levels, each with every
// projection and nearly every linearProgram1 of its triangle, for 2.7 active lanes on average (lp3_fast10 above: 6400 of the
// 9700 vector instructions of a wavefront-substep).  Here the lanes' roles are re-dealt for LP3 only: every agent with a
// pending level gets a row of 16 lanes (four agents per pass, passes until the wavefront's pending agents are served, rounds
// until no agent has a further violated line).  Inside a row lane j owns line j: the i projections of a level are ONE step,
// linearProgram2 over them finds the first violated line with a ballot, and linearProgram1's loop over the earlier lines --
// a min / max of quotients with an any-fail flag, exact and order-independent -- is one step plus a DPP row reduction.  Every
// operation on the path RVO2 takes is the same operation on the same operands (the early exits of linearProgram1 become the
// final tL > tR test: tL only grows, tR only shrinks), so the results stay bit-identical to oracle/orca_oracle.c.
// Rows talk through LDS inside ONE wavefront (in-order LDS, compiler-only fences): the agents' lines L[10][TL] (stored by the
// owners), the projected lines P[9][TL], one (result, maxSpeed, distance) record and one ticket per agent.
#define ORCA_LDS_FENCE() asm volatile("" ::: "memory")
typedef unsigned long long mask64;
// LLVM's compare predicates, as __builtin_amdgcn_fcmpf / sicmp / uicmp take them (llvm/IR/InstrTypes.h CmpInst::Predicate)
enum { FCMP_OGT = 2, FCMP_OGE = 3, FCMP_OLT = 4, FCMP_OLE = 5, ICMP_NE = 33, ICMP_SGT = 38, ICMP_SLT = 40 };
// P [9][8] projected lines of the eight rows in flight, A [9][8] their chords on the speed circle (tL0, tR0; (+inf, -inf) when
// the line misses the circle, which fails linearProgram1 through its own tL > tR test), q [T] agent records, sel [T] tickets
struct RowLds { float4* P; float2* A; float4* q; int* sel; };

// min and max over the 16-lane row of the calling lane, left in every lane of the row: four rotate-and-combine steps
// (row_ror 8, 4, 2, 1) with the DPP operand fused into v_min_f32 / v_max_f32 -- the two chains alternate, one wait state
// covers the VALU-write -> DPP-read hazard.  The operands are finite or +-inf, never NaN.
// then the interval of linearProgram1: tR <- min(tR, row min), tL <- max(tL, row max) -- inside the block, as the two instructions
// themselves (fminf / fmaxf on values the compiler cannot see through would each be preceded by a quieting v_max x, x, x)
__device__ __forceinline__ void row_minmax16(float& mn, float& mx, float& tR, float& tL)
{
    asm volatile("s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32 %2, %2, %0\n\t"
                 "v_max_f32 %3, %3, %1"
                 : "+v"(mn), "+v"(mx), "+v"(tR), "+v"(tL));
}

// min and max over the 8-lane half row of the calling lane, left in every lane of it: neighbour swap, quad swap, half-row mirror
// (quad_perm:[1,0,3,2], quad_perm:[2,3,0,1], row_half_mirror) -- three steps instead of four.
__device__ __forceinline__ void row_minmax8(float& mn, float& mx, float& tR, float& tL)
{
    asm volatile("s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32 %2, %2, %0\n\t"
                 "v_max_f32 %3, %3, %1"
                 : "+v"(mn), "+v"(mx), "+v"(tR), "+v"(tL));
}

// One sub-phase of a round of lp3_rows: the `npend` agents ticketed in R.sel (their records in R.q) are served 64 / RW at a time,
// each by a group of RW lanes (RW = 16: a DPP row, any level; RW = 8: half a row, levels 0 .. 8 -- lane j of a group owns line j and
// a level i needs the lines j < i).  All 64 lanes call it.
// (Measured and rejected: a group that keeps its agent for ALL its violated lines instead of one per round -- the wavefront then waits
//  for the longest chain of the eight agents of a pass while the other groups idle: 512 -> 548 us in the dense phase of cfg4; re-dealing
//  the agents that still have a violated line every round packs the groups densely.)
template <int RW, int FM>
__device__ __forceinline__ void lp3_serve(const Lines& L, const RowLds& R, int npend)
{
    constexpr int GROUPS = 64 / RW;
    const int TL = L.T;
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    const int grp_sh = lane & (64 - RW), j = lane & (RW - 1), jj = j < 9 ? j : 8, slot = lane / RW;
#pragma nounroll
    for (int p0 = 0; p0 < npend; p0 += GROUPS) {
        const int idx = p0 + slot;
        const bool rowvalid = idx < npend;
        const int e = rowvalid ? R.sel[wbase + idx] : 0;
        const int a = e & 0xFFFF, lv = e >> 16;
        const float4 qa = R.q[a];
        const float4 li = L.p[__umul24(lv, TL) + a];              // (24-bit multiplies: full rate, v_mul_lo_u32 is not)
        const float4 lj = L.p[__umul24(jj, TL) + a];
        const bool task = rowvalid && j < lv;
        // RVO2 linearProgram3: line j projected on line i
        const float d = odet<FM>(li.z, li.w, lj.z, lj.w);
        const bool par = fabsf(d) <= RVO_EPSILON;
        const bool same = odot<FM>(li.z, li.w, lj.z, lj.w) > 0.0f;
        const float sp = odiv<FM>(odet<FM>(lj.z, lj.w, li.x - lj.x, li.y - lj.y), d);
        float4 pr;
        pr.x = par ? 0.5f * (li.x + lj.x) : omad<FM>(sp, li.z, li.x);
        pr.y = par ? 0.5f * (li.y + lj.y) : omad<FM>(sp, li.w, li.y);
        const float ex = lj.z - li.z, ey = lj.w - li.w;
        const float inv = orsqrt<FM>(odot<FM>(ex, ey, ex, ey));
        pr.z = ex * inv; pr.w = ey * inv;
        const bool live = task && !(par && same);   // RVO2 drops a parallel line that points the same way
        // what linearProgram1 computes from this line alone, should it become the violated one: the circle's chord
        const float vm = qa.z;
        const float dot = odot<FM>(pr.x, pr.y, pr.z, pr.w);
        const float disc = dot * dot + vm * vm - odot<FM>(pr.x, pr.y, pr.x, pr.y);
        const float sq = osqrt<FM>(fmaxf(disc, 0.0f));
        const float2 aux = (disc < 0.0f) ? make_float2(INFINITY, -INFINITY) : make_float2(-dot - sq, -dot + sq);
        // (half rows: lane j's slot is its own and slots j >= lv are never read -- every lane stores, nothing to branch around; whole rows
        //  share slot 8 among the lanes j >= 8)
        if (RW == 8 || task) { R.P[jj * 8 + slot] = pr; R.A[jj * 8 + slot] = aux; }
        // linearProgram2(projLines, radius, (-dir.y, dir.x), directionOpt = true) starts on the circle
        const float ox = -li.w, oy = li.z;
        float qx = ox * vm, qy = oy * vm;
        int k = -1;
        // The walk's lane conditions are kept as wave masks (one bit per lane, scalar registers): every compare below writes its mask
        // directly (v_cmp -> SGPR pair), the logic between them is scalar, and a mask goes back into a select as it is
        // (inverse ballot).  Written on bools the vote was `v_cndmask 0 / 1, v_cmp_ne` on top of the compares, and the interval's
        // fminf / fmaxf quieted their operands first (they come from LDS and from the DPP block: four v_max x, x, x per trip).
        const mask64 live_m = __builtin_amdgcn_ballot_w64(live);
        mask64 fail_m = 0, done_m = ~__builtin_amdgcn_ballot_w64(rowvalid);
        ORCA_LDS_FENCE();
#pragma nounroll
        for (int it = 0; it < 9; ++it) {
            const mask64 vmask = __builtin_amdgcn_fcmpf(odet<FM>(pr.z, pr.w, pr.x - qx, pr.y - qy), 0.0f, FCMP_OGT) & __builtin_amdgcn_sicmp(j, k, ICMP_SGT) & live_m & ~done_m;
            if (vmask == 0) break;
            const unsigned rb = (unsigned)(vmask >> grp_sh) & ((1u << RW) - 1u);
            const mask64 any_m = __builtin_amdgcn_uicmp(rb, 0u, ICMP_NE);
            // first violated line of my group (a group without one is done -- linearProgram2 succeeded -- and reads some valid slot)
            int kk;
            if constexpr (RW == 8) kk = (int)__builtin_ctz(rb | 0x100u) & 7; else kk = __builtin_amdgcn_inverse_ballot_w64(any_m) ? (int)__builtin_ctz(rb) : 0;
            const float4 lk = R.P[kk * 8 + slot];
            const float2 ak = R.A[kk * 8 + slot];
            // linearProgram1(projLines, kk, ...): lane j < kk evaluates line j against line kk
            const mask64 in_m = live_m & __builtin_amdgcn_sicmp(j, kk, ICMP_SLT);
            const float den = odet<FM>(lk.z, lk.w, pr.z, pr.w);
            const float num = odet<FM>(pr.z, pr.w, lk.x - pr.x, lk.y - pr.y);
            float t = odiv<FM>(num, den);
            asm("" : "+v"(t));                                   // (evaluated by every lane: the compiler would branch around the reciprocal of the lanes that select it)
            const mask64 parl_m = __builtin_amdgcn_fcmpf(fabsf(den), RVO_EPSILON, FCMP_OLE);
            // a parallel earlier line with this line on its wrong side fails linearProgram1 outright: it enters the
            // reduction as the empty interval (tR = -inf, tL = +inf), which the tL > tR test below turns into the failure
            const mask64 failp_m = in_m & parl_m & __builtin_amdgcn_fcmpf(num, 0.0f, FCMP_OLT);
            const mask64 ge_m = __builtin_amdgcn_fcmpf(den, 0.0f, FCMP_OGE);
            const bool failp = __builtin_amdgcn_inverse_ballot_w64(failp_m);
            float rmin = failp ? -INFINITY : (__builtin_amdgcn_inverse_ballot_w64(in_m & ~parl_m & ge_m) ? t : INFINITY);
            float rmax = failp ? INFINITY : (__builtin_amdgcn_inverse_ballot_w64(in_m & ~parl_m & ~ge_m) ? t : -INFINITY);
            float tR = ak.y, tL = ak.x;
            if constexpr (RW == 16) row_minmax16(rmin, rmax, tR, tL); else row_minmax8(rmin, rmax, tR, tL);   // tR = min(chord, row's min), tL = max(chord, row's max)
            const mask64 ok_m = ~__builtin_amdgcn_fcmpf(tL, tR, FCMP_OGT);
            const float tt = (odot<FM>(ox, oy, lk.z, lk.w) > 0.0f) ? tR : tL;
            const mask64 set_m = any_m & ok_m, bad_m = any_m & ~ok_m;   // (a group with a violated line is not done: any implies !done)
            const bool set = __builtin_amdgcn_inverse_ballot_w64(set_m);
            qx = set ? omad<FM>(tt, lk.z, lk.x) : qx;
            qy = set ? omad<FM>(tt, lk.w, lk.y) : qy;
            k = set ? kk : k;
            fail_m |= bad_m;                                    // linearProgram2 failed: LP3 keeps the old result
            done_m |= ~any_m | bad_m;                           // no line violated: linearProgram2 succeeded
        }
        const bool fail2 = __builtin_amdgcn_inverse_ballot_w64(fail_m);
        const bool take = rowvalid && !fail2;
        const float nrx = take ? qx : qa.x, nry = take ? qy : qa.y;
        const float ndist = odet<FM>(li.z, li.w, li.x - nrx, li.y - nry);
        if (rowvalid && j == 0) R.q[a] = make_float4(nrx, nry, qa.z, ndist);
        ORCA_LDS_FENCE();
    }
}

// (Round 5, measured and rejected: the same sub-phase with linearProgram1 evaluated for EVERY line k of the level up front -- one batch of
//  LDS reads, a DPP reduction per k, then a lane-local walk with no vote or LDS access inside -- as linearProgram2 does below.  Bit-identical,
//  and slower: 471 vs 404 us in the dense phase of cfg4 (fma arithmetic).  A level needs 3.1 linearProgram1 calls; evaluating all 8 costs more
//  instructions than the serial chains cost in latency: the kernel is bound by instruction issue, not by those chains.  HISTORY.md.)
// Called by ALL 64 lanes of a wavefront (lanes without an agent pass cnt = failed = 0).  Lr: the caller's ten lines (also
// stored in L's column L.tid); on return (rx, ry) is linearProgram3's result for lanes with failed < cnt.
// A round serves every agent's next violated line: those at levels 0 .. 8 (the lines j < 8 fit half a row) eight agents per pass on
// 8-lane groups, those at level 9 (an agent with ten neighbours whose LAST line is violated: needs nine projected lines) four per
// pass on whole rows.  Round 3: with rows of 16 for everybody a pass served four agents and nine of its sixteen lanes at most.
template <int FM>
__device__ __forceinline__ void lp3_rows(const float4 (&Lr)[10], const Lines& L, const RowLds& R, int cnt, int failed, float vmax, float& rx, float& ry)
{
    const int me = L.tid;
    const int wbase = threadIdx.x & ~63;
    float distance = 0.0f;
    // the first violated line is the one linearProgram2 failed on: it tested det(dir, point - result) > 0 on this very result and
    // left the result where it was (RVO2 walks i upwards from there; distance starts at 0).  The scan for the NEXT violated line
    // closes a round, so a wavefront runs one scan per round it serves, not one more to find out that nothing is left.
    int lvl = failed < cnt ? failed : -1;
#pragma nounroll
    for (int round = 0; round < 10; ++round) {
        const bool pending = lvl >= 0;
        if (__builtin_amdgcn_ballot_w64(pending) == 0) break;
        R.q[me] = make_float4(rx, ry, vmax, distance);          // (every lane: a lane that is not pending reads its own record back below)
        // sub-phase A: levels 0 .. 8 on 8-lane groups; sub-phase B: level 9 on 16-lane rows
        const bool pa = pending && lvl <= 8, pb = pending && lvl == 9;
        const unsigned long long ma = __builtin_amdgcn_ballot_w64(pa), mb = __builtin_amdgcn_ballot_w64(pb);
        if (ma != 0) {
            if (pa) {
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0u));
                R.sel[wbase + rank] = me | (lvl << 16);
            }
            ORCA_LDS_FENCE();
            lp3_serve<8, FM>(L, R, __builtin_popcountll(ma));
        }
        if (mb != 0) {
            ORCA_LDS_FENCE();
            if (pb) {
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
                R.sel[wbase + rank] = me | (lvl << 16);
            }
            ORCA_LDS_FENCE();
            lp3_serve<16, FM>(L, R, __builtin_popcountll(mb));
        }
        const float4 qo = R.q[me];
        rx = qo.x; ry = qo.y; distance = qo.w;
        const int next_i = pending ? lvl + 1 : 10;
        ORCA_LDS_FENCE();
        // owner: my next violated line at or behind next_i, with the current result and distance
        lvl = -1;
#pragma unroll
        for (int i = 9; i >= 1; --i) {
            const float4 li = Lr[i];
            const bool v = (i >= next_i) && (i < cnt) && (odet<FM>(li.z, li.w, li.x - rx, li.y - ry) > distance);
            lvl = v ? i : lvl;
        }
    }
}

// Register-resident solve for maxNeighbors = 10 (ORCA_DEFAULTS): same arithmetic and the same order of
// operations as the generic path, organised for the SIMD:
//  * neighbours: the 10 smallest (distSq, row) pairs in lexicographic order -- what RVO2's insertion with
//    strict '<' produces -- kept sorted in ten 64-bit keys (high word = float bits of distSq, low word = row)
//    that order like positive doubles, so one insertion is ten v_min_f64 / v_max_f64 compare-exchanges;
//  * ORCA lines in registers; linearProgram2 / linearProgram1 statically unrolled over the (line, earlier
//    line) triangle, each line's LP1 skipped wave-uniformly when no lane violates that line;
//  * linearProgram3 (infeasible programme: about a third of the agents of a circular crossing, every substep) stays in
//    registers too (lp3_fast10).
// insertion of one candidate into the ten sorted (distSq, row) keys
// compare-exchange of two keys: a <- min, b <- max.  The keys are bit patterns assembled from (float bits, row), never NaN; written as
// fmin / fmax the compiler cannot know that and quiets every key first (v_max_f64 x, x, x: one more 4-cycle instruction per candidate).
__device__ __forceinline__ void key_ce(double& a, double& b)
{
    double lo, hi;
    asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo), "=v"(hi) : "v"(a), "v"(b));
    a = lo; b = hi;
}
__device__ __forceinline__ void key_insert10(double (&key)[10], double x)
{
#pragma unroll
    for (int s = 0; s < 10; ++s) key_ce(key[s], x);
}
__device__ __forceinline__ double key_sentinel() { return __hiloint2double(0x7F7FFFFF, (int)0xFFFFFFFFu); }
#define ORCA_CE(a, b) key_ce(a, b);

// From the ten neighbour keys on: ORCA lines, linearProgram2, linearProgram3.  fetch(b, q, rad): (x, y, vx, vy) and radius +
// margin of row b (LDS rows of the world in the crowd kernel, global rows found through the grid in the big-world kernel).
template <int FM, class Fetch>
__device__ __forceinline__ void orca_solve_fast10(bool active, bool lp3_static, const double (&key)[10], int row, Fetch&& fetch, float px, float py, float vx,
                                  float vy, float my_r, float vmax, float pvx, float pvy, float time_horizon, float dt, const Lines& L,
                                  const RowLds& R, float& nvx, float& nvy, unsigned long long* g_ost, unsigned long long& g_ost_last)
{
    constexpr int KF = 10;
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < KF; ++s) cnt += (__double2hiint(key[s]) != 0x7F7FFFFF) ? 1 : 0;
    OSTAMP(1);

    // (opaque: left visible, `coll ? 1 / dt : 1 / T` is rewritten as 1 / (coll ? dt : T) -- an IEEE division per LINE, twelve instructions
    //  each, where two per launch do; same bits either way)
    float invT = 1.0f / time_horizon;
    float invDt = 1.0f / dt;
    asm("" : "+v"(invT), "+v"(invDt));
    float4 Lr[KF];
#pragma unroll
    for (int k = 0; k < KF; ++k) {
        const int b = (k < cnt) ? __double2loint(key[k]) : row; // unused slots read my own row (finite, never used)
        float4 q;
        float rad;
        fetch(b, q, rad);
        Lr[k] = orca_line<FM>(px, py, vx, vy, q, my_r + rad, invT, invDt);
    }

    OSTAMP(2);
    // linearProgram2(lines, maxSpeed, prefVelocity, directionOpt = false)
    float rx, ry;
    if (pvx * pvx + pvy * pvy > vmax * vmax) {
        const float inv = orsqrt<FM>(pvx * pvx + pvy * pvy);
        rx = pvx * inv * vmax; ry = pvy * inv * vmax;
    } else { rx = pvx; ry = pvy; }
    int failed = cnt;
    {
        // linearProgram1(i) does not depend on the point linearProgram2 has reached -- only on line i, the lines j < i, the speed
        // circle and the preferred velocity: its interval [tL, tR], its failure and the point it returns are functions of the LINES.
        // So they are evaluated for every i up front, as one block of independent arithmetic (45 pair bodies + 10 chords with no
        // branch, vote or select chain between them: the two wavefronts of a SIMD overlap their latencies), and the walk over the
        // violated lines is ten short steps.  The operations on the path RVO2 takes are the same on the same operands: tL only
        // grows and tR only shrinks, so RVO2's early exits (an empty interval after some j; a parallel line on the wrong side) equal
        // the tests on the final interval, and min / max are exact in any order.  In the dense phase of a crossing 99.8 % of the
        // agents violate a line and a wavefront of 50 agents enters linearProgram1 for nearly every i anyway (round 3).  Against the
        // round-4 form (linearProgram1(i) inside the walk, behind a vote, with the `ok` chain through its bodies): bit-identical,
        // 472 -> 450 us exact, 431 -> 406 us fast in the dense phase of cfg4 (profiles/r5a_orca_modes_ab.txt).
        bool anyviol = false;
#pragma unroll
        for (int i = 0; i < KF; ++i) anyviol = anyviol || ((i < cnt) && (odet<FM>(Lr[i].z, Lr[i].w, Lr[i].x - rx, Lr[i].y - ry) > 0.0f));
        if (__builtin_amdgcn_ballot_w64(anyviol) != 0) {   // (nobody violates a line at the preferred velocity: it is everybody's result)
            float cx[KF], cy[KF];
            unsigned okbits = 0;
#pragma unroll
            for (int i = 0; i < KF; ++i) {
                const float4 ln = Lr[i];
                const float dot = odot<FM>(ln.x, ln.y, ln.z, ln.w);
                const float disc = dot * dot + vmax * vmax - odot<FM>(ln.x, ln.y, ln.x, ln.y);
                const float sq = osqrt<FM>(fmaxf(disc, 0.0f));
                float tL = -dot - sq, tR = -dot + sq;
                bool failp = false;
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    const float4 lj = Lr[j];
                    const float den = odet<FM>(ln.z, ln.w, lj.z, lj.w);
                    const float num = odet<FM>(lj.z, lj.w, ln.x - lj.x, ln.y - lj.y);
                    const float t = odiv<FM>(num, den);
                    const bool right = den > RVO_EPSILON, left = den < -RVO_EPSILON;      // (neither: parallel, |den| <= RVO_EPSILON)
                    tR = fminf(tR, right ? t : INFINITY);
                    tL = fmaxf(tL, left ? t : -INFINITY);
                    failp = failp || (!right && !left && num < 0.0f);
                }
                float t = odot<FM>(ln.z, ln.w, pvx - ln.x, pvy - ln.y);
                t = (t < tL) ? tL : ((t > tR) ? tR : t);
                cx[i] = omad<FM>(t, ln.z, ln.x);
                cy[i] = omad<FM>(t, ln.w, ln.y);
                okbits |= (!(disc < 0.0f) && !failp && !(tL > tR)) ? (1u << i) : 0u;
            }
            bool done = false;
#pragma unroll
            for (int i = 0; i < KF; ++i) {
                const float4 ln = Lr[i];
                const bool viol = !done && (i < cnt) && (odet<FM>(ln.z, ln.w, ln.x - rx, ln.y - ry) > 0.0f);
                const bool ok = ((okbits >> i) & 1u) != 0;
                const bool set = viol && ok, bad = viol && !ok;
                rx = set ? cx[i] : rx;
                ry = set ? cy[i] : ry;
                failed = bad ? i : failed;
                done = done || bad;
            }
        }
    }
    OSTAMP(3);
    if (__builtin_amdgcn_ballot_w64(failed < cnt) != 0) { // some lane's programme is infeasible: linearProgram3
        if (active) {
#pragma unroll
            for (int k = 0; k < KF; ++k) L.set(k, Lr[k]);
        }
        ORCA_LDS_FENCE();
        if (lp3_static) lp3_fast10<FM>(Lr, L, cnt, failed, vmax, rx, ry);   // (A/B switch, CROWDSTEP_ORCA_LP3=static)
        else lp3_rows<FM>(Lr, L, R, cnt, failed, vmax, rx, ry);
    }
    OSTAMP(4);
    nvx = rx; nvy = ry;
}

//  The crowd kernel's form: the world's rows in LDS, brute-force walk in index order.
//  Called by all 64 lanes of a wavefront (`active` = this lane holds an agent): linearProgram3 re-deals the lanes (lp3_rows).
template <int FM>
__device__ __forceinline__ void orca_velocity_fast10(bool active, bool lp3_static, const float4* pv, const float* rr, int rows, int row, float px, float py, float vx,
                                     float vy, float my_r, float vmax, float pvx, float pvy, float neighbor_dist,
                                     float time_horizon, float dt, const Lines& L, const RowLds& R, float& nvx, float& nvy,
                                     unsigned long long* g_ost, unsigned long long& g_ost_last)
{
    OSTAMP(0);
    double key[10];
    const float range2 = neighbor_dist * neighbor_dist;
    // (row, opaque per substep: the 25 lane masks `b != row` are loop-invariant, the compiler keeps them in 50 SGPRs across the substep
    //  loop, runs out, spills them to VGPR lanes and pays two v_readlane per candidate to get them back; one v_cmp per candidate is cheaper)
    int row_o = row;
    asm volatile("" : "+v"(row_o));
    const float range2_l = active ? range2 : -1.0f;                      // a lane without an agent has nobody in range
    auto candidate = [&](int b) -> double {   // row b's key, or a sentinel (out of range, myself, a lane without an agent)
        const float4 q = pv[b];
        const float ddx = px - q.x, ddy = py - q.y;
        const float dsq = ddx * ddx + ddy * ddy;
        const bool in = (dsq < range2_l) & (b != row_o);                 // (no short circuit: the row is read by every lane, nothing to branch around)
        // a sentinel is any key whose high word is 0x7F7FFFFF (no distance below neighborDist^2 has these bits); its low word is never
        // read (slots k >= cnt), so it can be b as well: one select per key, the row index a constant
        return __hiloint2double(in ? (int)__float_as_uint(dsq) : 0x7F7FFFFF, b);
    };
    // The ten smallest keys in order are a function of the key SET: rows are taken eight at a time through a 19-comparator sorting
    // network and merged into the list by a pruned odd-even merge (25 comparators; orca_sortnet.h, generated and checked by
    // tools/gen_sortnet.py) -- 44 compare-exchanges per eight rows where one-at-a-time insertion takes 80; the rows left over are inserted.
    int b0 = 0;
    if (rows >= 8) {
        double C[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) C[c] = candidate(c);
        ORCA_SORT8(ORCA_CE, C)
#pragma unroll
        for (int c = 0; c < 8; ++c) key[c] = C[c];
        key[8] = key[9] = key_sentinel();
#pragma nounroll
        for (b0 = 8; b0 + 8 <= rows; b0 += 8) {
#pragma unroll
            for (int c = 0; c < 8; ++c) C[c] = candidate(b0 + c);
            ORCA_SORT8(ORCA_CE, C)
            ORCA_MERGE10_8(ORCA_CE, key, C)
        }
    } else {
#pragma unroll
        for (int s = 0; s < 10; ++s) key[s] = key_sentinel();
    }
    for (int b = b0; b < rows; ++b) key_insert10(key, candidate(b));
    orca_solve_fast10<FM>(active, lp3_static, key, row, [&](int b, float4& q, float& rad) { q = pv[b]; rad = rr[b]; }, px, py, vx, vy, my_r, vmax,
                      pvx, pvy, time_horizon, dt, L, R, nvx, nvy, g_ost, g_ost_last);
}

// MAXT = 64: floor(64 / rows) worlds per one-wavefront block; MAXT = 256 / 512: one world of up to MAXT rows per block
// FM: the arithmetic of the register-resident build (0 exact / 1 fast / 2 fast + fma, see odiv above); the generic build is always exact
// at most two wavefronts per SIMD: the register-resident build needs ~220 VGPRs whatever the scheduler is told, and knowing the
// bound it stops trading instructions for registers it cannot use (-0.7 %, profiles/r5b_ab_scheduler_variants.txt)
#define ORCA_WPE_ATTR __attribute__((amdgpu_waves_per_eu(1, 2)))
template <bool FAST10, int MAXT, int FM = 0>
__global__ __launch_bounds__(MAXT) ORCA_WPE_ATTR void k_orca_step(const OArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int T = blockDim.x;
    const int K = a.K;
    const int KL = FAST10 ? 0 : a.K + a.KO;                          // lines per agent: obstacle lines first, then agents
    const int KLL = FAST10 ? 10 : KL;                                // FAST10 keeps a copy of its ten lines for LP3's run-time line index
    // per-lane columns are TL = wpb * rows lanes wide (the lanes that hold an agent: 50 of 64 for 25-agent worlds); the
    // register-resident build (FAST10) keeps lines, projected lines and neighbour keys in registers and has no columns at all
    const int TL = a.wpb * a.rows;
    const int KN = FAST10 ? 0 : K;
    const int KLP = FAST10 ? 0 : KL;
    float4* lds_pv = reinterpret_cast<float4*>(smem_raw);            // [2][T] x, y, vx, vy
    float4* lds_L = lds_pv + 2 * T;                                  // [KL][TL] ORCA lines
    float4* lds_P = lds_L + KLL * TL;                                // [KL][TL] LP3 projection lines
    float4* lds_rowP = lds_P + KLP * TL;                             // lp3_rows (FAST10, one-wavefront blocks): [9][8] projected lines,
    float2* lds_rowA = reinterpret_cast<float2*>(lds_rowP + ((FAST10 && !a.lp3_static) ? 72 : 0));   // [9][8] their chords,
    float4* lds_q = reinterpret_cast<float4*>(lds_rowA + ((FAST10 && !a.lp3_static) ? 72 : 0));       // [T] (result, maxSpeed, distance) per agent
    float* lds_r = reinterpret_cast<float*>(lds_q + (FAST10 ? T : 0)); // [T] radius + margin
    float* lds_nd = lds_r + T;                                       // [KN][TL] neighbour distSq
    int* lds_ni = reinterpret_cast<int*>(lds_nd + KN * TL);          // [KN][TL] neighbour row
    float* lds_rp = reinterpret_cast<float*>(lds_ni + KN * TL);      // [T] plain radius (respawn rule)
    float* lds_g0x = lds_rp + T;                                     // [T] respawn scratch
    int* lds_flag = reinterpret_cast<int*>(lds_g0x + T);             // [T] respawn scratch
    float* lds_od = reinterpret_cast<float*>(lds_flag + T);          // [KO][TL] obstacle edge distSq
    int* lds_oi = reinterpret_cast<int*>(lds_od + a.KO * TL);        // [KO][TL] obstacle edge (first vertex)
    int* lds_sel = lds_oi + a.KO * TL;                               // [T] lp3_rows tickets (FAST10)

    const int tid = threadIdx.x;
    const int rows = a.rows, n = a.n;
    const int lw = tid / rows, row = tid - lw * rows;
    const int w = blockIdx.x * a.wpb + lw;
    const bool valid = (lw < a.wpb) && (w < a.W);
    const bool robot_row = (a.flags & CS_ROBOT_ROW) != 0;
    const bool human = valid && row < n;
    const bool is_robot = valid && robot_row && row == n;
    const int base = lw * rows;
    const float dt = a.dt;
    const bool respawn_here = valid && (a.world_flags == nullptr || (a.world_flags[w] & 1));

    float px = 0, py = 0, vx = 0, vy = 0, pvx = 0, pvy = 0, r = 0, vmax = 0, margin = 0;
    float th = 0, om = 0;
    const long sidx = (long)w * rows + row;
    float* srow = nullptr;
    if (valid) {
        srow = a.S + sidx * a.as;
        const long fs = a.fs;
        px = srow[0]; py = srow[fs]; th = srow[2 * fs]; vx = srow[3 * fs]; vy = srow[4 * fs];
        pvx = srow[5 * fs]; pvy = srow[6 * fs]; om = srow[7 * fs]; r = srow[8 * fs]; vmax = srow[12 * fs];
        margin = a.margin[sidx];
    }
    float* gi = nullptr;
    float g0x = 0, g0y = 0;
    if (human) { gi = a.goals + ((long)w * n + row) * a.G * 2; g0x = gi[0]; g0y = gi[1]; }

    // the true robot (moves with the action); the simulator's copy is the state row
    const bool has_robot = a.robot != nullptr;
    const bool robot_moves = a.action != nullptr && has_robot;
    float rbx = 0, rby = 0, rbt = 0, rbvx = 0, rbvy = 0, ax = 0, ay = 0;
    if (valid && has_robot) {
        const float* rb = a.robot + (long)w * 13;
        rbx = rb[0]; rby = rb[1]; rbt = rb[2]; rbvx = rb[3]; rbvy = rb[4];
        if (robot_moves) { ax = a.action[(long)w * 2]; ay = a.action[(long)w * 2 + 1]; }
    }

    if (valid) { lds_pv[tid] = make_float4(px, py, vx, vy); lds_r[tid] = r + margin; lds_rp[tid] = r; }
    __syncthreads();

    const Lines L{lds_L, TL, tid}, P{lds_P, TL, tid};
    const RowLds RL{lds_rowP, lds_rowA, lds_q, lds_sel};
    unsigned long long ost_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long* g_ost = nullptr;
    unsigned long long g_ost_last = 0;
#ifdef CS_STAMPS
    g_ost = ost_acc;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(g_ost_last)::"memory");
#endif
    int cur = 0;
    const bool prio_young = MAXT == 64 && (int)blockIdx.x >= a.young_from;   // (of two ready wavefronts of equal priority the arbiter issues the older: the younger takes every other substep)
    if constexpr (MAXT == 64) __builtin_amdgcn_s_setprio(1);   // (base priority 1: above the generator's wavefronts of a refill pass, sfmstep_kernel.h)
    for (int sub = 0; sub < a.nsub; ++sub) {
        if constexpr (MAXT == 64) { if (prio_young) { if (sub & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); } }
        const int nxt = cur ^ 1;
        if (valid && robot_moves && (is_robot || (!robot_row && row == 0)))   // robot.step(action, dt): holonomic or unicycle (robotstep.h)
            csimpl::robot_action_step(a.flags, rbx, rby, rbt, rbvx, rbvy, ax, ay, dt);
        float nvx = 0.0f, nvy = 0.0f;
        if constexpr (FAST10) {
            // every lane of the wavefront takes part (linearProgram3 re-deals the lanes); lanes without a human carry no lines
            const int hb = human ? base : 0, hr = human ? row : 0;
            orca_velocity_fast10<FM>(human, a.lp3_static != 0, lds_pv + cur * T + hb, lds_r + hb, rows, hr, px, py, vx, vy, r + margin, vmax, pvx, pvy,
                                 a.neighbor_dist, a.time_horizon, dt, L, RL, nvx, nvy, g_ost, g_ost_last);
        }
        if (human) {
            const float4* pv = lds_pv + cur * T + base;
            const float* rr = lds_r + base;
            if constexpr (!FAST10) {
                // this agent's own RVO2 parameters, where the caller gave some (constant over the launch)
                float nbd = a.neighbor_dist, thz = a.time_horizon, tho = a.time_horizon_obst;
                int Kme = K;
                if (a.agent_params != nullptr) {
                    const float* ap = a.agent_params + ((long)w * rows + row) * 4;
                    nbd = ap[0]; thz = ap[2]; tho = ap[3];
                    Kme = (int)ap[1] < K ? (int)ap[1] : K;
                }
                // ---- Agent::computeNeighbors / insertAgentNeighbor (index order; strict <, ties keep order)
                int cnt = 0;
                float rangeSq = nbd * nbd;
                if (Kme > 0) {
                    for (int b = 0; b < rows; ++b) {
                        if (b == row) continue;
                        const float4 q = pv[b];
                        const float ddx = px - q.x, ddy = py - q.y;
                        const float dsq = ddx * ddx + ddy * ddy;
                        if (dsq < rangeSq) {
                            if (cnt < Kme) ++cnt;
                            int i = cnt - 1;
                            while (i != 0 && dsq < lds_nd[(i - 1) * TL + tid]) {
                                lds_nd[i * TL + tid] = lds_nd[(i - 1) * TL + tid];
                                lds_ni[i * TL + tid] = lds_ni[(i - 1) * TL + tid];
                                --i;
                            }
                            lds_nd[i * TL + tid] = dsq;
                            lds_ni[i * TL + tid] = b;
                            if (cnt == Kme) rangeSq = lds_nd[(cnt - 1) * TL + tid];
                        }
                    }
                }
                // ---- Agent::computeNewVelocity: obstacle half-planes first (static-obstacle worlds), then one per neighbour
                int nobst = 0;
                if (a.nv > 0) {
                    const float rng = tho * vmax + (r + margin);   // rangeSq = sqr(timeHorizonObst * maxSpeed + radius)
                    const int no = obstacle_neighbors(a.verts, a.nv, a.KO, px, py, rng * rng, lds_od, lds_oi, TL, tid);
                    nobst = obstacle_lines(a.verts, lds_oi, no, TL, tid, px, py, vx, vy, r + margin, 1.0f / tho, L);
                }
                const float invT = 1.0f / thz;
                const float invDt = 1.0f / dt;
                for (int k = 0; k < cnt; ++k) {
                    const int b = lds_ni[k * TL + tid];
                    L.set(nobst + k, orca_line(px, py, vx, vy, pv[b], (r + margin) + rr[b], invT, invDt));
                }
                const int total = nobst + cnt;
                const int failed = lp2(L, total, vmax, pvx, pvy, false, nvx, nvy);
                if (failed < total) lp3(L, P, total, nobst, failed, vmax, nvx, nvy);
            }
            // ---- Agent::update, then the reference's read-back + update_goals_orca (:390-394, :125-133)
            vx = nvx; vy = nvy;
            px += vx * dt; py += vy * dt;
            float ddx = g0x - px, ddy = g0y - py;
            if (osqrt<FM>(ddx * ddx + ddy * ddy) < r) { // update_goals: strict <  (:66-70)
                int k = a.G;
                for (int g = 0; g < a.G; ++g) if (isnan(gi[2 * g])) { k = g; break; }
                if (a.peek_out == nullptr) {
                    const float r0 = gi[0], r1 = gi[1];
                    for (int g = 0; g + 1 < k; ++g) { gi[2 * g] = gi[2 * g + 2]; gi[2 * g + 1] = gi[2 * g + 3]; }
                    if (k > 0) { gi[2 * (k - 1)] = r0; gi[2 * (k - 1) + 1] = r1; }
                    g0x = gi[0]; g0y = gi[1];
                } else if (k > 1) { g0x = gi[2]; g0y = gi[3]; }
                ddx = g0x - px; ddy = g0y - py;
            }
            const float nrm = osqrt<FM>(ddx * ddx + ddy * ddy);
            if constexpr (FM != 0) {
                const float inrm = __builtin_amdgcn_rcpf(nrm);
                const bool far = nrm > vmax;
                pvx = far ? ddx * inrm : ddx; pvy = far ? ddy * inrm : ddy;
            } else {
                if (nrm > vmax) { pvx = ieee_div(ddx, nrm); pvy = ieee_div(ddy, nrm); } else { pvx = ddx; pvy = ddy; }
            }
            lds_pv[nxt * T + tid] = make_float4(px, py, vx, vy);
        } else if (is_robot) {
            // set_state_orca(robot) AFTER doStep (:389): the simulator's robot agent takes the true state (moved by the
            // action of this launch, or by whoever updated w->d_robot before it: cs_robot_model_step)
            if (has_robot) { px = rbx; py = rby; vx = rbvx; vy = rbvy; }
            lds_pv[nxt * T + tid] = make_float4(px, py, vx, vy);
        }
        __syncthreads();
        if (a.flags & CS_RESPAWN) { // motion_model_manager.py:407-422, sequential inside a world
            const float rdx = px - g0x, rdy = py - g0y;
            const int flag = (human && respawn_here && osqrt<FM>(rdx * rdx + rdy * rdy) < 3.0f) ? 1 : 0;
            bool any_flag;
            if constexpr (MAXT == 64) any_flag = __builtin_amdgcn_ballot_w64(flag != 0) != 0;
            else any_flag = __syncthreads_or(flag) != 0;        // a world spans several wavefronts: block-wide vote
            if (any_flag) {
                lds_flag[tid] = flag;
                lds_g0x[tid] = g0x;
                __syncthreads();
                if (valid && row == 0) {
                    float4* pvn = lds_pv + nxt * T + base;
                    const float* rp = lds_rp + base;
                    // h.radius + h.safety_space with safety_space == 0 for ORCA humans (mmm.py:154-158); the maxima of the world as the first
                    // respawned human finds it: the c-th lands c steps behind (respawnx.h: the reference's float64 sum, rounded once at :416)
                    float mx = pvn[0].x, mr = rp[0];
                    for (int j = 1; j < n; ++j) { mx = fmaxf(mx, pvn[j].x); mr = fmaxf(mr, rp[j]); }
                    if (robot_row) { mx = fmaxf(mx, pvn[n].x); mr = fmaxf(mr, rp[n]); }
                    int c = 0;
                    for (int i = 0; i < n; ++i) {
                        if (!lds_flag[base + i]) continue;
                        float4 q = pvn[i];
                        q.x = csimpl::respawn_x(mx, mr, a.bx, c++);
                        q.y = (q.y >= 0.0f) ? fminf(q.y, a.by) : fmaxf(q.y, -a.by);
                        pvn[i] = q;
                    }
                }
                __syncthreads();
                if (flag) {
                    const float4 q = lds_pv[nxt * T + tid];
                    px = q.x; py = q.y; g0y = py;
                    if (a.peek_out == nullptr) for (int g = 0; g < a.G; ++g) { gi[2 * g] = g0x; gi[2 * g + 1] = g0y; }
                }
            }
        }
        OSTAMP(5);
        cur = nxt;
    }
#ifdef CS_STAMPS
    if (a.stamps && (threadIdx.x & 63) == 0)
        for (int k = 0; k < 8; ++k) a.stamps[(size_t)blockIdx.x * 8 + k] = ost_acc[k];
#endif

    if (a.peek_out != nullptr) { // get_human_states(include_goal=True, headed=False) of the next state
        if (human) {
            float* o = a.peek_out + ((long)w * n + row) * 8;
            o[0] = px; o[1] = py; o[2] = th; o[3] = vx; o[4] = vy; o[5] = om; o[6] = g0x; o[7] = g0y;
        }
        return;
    }
    if (valid) {
        const long fs = a.fs;
        srow[0] = px; srow[fs] = py; srow[3 * fs] = vx; srow[4 * fs] = vy;
        if (human) { srow[5 * fs] = pvx; srow[6 * fs] = pvy; srow[10 * fs] = g0x; srow[11 * fs] = g0y; }
        if (robot_moves && (is_robot || (!robot_row && row == 0))) {
            float* rb = a.robot + (long)w * 13;
            rb[0] = rbx; rb[1] = rby; rb[2] = rbt; rb[3] = rbvx; rb[4] = rbvy;
        }
    }
}


// ---- the robot's own ORCA model (imitation learning): motion_model_manager.py:580-589 builds a SECOND simulator whose agents
// are the humans (preferred velocity (0, 0), re-set from the true human states before every doStep, :641-643) and the robot
// (last agent); only the robot's new velocity and position are read back (:645-651), so one doStep of that simulator is the
// robot's computeNeighbors + computeNewVelocity + update: one lane per world, the generic LDS-column code above.
// The robot's preferred velocity is update_goals_orca(robot) of its current position (:134-141).
struct ORArgs {
    int W, n, rows, robot_row, write_row, K, KO, nv, wpb;
    float dt, neighbor_dist, time_horizon, time_horizon_obst, robot_margin;
    float* S; long as, fs;
    const float* hmargin;   // [W][rows]
    float* robot;           // [W][13]
    const float* verts;
    int just_velocities;    // update_robot(just_velocities=True), motion_model_manager.py:641-653: the new velocity is kept, the position
                            // stays (the reference puts the robot's simulator agent back on robot.position)
};

__global__ __launch_bounds__(64) void k_orca_robot_step(const ORArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int T = 64, tid = threadIdx.x;
    const int KL = a.K + a.KO;
    float4* lds_L = reinterpret_cast<float4*>(smem_raw);             // [KL][T]
    float4* lds_P = lds_L + KL * T;                                  // [KL][T]
    float* lds_nd = reinterpret_cast<float*>(lds_P + KL * T);        // [K][T]
    int* lds_ni = reinterpret_cast<int*>(lds_nd + a.K * T);          // [K][T]
    float* lds_od = reinterpret_cast<float*>(lds_ni + a.K * T);      // [KO][T]
    int* lds_oi = reinterpret_cast<int*>(lds_od + a.KO * T);         // [KO][T]
    const int w = blockIdx.x * a.wpb + tid;
    if (tid >= a.wpb || w >= a.W) return;
    float* rb = a.robot + (long)w * 13;
    float px = rb[0], py = rb[1], vx = rb[3], vy = rb[4];
    const float r = rb[8] + a.robot_margin, gx = rb[10], gy = rb[11], vmax = rb[12];
    float pvx, pvy;
    {
        const float ddx = gx - px, ddy = gy - py;
        const float nrm = sqrtf(ddx * ddx + ddy * ddy);
        if (nrm > vmax) { pvx = ddx / nrm; pvy = ddy / nrm; } else { pvx = ddx; pvy = ddy; }
    }
    const Lines L{lds_L, T, tid}, P{lds_P, T, tid};
    const float* Sw = a.S + (long)w * a.rows * a.as;
    int cnt = 0;
    float rangeSq = a.neighbor_dist * a.neighbor_dist;
    if (a.K > 0) {
        for (int b = 0; b < a.n; ++b) {
            const float* s = Sw + (long)b * a.as;
            const float ddx = px - s[0], ddy = py - s[a.fs];
            const float dsq = ddx * ddx + ddy * ddy;
            if (dsq < rangeSq) {
                if (cnt < a.K) ++cnt;
                int i = cnt - 1;
                while (i != 0 && dsq < lds_nd[(i - 1) * T + tid]) {
                    lds_nd[i * T + tid] = lds_nd[(i - 1) * T + tid];
                    lds_ni[i * T + tid] = lds_ni[(i - 1) * T + tid];
                    --i;
                }
                lds_nd[i * T + tid] = dsq;
                lds_ni[i * T + tid] = b;
                if (cnt == a.K) rangeSq = lds_nd[(cnt - 1) * T + tid];
            }
        }
    }
    int nobst = 0;
    if (a.nv > 0) {
        const float rng = a.time_horizon_obst * vmax + r;
        const int no = obstacle_neighbors(a.verts, a.nv, a.KO, px, py, rng * rng, lds_od, lds_oi, T, tid);
        nobst = obstacle_lines(a.verts, lds_oi, no, T, tid, px, py, vx, vy, r, 1.0f / a.time_horizon_obst, L);
    }
    const float invT = 1.0f / a.time_horizon;
    const float invDt = 1.0f / a.dt;
    for (int k = 0; k < cnt; ++k) {
        const int b = lds_ni[k * T + tid];
        const float* s = Sw + (long)b * a.as;
        const float4 q = make_float4(s[0], s[a.fs], s[3 * a.fs], s[4 * a.fs]);
        L.set(nobst + k, orca_line(px, py, vx, vy, q, r + (s[8 * a.fs] + a.hmargin[(long)w * a.rows + b]), invT, invDt));
    }
    const int total = nobst + cnt;
    float nvx, nvy;
    const int failed = lp2(L, total, vmax, pvx, pvy, false, nvx, nvy);
    if (failed < total) lp3(L, P, total, nobst, failed, vmax, nvx, nvy);
    vx = nvx; vy = nvy;
    if (!a.just_velocities) { px += vx * a.dt; py += vy * a.dt; }
    rb[0] = px; rb[1] = py; rb[3] = vx; rb[4] = vy;
    if (a.write_row) {
        float* s = a.S + ((long)w * a.rows + a.n) * a.as;
        s[0] = px; s[a.fs] = py; s[3 * a.fs] = vx; s[4 * a.fs] = vy;
    }
}


// The same for the reference's defaults (maxNeighbors = 10) without walls: the register-resident solve of the crowd kernel
// (orca_velocity_fast10), one lane per world, the humans' rows of the block's worlds staged in LDS by all 64 lanes.
__global__ __launch_bounds__(64) void k_orca_robot_step_fast(const ORArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, ent = a.n + 1;                       // the humans and the robot itself as the last row
    float4* lds_pv = reinterpret_cast<float4*>(smem_raw);            // [wpb][ent] x, y, vx, vy
    float* lds_rr = reinterpret_cast<float*>(lds_pv + a.wpb * ent);  // [wpb][ent] radius + margin
    float4* lds_ln = reinterpret_cast<float4*>(lds_rr + ((a.wpb * ent + 3) & ~3));   // [10][64] the lanes' ORCA lines for LP3
    float4* lds_pr = lds_ln + 10 * 64;                               // [9][8] projected lines of the rows in flight (lp3_rows)
    float2* lds_pa = reinterpret_cast<float2*>(lds_pr + 72);         // [9][8] their chords
    float4* lds_q = reinterpret_cast<float4*>(lds_pa + 72);          // [64]
    int* lds_sel = reinterpret_cast<int*>(lds_q + 64);               // [64]
    const int w0 = blockIdx.x * a.wpb;
    for (int k = tid; k < a.wpb * a.n; k += 64) {
        const int wl = k / a.n, b = k - wl * a.n, w = w0 + wl;
        if (w < a.W) {
            const float* s = a.S + ((long)w * a.rows + b) * a.as;
            lds_pv[wl * ent + b] = make_float4(s[0], s[a.fs], s[3 * a.fs], s[4 * a.fs]);
            lds_rr[wl * ent + b] = s[8 * a.fs] + a.hmargin[(long)w * a.rows + b];
        }
    }
    const int w = w0 + tid;
    const bool mine = tid < a.wpb && w < a.W;
    float px = 0, py = 0, vx = 0, vy = 0, r = 1, gx = 0, gy = 0, vmax = 1;
    float* rb = a.robot + (long)(mine ? w : 0) * 13;
    if (mine) {
        px = rb[0]; py = rb[1]; vx = rb[3]; vy = rb[4];
        r = rb[8] + a.robot_margin; gx = rb[10]; gy = rb[11]; vmax = rb[12];
        lds_pv[tid * ent + a.n] = make_float4(px, py, vx, vy);
        lds_rr[tid * ent + a.n] = r;
    }
    __syncthreads();
    float pvx, pvy;
    {
        const float ddx = gx - px, ddy = gy - py;
        const float nrm = sqrtf(ddx * ddx + ddy * ddy);
        if (nrm > vmax) { pvx = ddx / nrm; pvy = ddy / nrm; } else { pvx = ddx; pvy = ddy; }
    }
    float nvx, nvy;
    unsigned long long ost_last = 0;
    const Lines L{lds_ln, 64, tid};
    const RowLds RL{lds_pr, lds_pa, lds_q, lds_sel};
    const int mt = mine ? tid : 0;                                    // all 64 lanes take part (lp3_rows re-deals them)
    orca_velocity_fast10<0>(mine, false, lds_pv + mt * ent, lds_rr + mt * ent, ent, a.n, px, py, vx, vy, r, vmax, pvx, pvy, a.neighbor_dist,
                         a.time_horizon, a.dt, L, RL, nvx, nvy, nullptr, ost_last);
    if (!mine) return;
    vx = nvx; vy = nvy;
    if (!a.just_velocities) { px += vx * a.dt; py += vy * a.dt; }
    rb[0] = px; rb[1] = py; rb[3] = vx; rb[4] = vy;
    if (a.write_row) {
        float* s = a.S + ((long)w * a.rows + a.n) * a.as;
        s[0] = px; s[a.fs] = py; s[3 * a.fs] = vx; s[4 * a.fs] = vy;
    }
}

// ---- worlds beyond one block (second half): neighbour search through a uniform grid ---------------------
// RVO2 finds an agent's neighbours with a kd-tree (Agent::computeNeighbors -> KdTree::queryAgentTreeRecursive); what the
// tree returns is the set of the maxNeighbors nearest agents inside neighborDist, ordered by (distSq, insertion order).  The
// crowd kernel above gets the same list by a brute-force walk over the world's rows in LDS; a world of thousands of agents
// spans many blocks, so here the agents are binned into square cells of edge neighborDist (a hashed table of buckets in HBM,
// counting sort per substep) and an agent walks the 3 x 3 cells around its own: every agent inside neighborDist is in one of
// them.  The ten smallest (distSq, row) keys are order-independent, so the visiting order of the grid does not matter: the
// neighbour list, the ORCA lines and the solve are bit-identical to the restatement's index-order walk.  State is double-buffered
// in HBM (every agent reads the old rows, RVO2's doStep is a Jacobi update), one launch per substep.
struct BigArgs {
    int W, n, rows, G, NB, robot_row, flags;
    int K, KO, nv;         // generic build: maxNeighbors, obstacle neighbours kept, obstacle vertices
    float dt, neighbor_dist, time_horizon, time_horizon_obst, inv_cell;
    const float* Sin; float* Sout; long as, fs;
    float* goals; const float* margin;
    const int2* cellxy; const int* start; const int* sorted;   // the grid (csimpl::grid_build, bigworld.hip)
    int lp3_static;
    float* peek_out;       // cs_peek: [W][n][8] rows of the stepped state; goal lists are not rotated in memory
    float* robot;          // [W][13] cs_worlds.d_robot: the true robot (moved by the action), copied into its state row AFTER the step
    const float* action;   // [W][2] or null
    const float* verts;    // RVO2 obstacle vertex records (generic build)
    const float* agent_params;   // [W][rows][4] per-agent neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, or null (see OArgs)
};

// FAST10: maxNeighbors = 10 without static obstacles, the register-resident solve of the crowd kernel.  Otherwise the generic
// LDS-column code (any maxNeighbors <= 16, obstacle lines): the neighbour columns are filled in (distSq, row) order -- what RVO2's
// index-order insertion with its strict < yields -- so the grid's visiting order does not matter here either.
template <bool FAST10>
__global__ __launch_bounds__(64) void k_bw_orca_step(const BigArgs a)
{
    extern __shared__ __align__(16) unsigned char bw_smem[];
    const int KL = FAST10 ? 10 : a.K + a.KO;
    float4* lds_ln = reinterpret_cast<float4*>(bw_smem);                         // [KL][64] ORCA lines
    float4* lds_pr = lds_ln + KL * 64;                                           // FAST10: [72] lp3_rows projections; generic: [KL][64] LP3 projections
    float2* lds_pa = reinterpret_cast<float2*>(lds_pr + (FAST10 ? 72 : KL * 64)); // FAST10: [72] chords
    float4* lds_q = reinterpret_cast<float4*>(lds_pa + (FAST10 ? 72 : 0));       // FAST10: [64]
    int* lds_sel = reinterpret_cast<int*>(lds_q + (FAST10 ? 64 : 0));            // FAST10: [64]
    float* lds_nd = reinterpret_cast<float*>(lds_sel + (FAST10 ? 64 : 0));       // generic: [K][64] neighbour distSq
    int* lds_ni = reinterpret_cast<int*>(lds_nd + (FAST10 ? 0 : a.K * 64));      // generic: [K][64] neighbour row
    float* lds_od = reinterpret_cast<float*>(lds_ni + (FAST10 ? 0 : a.K * 64));  // generic: [KO][64] obstacle edge distSq
    int* lds_oi = reinterpret_cast<int*>(lds_od + (FAST10 ? 0 : a.KO * 64));     // generic: [KO][64] obstacle edge
    const int tid = threadIdx.x, i = blockIdx.x * 64 + tid, w = blockIdx.y, n = a.n, rows = a.rows;
    const bool human = i < n;
    const bool is_robot = a.robot_row && i == n;
    const float* Sw = a.Sin + (long)w * rows * a.as;
    const long fs = a.fs;
    float px = 0, py = 0, vx = 0, vy = 0, pvx = 0, pvy = 0, r = 0, vmax = 0, margin = 0;
    const float* srow = Sw + (long)(i < rows ? i : 0) * a.as;
    if (human) {
        px = srow[0]; py = srow[fs]; vx = srow[3 * fs]; vy = srow[4 * fs];
        pvx = srow[5 * fs]; pvy = srow[6 * fs]; r = srow[8 * fs]; vmax = srow[12 * fs];
        margin = a.margin[(long)w * rows + i];
    }
    const int2* cxy = a.cellxy + (long)w * rows;
    const int* st = a.start + (long)w * a.NB;          // positions in the job-wide sorted list
    const int* so = a.sorted;
    const float* mg = a.margin + (long)w * rows;
    // this agent's own RVO2 parameters, where the caller gave some (the grid's cell edge is the scalar neighborDist: the largest one)
    float nbd = a.neighbor_dist, thz = a.time_horizon, tho = a.time_horizon_obst;
    int Kme = a.K;
    if (a.agent_params != nullptr && human) {
        const float* ap = a.agent_params + ((long)w * rows + i) * 4;
        nbd = ap[0]; thz = ap[2]; tho = ap[3];
        Kme = (int)ap[1] < a.K ? (int)ap[1] : a.K;
    }
    const float range2 = nbd * nbd;
    // Agent::computeNeighbors through the grid: the 3 x 3 cells around mine hold every agent closer than neighborDist
    auto walk = [&](auto&& visit) {
        const int2 mc = cxy[i];
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int cx = mc.x + dx, cy = mc.y + dy;
                const int bk = csimpl::cell_bucket(cx, cy, a.NB);
                for (int p = st[bk]; p < st[bk + 1]; ++p) {
                    const int b = so[p];
                    const int2 cb = cxy[b];
                    if (cb.x != cx || cb.y != cy || b == i) continue;      // another cell hashed into this bucket, or myself
                    const float* sb = Sw + (long)b * a.as;
                    const float ddx = px - sb[0], ddy = py - sb[fs];
                    const float dsq = ddx * ddx + ddy * ddy;
                    if (dsq < range2) visit(dsq, b);
                }
            }
    };
    float nvx = 0.0f, nvy = 0.0f;
    if constexpr (FAST10) {
        double key[10];
#pragma unroll
        for (int s = 0; s < 10; ++s) key[s] = key_sentinel();
        if (human) walk([&](float dsq, int b) { key_insert10(key, __hiloint2double((int)__float_as_uint(dsq), b)); });
        unsigned long long ost_last = 0;
        const Lines L{lds_ln, 64, tid};
        const RowLds RL{lds_pr, lds_pa, lds_q, lds_sel};
        orca_solve_fast10<0>(human, a.lp3_static != 0, key, human ? i : 0,
                          [&](int b, float4& q, float& rad) {
                              const float* sb = Sw + (long)b * a.as;
                              q = make_float4(sb[0], sb[fs], sb[3 * fs], sb[4 * fs]);
                              rad = sb[8 * fs] + mg[b];
                          },
                          px, py, vx, vy, r + margin, vmax, pvx, pvy, a.time_horizon, a.dt, L, RL, nvx, nvy, nullptr, ost_last);
    } else if (human) {
        const int K = Kme;
        int cnt = 0;
        if (K > 0)
            walk([&](float dsq, int b) {
                // insertAgentNeighbor over an index-order walk keeps the K smallest (distSq, row) pairs, ties by row: the same list
                // from any visiting order when the comparison is lexicographic
                auto before = [&](int s) { const float d = lds_nd[s * 64 + tid]; return dsq < d || (dsq == d && b < lds_ni[s * 64 + tid]); };
                int s;
                if (cnt < K) s = cnt++;
                else if (before(K - 1)) s = K - 1;
                else return;
                while (s != 0 && before(s - 1)) {
                    lds_nd[s * 64 + tid] = lds_nd[(s - 1) * 64 + tid];
                    lds_ni[s * 64 + tid] = lds_ni[(s - 1) * 64 + tid];
                    --s;
                }
                lds_nd[s * 64 + tid] = dsq;
                lds_ni[s * 64 + tid] = b;
            });
        const Lines L{lds_ln, 64, tid}, P{lds_pr, 64, tid};
        int nobst = 0;
        if (a.nv > 0) {
            const float rng = tho * vmax + (r + margin);   // rangeSq = sqr(timeHorizonObst * maxSpeed + radius)
            const int no = obstacle_neighbors(a.verts, a.nv, a.KO, px, py, rng * rng, lds_od, lds_oi, 64, tid);
            nobst = obstacle_lines(a.verts, lds_oi, no, 64, tid, px, py, vx, vy, r + margin, 1.0f / tho, L);
        }
        const float invT = 1.0f / thz;
        const float invDt = 1.0f / a.dt;
        for (int k = 0; k < cnt; ++k) {
            const int b = lds_ni[k * 64 + tid];
            const float* sb = Sw + (long)b * a.as;
            L.set(nobst + k, orca_line(px, py, vx, vy, make_float4(sb[0], sb[fs], sb[3 * fs], sb[4 * fs]), (r + margin) + (sb[8 * fs] + mg[b]), invT, invDt));
        }
        const int total = nobst + cnt;
        const int failed = lp2(L, total, vmax, pvx, pvy, false, nvx, nvy);
        if (failed < total) lp3(L, P, total, nobst, failed, vmax, nvx, nvy);
    }
    // the true robot: robot.step(action, dt) (holonomic), then set_state_orca(robot) AFTER doStep (motion_model_manager.py:389): the
    // humans of this substep saw the row as the previous substep left it
    if (a.peek_out == nullptr && (is_robot || (!a.robot_row && i == 0 && a.robot != nullptr && a.action != nullptr))) {
        float rbx = srow[0], rby = srow[fs], rbvx = srow[3 * fs], rbvy = srow[4 * fs];
        if (a.robot != nullptr) {
            float* rb = a.robot + (long)w * 13;
            rbx = rb[0]; rby = rb[1]; rbvx = rb[3]; rbvy = rb[4];
            if (a.action != nullptr) {
                const float ax = a.action[(long)w * 2], ay = a.action[(long)w * 2 + 1];
                float rbt = rb[2];
                csimpl::robot_action_step(a.flags, rbx, rby, rbt, rbvx, rbvy, ax, ay, a.dt);
                rb[0] = rbx; rb[1] = rby; rb[2] = rbt; rb[3] = rbvx; rb[4] = rbvy;
            }
        }
        if (is_robot) {
            float* o = a.Sout + ((long)w * rows + i) * a.as;
            o[0] = rbx; o[fs] = rby; o[3 * fs] = rbvx; o[4 * fs] = rbvy;
        }
    }
    if (!human) return;
    // Agent::update, then the reference's read-back + update_goals_orca (motion_model_manager.py:390-394, :125-133)
    vx = nvx; vy = nvy;
    px += vx * a.dt; py += vy * a.dt;
    float* gi = a.goals + ((long)w * n + i) * a.G * 2;
    float g0x = gi[0], g0y = gi[1];
    float ddx = g0x - px, ddy = g0y - py;
    if (sqrtf(ddx * ddx + ddy * ddy) < r) { // update_goals: strict <  (:66-70)
        int k = a.G;
        for (int g = 0; g < a.G; ++g) if (isnan(gi[2 * g])) { k = g; break; }
        if (a.peek_out == nullptr) {
            const float r0 = gi[0], r1 = gi[1];
            for (int g = 0; g + 1 < k; ++g) { gi[2 * g] = gi[2 * g + 2]; gi[2 * g + 1] = gi[2 * g + 3]; }
            if (k > 0) { gi[2 * (k - 1)] = r0; gi[2 * (k - 1) + 1] = r1; }
            g0x = gi[0]; g0y = gi[1];
        } else if (k > 1) { g0x = gi[2]; g0y = gi[3]; }   // cs_peek commits nothing: the head the rotated list would have
        ddx = g0x - px; ddy = g0y - py;
    }
    const float nrm = sqrtf(ddx * ddx + ddy * ddy);
    if (nrm > vmax) { pvx = ddx / nrm; pvy = ddy / nrm; } else { pvx = ddx; pvy = ddy; }
    if (a.peek_out != nullptr) {   // get_human_states(include_goal=True, headed=False) of the next state (:294-298)
        float* q = a.peek_out + ((long)w * n + i) * 8;
        q[0] = px; q[1] = py; q[2] = srow[2 * fs]; q[3] = vx; q[4] = vy; q[5] = srow[7 * fs]; q[6] = g0x; q[7] = g0y;
        return;
    }
    float* o = a.Sout + ((long)w * rows + i) * a.as;
    o[0] = px; o[fs] = py; o[3 * fs] = vx; o[4 * fs] = vy; o[5 * fs] = pvx; o[6 * fs] = pvy; o[10 * fs] = g0x; o[11 * fs] = g0y;
}

// ---- diagnostic: ieee_div / ieee_sqrt against the compiler's operators, bit for bit, on random operands of the range the
// linear programmes work in (exponents 2^-60 .. 2^40, both signs, every mantissa pattern equally likely)
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float rand_operand(unsigned u)
{
    const unsigned expo = 67u + ((u >> 23) & 0xFFu) % 101u;          // biased exponent 67 .. 167  = 2^-60 .. 2^40
    return __uint_as_float((u & 0x80000000u) | (expo << 23) | (u & 0x007FFFFFu));
}
__global__ void k_divsqrt_check(unsigned long long per_thread, unsigned seed, unsigned long long* out)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad_div = 0, bad_sqrt = 0;
    for (unsigned long long k = 0; k < per_thread; ++k) {
        const unsigned long long h = mix64((gid * per_thread + k) ^ ((unsigned long long)seed << 40));
        const float a = rand_operand((unsigned)h), b = rand_operand((unsigned)(h >> 32));
        const float q0 = a / b, q1 = ieee_div(a, b);
        const float x = fabsf(a), s0 = sqrtf(x), s1 = ieee_sqrt(x);
        const float z = (k & 1023) == 0 ? 0.0f : a;                  // exact zeros: numerators and discriminants can be 0
        const float q2 = z / b, q3 = ieee_div(z, b);
        if (__float_as_uint(q0) != __float_as_uint(q1) || __float_as_uint(q2) != __float_as_uint(q3)) {
            if (bad_div == 0) { out[2] = __float_as_uint(a); out[3] = __float_as_uint(b); }
            ++bad_div;
        }
        if (__float_as_uint(s0) != __float_as_uint(s1) || __float_as_uint(sqrtf(0.0f * x)) != __float_as_uint(ieee_sqrt(0.0f * x))) ++bad_sqrt;
    }
    if (bad_div) atomicAdd(&out[0], bad_div);
    if (bad_sqrt) atomicAdd(&out[1], bad_sqrt);
}

} // namespace

namespace csimpl {

// worlds of more than `dflt` rows take the grid path; CROWDSTEP_BIGWORLD_MIN_ROWS lowers the threshold (tests run the grid path
// on worlds small enough for the restatement to check quickly)
int big_world_min_rows(int dflt)
{
    const char* e = std::getenv("CROWDSTEP_BIGWORLD_MIN_ROWS");
    const int v = e ? std::atoi(e) : dflt;
    return v < dflt ? (v < 1 ? 1 : v) : dflt;
}

// The arithmetic of the register-resident build (k_orca_step<FAST10 = true>) is a field of the context (cs_worlds.orca_math, ABI 4): exact
// (bit-identical to the restatement; THE DEFAULT), fast (v_rcp / v_sqrt / v_rsq) or fast + fma.  CS_ORCA_MATH_DEFAULT resolves to
// CROWDSTEP_ORCA_MATH=exact|fast|fma (read once) else exact.  Returns the template parameter FM: 0 exact, 1 fast, 2 fma.
int orca_default_math()
{
    static const int dflt = [] {
        const char* e = std::getenv("CROWDSTEP_ORCA_MATH");
        if (e && std::strcmp(e, "fast") == 0) return (int)CS_ORCA_MATH_FAST;
        if (e && std::strcmp(e, "fma") == 0) return (int)CS_ORCA_MATH_FMA;
        return (int)CS_ORCA_MATH_EXACT;
    }();
    return dflt;
}
static int orca_math_of(const cs_worlds* w)
{
    const int m = w->orca_math == CS_ORCA_MATH_DEFAULT ? orca_default_math() : w->orca_math;
    return m - CS_ORCA_MATH_EXACT;
}

// dynamic LDS of the one-block kernel (k_orca_step) for these worlds: [2][T] rows + radii / respawn scratch, and either the
// register-resident build's line copies (maxNeighbors = 10, no obstacles) or the generic build's per-agent columns
static size_t orca_block_shmem(const cs_worlds* w, bool lp3_static)
{
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    const int T = rows <= 64 ? 64 : (rows <= 256 ? 256 : 512);
    const int wpb = rows <= 64 ? 64 / rows : 1;
    const int TL = wpb * rows;
    const int K = w->orca_max_neighbors, nv = w->orca_n_vertices;
    const int KO = nv > 0 ? (nv < KOBST ? nv : KOBST) : 0;
    const bool fast10 = K == 10 && nv == 0 && w->d_orca_agent_params == nullptr;
    return (size_t)T * (2 * sizeof(float4) + 4 * sizeof(float)) +
           (fast10 ? (size_t)10 * TL * sizeof(float4) + (lp3_static ? 0 : 72 * (sizeof(float4) + sizeof(float2))) + (size_t)T * (sizeof(float4) + sizeof(int))
                   : (size_t)(K + KO) * TL * (2 * sizeof(float4) + 2 * sizeof(float)));
}

// worlds of more than 512 rows, and worlds whose generic-build columns outgrow a block's 160 KB of LDS, take the grid path
bool orca_uses_grid(const cs_worlds* w)
{
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    return rows > big_world_min_rows(512) || orca_block_shmem(w, true) > 160 * 1024;
}

size_t orca_big_scratch_bytes(const cs_worlds* w)
{
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    const size_t state_bytes = (size_t)w->W * rows * 13 * sizeof(float);
    return ((state_bytes + 255) & ~(size_t)255) + grid_bytes(w->W, rows, big_world_buckets(rows));
}

static int orca_big_launch(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float* d_peek, hipStream_t stream)
{
    const int n = w->n, W = w->W;
    const bool robot_row = (w->flags & CS_ROBOT_ROW) != 0;
    const int rows = n + (robot_row ? 1 : 0);
    if (d_action && !w->d_robot) return fail(CS_ERR_ARG, "a robot action needs cs_worlds.d_robot");
    const int NB = big_world_buckets(rows);
    const size_t state_bytes = (size_t)W * rows * 13 * sizeof(float);
    const size_t state_pad = (state_bytes + 255) & ~(size_t)255;
    char* base = nullptr;
    {
        const int rcs = scratch((void**)&base, orca_big_scratch_bytes(w), SCRATCH_ORCA_BIG, stream);
        if (rcs) return rcs;
    }
    BigArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = W; a.n = n; a.rows = rows; a.robot_row = robot_row ? 1 : 0; a.flags = w->flags; a.G = w->G; a.NB = NB; a.dt = dt;
    a.neighbor_dist = w->orca_neighbor_dist; a.time_horizon = w->orca_time_horizon;
    // cell edge = neighborDist (a floor keeps a degenerate neighborDist = 0 from dividing by zero: nobody is a neighbour then)
    a.inv_cell = 1.0f / (w->orca_neighbor_dist > 1e-3f ? w->orca_neighbor_dist : 1e-3f);
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; } else { a.as = 1; a.fs = (long)W * rows; }
    a.goals = w->d_goals; a.margin = w->d_safety;
    a.robot = w->d_robot; a.action = d_action;
    a.K = w->orca_max_neighbors; a.nv = w->orca_n_vertices; a.verts = w->d_orca_vertices; a.time_horizon_obst = w->orca_time_horizon_obst;
    a.KO = a.nv > 0 ? (a.nv < KOBST ? a.nv : KOBST) : 0;
    a.agent_params = w->d_orca_agent_params;
    const bool fast10 = a.K == 10 && a.nv == 0 && a.agent_params == nullptr;   // the register-resident solve: no obstacle lines, one parameter set
    float* S2 = (float*)base;
    void* grid_mem = base + state_pad;
    const char* lp3_env = std::getenv("CROWDSTEP_ORCA_LP3");
    a.lp3_static = (lp3_env && std::strcmp(lp3_env, "static") == 0) ? 1 : 0;
    const size_t shmem = fast10 ? (size_t)(10 * 64 + 72 + 64) * sizeof(float4) + 72 * sizeof(float2) + 64 * sizeof(int)
                                : (size_t)(a.K + a.KO) * 64 * (2 * sizeof(float4)) + (size_t)(a.K + a.KO) * 64 * (sizeof(float) + sizeof(int));
    if (shmem > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_bw_orca_step<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    HIP_TRY(hipMemcpyAsync(S2, w->d_state, state_bytes, hipMemcpyDeviceToDevice, stream));   // the columns a step does not write
    const float* cur = w->d_state;
    float* nxt = S2;
    a.peek_out = d_peek;                       // cs_peek: one substep, rows to d_peek, nothing committed (no respawn: post_update=False, :705)
    for (int sub = 0; sub < n_substeps; ++sub) {
        a.Sin = cur; a.Sout = nxt;
        GridView g;
        const int rcg = grid_build(cur, a.as, a.fs, W, rows, NB, nullptr, a.inv_cell, grid_mem, g, stream);
        if (rcg) return rcg;
        a.cellxy = g.cellxy; a.start = g.start; a.sorted = g.sorted;
        if (fast10) hipLaunchKernelGGL(k_bw_orca_step<true>, dim3((rows + 63) / 64, W), dim3(64), shmem, stream, a);
        else hipLaunchKernelGGL(k_bw_orca_step<false>, dim3((rows + 63) / 64, W), dim3(64), shmem, stream, a);
        if (d_peek) { HIP_TRY(hipGetLastError()); return CS_OK; }
        if (w->flags & CS_RESPAWN)
            big_respawn_launch(nxt, a.as, a.fs, W, n, rows, w->d_goals, w->G, nullptr, 1, w->respawn_bound_x, w->respawn_bound_y, w->d_world_flags, stream);
        const float* t = cur; cur = nxt; nxt = const_cast<float*>(t);
    }
    HIP_TRY(hipGetLastError());
    if (cur != w->d_state) HIP_TRY(hipMemcpyAsync(w->d_state, cur, state_bytes, hipMemcpyDeviceToDevice, stream));
    return CS_OK;
}

int orca_launch(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float* d_peek, hipStream_t stream)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (w->W <= 0 || w->n <= 0 || w->G <= 0) return fail(CS_ERR_ARG, "W, n, G must be positive");
    if (!w->d_state || !w->d_goals || !w->d_safety) return fail(CS_ERR_ARG, "null device buffer in cs_worlds");
    if (w->O != 0) return fail(CS_ERR_ARG, "ORCA worlds take their static obstacles as RVO2 vertex records "
                                              "(cs_worlds.d_orca_vertices), not as the SFM segment array");
    if (w->orca_n_vertices < 0 || (w->orca_n_vertices > 0 && !w->d_orca_vertices)) return fail(CS_ERR_ARG, "bad ORCA obstacle vertices");
    if (w->orca_n_vertices > 0 && !(w->orca_time_horizon_obst > 0.0f)) return fail(CS_ERR_ARG, "bad ORCA parameters");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    if (w->orca_max_neighbors < 0 || w->orca_max_neighbors > KMAX) return fail(CS_ERR_ARG, "orca_max_neighbors must be in 0..16");
    if (!(w->orca_time_horizon > 0.0f) || !(w->orca_neighbor_dist >= 0.0f)) return fail(CS_ERR_ARG, "bad ORCA parameters");
    if (w->orca_math < CS_ORCA_MATH_DEFAULT || w->orca_math > CS_ORCA_MATH_FMA) return fail(CS_ERR_ARG, "cs_worlds.orca_math: CS_ORCA_MATH_DEFAULT / EXACT / FAST / FMA");
    if (orca_uses_grid(w)) return orca_big_launch(w, dt, n_substeps, d_action, d_peek, stream);
    OArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = w->W; a.n = w->n; a.rows = rows; a.G = w->G; a.flags = w->flags; a.nsub = n_substeps;
    a.wpb = rows <= 64 ? 64 / rows : 1; a.K = w->orca_max_neighbors;
    a.dt = dt; a.neighbor_dist = w->orca_neighbor_dist; a.time_horizon = w->orca_time_horizon;
    a.bx = w->respawn_bound_x; a.by = w->respawn_bound_y;
    a.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; } else { a.as = 1; a.fs = (long)w->W * rows; }
    a.goals = w->d_goals; a.margin = w->d_safety; a.robot = w->d_robot; a.action = d_action;
    a.peek_out = d_peek; a.world_flags = w->d_world_flags;
    a.verts = w->d_orca_vertices; a.nv = w->orca_n_vertices; a.time_horizon_obst = w->orca_time_horizon_obst;
    a.KO = a.nv > 0 ? (a.nv < KOBST ? a.nv : KOBST) : 0;
#ifdef CS_STAMPS
    a.stamps = g_stamp_buf;
#endif
    if (d_peek) a.flags &= ~CS_RESPAWN;
    const int T = rows <= 64 ? 64 : (rows <= 256 ? 256 : 512);   // worlds of more than 64 rows: one world per block
    const int grid = (w->W + a.wpb - 1) / a.wpb;
    a.agent_params = w->d_orca_agent_params;
    const bool fast10 = a.K == 10 && a.nv == 0 && a.agent_params == nullptr; // the register-resident solve: no obstacle lines, one parameter set
    const int TL = a.wpb * rows;                // lanes that hold an agent: the width of the per-lane LDS columns
    // linearProgram3 of the register-resident build: one 16-lane row per (agent, violated line) (lp3_rows) in one-wavefront blocks;
    // worlds of more than 64 rows keep the statically unrolled walk (their blocks have no LDS left for the projected lines)
    const char* lp3_env = std::getenv("CROWDSTEP_ORCA_LP3");
    a.lp3_static = (T > 64 || (lp3_env && std::strcmp(lp3_env, "static") == 0)) ? 1 : 0;
    const size_t shmem = orca_block_shmem(w, a.lp3_static != 0);
    a.young_from = (T == 64 && grid == 2 * csimpl::device_simds()) ? grid / 2 : 0x7fffffff;
    auto launch = [&](auto kernel) -> int {
        if (shmem > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(T), shmem, stream, a);
        return CS_OK;
    };
    int rc;
    const int fm = fast10 ? orca_math_of(w) : 0;   // the generic build (other maxNeighbors, static obstacles, per-agent parameters) is always exact
    if (T == 64) {
        if (!fast10) rc = launch(k_orca_step<false, 64>);
        else rc = fm == 0 ? launch(k_orca_step<true, 64, 0>) : (fm == 1 ? launch(k_orca_step<true, 64, 1>) : launch(k_orca_step<true, 64, 2>));
    } else if (T == 256) {
        if (!fast10) rc = launch(k_orca_step<false, 256>);
        else rc = fm == 0 ? launch(k_orca_step<true, 256, 0>) : (fm == 1 ? launch(k_orca_step<true, 256, 1>) : launch(k_orca_step<true, 256, 2>));
    } else {
        if (!fast10) rc = launch(k_orca_step<false, 512>);
        else rc = fm == 0 ? launch(k_orca_step<true, 512, 0>) : (fm == 1 ? launch(k_orca_step<true, 512, 1>) : launch(k_orca_step<true, 512, 2>));
    }
    if (rc != CS_OK) return rc;
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int orca_variant(const cs_worlds* w, char* buf, size_t buflen)
{
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    const int T = rows <= 64 ? 64 : (rows <= 256 ? 256 : 512);
    const bool fast10 = w->orca_max_neighbors == 10 && w->orca_n_vertices == 0 && w->d_orca_agent_params == nullptr;
    const int wpb = rows <= 64 ? 64 / rows : 1;
    if (orca_uses_grid(w)) {
        std::snprintf(buf, buflen, "k_bw_orca_step<FAST10=%d> grid=(%d,%d) block=64 (one launch per substep)", fast10 ? 1 : 0, (rows + 63) / 64, w->W);
        return CS_OK;
    }
    std::snprintf(buf, buflen, "k_orca_step<FAST10=%d,MAXT=%d> grid=%d block=%d wpb=%d math=%s", fast10 ? 1 : 0, T, (w->W + wpb - 1) / wpb, T, wpb,
                  !fast10 ? "exact" : (orca_math_of(w) == 0 ? "exact" : (orca_math_of(w) == 1 ? "fast" : "fma")));
    return CS_OK;
}

int orca_robot_launch(const cs_worlds* w, float robot_margin, const float* d_human_margin, float dt, hipStream_t stream, int just_velocities)
{
    if (w->orca_max_neighbors < 0 || w->orca_max_neighbors > KMAX) return fail(CS_ERR_ARG, "orca_max_neighbors must be in 0..16");
    if (!(w->orca_time_horizon > 0.0f) || !(w->orca_neighbor_dist >= 0.0f)) return fail(CS_ERR_ARG, "bad ORCA parameters");
    if (w->orca_n_vertices < 0 || (w->orca_n_vertices > 0 && !w->d_orca_vertices)) return fail(CS_ERR_ARG, "bad ORCA obstacle vertices");
    if (w->orca_n_vertices > 0 && !(w->orca_time_horizon_obst > 0.0f)) return fail(CS_ERR_ARG, "bad ORCA parameters");
    ORArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = w->W; a.n = w->n; a.robot_row = (w->flags & CS_ROBOT_ROW) ? 1 : 0; a.rows = w->n + a.robot_row;
    a.write_row = a.robot_row && w->type != CS_ORCA;   // an ORCA crowd takes the moved robot after its own doStep (:389)
    a.K = w->orca_max_neighbors; a.nv = w->orca_n_vertices; a.KO = a.nv > 0 ? (a.nv < KOBST ? a.nv : KOBST) : 0;
    a.dt = dt; a.neighbor_dist = w->orca_neighbor_dist; a.time_horizon = w->orca_time_horizon;
    a.time_horizon_obst = w->orca_time_horizon_obst; a.robot_margin = robot_margin;
    a.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; } else { a.as = 1; a.fs = (long)w->W * a.rows; }
    a.hmargin = d_human_margin; a.robot = w->d_robot; a.verts = w->d_orca_vertices; a.just_velocities = just_velocities ? 1 : 0;
    const size_t shmem = (size_t)(a.K + a.KO) * 64 * (2 * sizeof(float4) + 2 * sizeof(float));
    if (shmem > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_orca_robot_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    // one lane per world, 16 worlds per wavefront: the linear programme is a serial, divergent chain per lane (one launch costs
    // about one ORCA substep, ~55 us, whatever the packing: 64 / 32 / 16 / 8 worlds per wavefront measured 1.27 / 1.24 / 1.18 /
    // 1.18 ms per 20-substep imitation step at 4096 worlds), so the packing only has to reach every CU
    a.wpb = 16;
    if (a.K == 10 && a.nv == 0 && a.n + 1 <= 128) {   // the reference's defaults, no walls: register-resident solve
        const size_t sh = (size_t)a.wpb * (a.n + 1) * sizeof(float4) + (size_t)(((a.wpb * (a.n + 1)) + 3) & ~3) * sizeof(float) +
                          (10 + 1) * 64 * sizeof(float4) + 72 * (sizeof(float4) + sizeof(float2)) + 64 * sizeof(int);
        hipLaunchKernelGGL(k_orca_robot_step_fast, dim3((w->W + a.wpb - 1) / a.wpb), dim3(64), sh, stream, a);
        HIP_TRY(hipGetLastError());
        return CS_OK;
    }
    hipLaunchKernelGGL(k_orca_robot_step, dim3((w->W + a.wpb - 1) / a.wpb), dim3(64), shmem, stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

} // namespace csimpl

extern "C" int cs_orca_default_math(void) { return csimpl::orca_default_math(); }

extern "C" int cs_debug_divsqrt_check(unsigned long long n_pairs, unsigned seed, unsigned long long* h_out, void* stream)
{
    if (!h_out) return csimpl::fail(CS_ERR_ARG, "null output");
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(&d, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), (hipStream_t)stream));
    const int block = 256, grid = 4096;
    const unsigned long long per_thread = (n_pairs + (unsigned long long)block * grid - 1) / ((unsigned long long)block * grid);
    hipLaunchKernelGGL(k_divsqrt_check, dim3(grid), dim3(block), 0, (hipStream_t)stream, per_thread, seed, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_out, d, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipFree(d));
    return CS_OK;
}
