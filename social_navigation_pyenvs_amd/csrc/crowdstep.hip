// crowdstep.hip -- hand-written CDNA4 (gfx950) kernels + the C ABI of include/crowdstep.h.
//
// Hot path: the reference's per-substep pedestrian update
//   update_humans_parallel            /root/reference/social_gym/src/forces_parallel.py:185-284
//   MotionModelManager.update_humans  /root/reference/social_gym/src/motion_model_manager.py:354-422
//   SocialNavGym.step substep loop    /root/reference/social_gym/social_nav_gym.py:240-245
// re-designed for MI355X: lane = agent row, floor(64/rows) independent worlds per wavefront,
// the interacting columns (px,py,vx,vy,r+safety) staged in LDS and broadcast-read in the O(N^2)
// pair loop, all substeps of one Gym step fused in one launch with the state in registers, so HBM
// sees each state row once per launch.  No MFMA (there is no dense contraction on this path).
//
// gfx950 only: no portability macros, no CPU fallback.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "crowdstep.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(CS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));      \
    } while (0)

// ------------------------------------------------------------------------------------------
// kernel arguments
// ------------------------------------------------------------------------------------------
enum : int {
    M_COMMIT_GOALS = 1,   // write rotated goals back to d_goals
    M_MUTATE_INPUT = 2,   // reproduce the in-place writes on the input rows (out != in)
    M_PEEK = 4,           // write [n][8] observable rows to peek_out, nothing else
    M_ROBOT_FROM_ARRAY = 8 // robot row comes from d_robot (mmm.py:359), not from the state array
};

struct KArgs {
    int W, n, rows, G, O, Smax;
    int type, flags, mode;
    int nsub, wpb;
    float dt;
    float* Sin;            // mutated only with M_MUTATE_INPUT
    float* Sout;
    long in_as, in_fs, out_as, out_fs;
    float* goals;
    const float* params;
    const float* safety;
    const float* obstacles;
    float* robot;
    const float* action;
    float* peek_out;
    const int* world_flags;
    float bx, by;
};

__device__ __forceinline__ float norm2(float x, float y) { return sqrtf(x * x + y * y); }

// social_gym/src/utils.py:7-13 (Python % == fmod for the operand signs reaching each branch)
__device__ __forceinline__ float bound_angle(float a)
{
    const float two_pi = 6.283185307179586f;
    const float pi = 3.141592653589793f;
    if (a >= two_pi) a = fmodf(a, two_pi);
    if (a <= -two_pi) a = fmodf(a, two_pi);
    if (a > pi) a -= two_pi;
    if (a < -pi) a += two_pi;
    return a;
}

// Social-force parameters used inside the pair loop (subset of the 20-vector, agent.py:268-388)
struct SocP {
    float Ai, Bi, Ci, Di, Ei, k1, k2, lam, gam, ns, ns1;
};

__device__ __forceinline__ SocP load_socp(const float* P)
{
    SocP s;
    s.Ai = P[1]; s.Bi = P[3]; s.Ci = P[5]; s.Di = P[7]; s.Ei = P[9];
    s.k1 = P[10]; s.k2 = P[11]; s.lam = P[12]; s.gam = P[13]; s.ns = P[14]; s.ns1 = P[15];
    return s;
}

// forces_parallel.py:109-130 / :60-83 -- force on (pi, vi) from (pj, vj); rij = r_i+r_j+safety_i+safety_j
template <int SOC>
__device__ __forceinline__ void pair_force(const SocP& p, float pix, float piy, float vix, float viy,
                                           float pjx, float pjy, float vjx, float vjy, float rij,
                                           float& fx, float& fy)
{
    const float dx = pix - pjx, dy = piy - pjy;
    const float dist = norm2(dx, dy);
    const float nx = dx / dist, ny = dy / dist;
    const float rd = rij - dist;
    const float m0 = fmaxf(0.0f, rd);
    if constexpr (SOC < 2) {
        const float tx = -ny, ty = nx;
        const float dv = (vjx - vix) * tx + (vjy - viy) * ty;
        const float fn = p.Ai * expf(rd / p.Bi) + p.k1 * m0;
        float ft = p.k2 * m0 * dv;
        if constexpr (SOC == 1) ft += p.Ci * expf(rd / p.Di);
        fx += fn * nx + ft * tx;
        fy += fn * ny + ft * ty;
    } else {
        const float vdx = vix - vjx, vdy = viy - vjy;
        const float ivx = p.lam * vdx - nx, ivy = p.lam * vdy - ny;
        const float inorm = norm2(ivx, ivy);
        const float ix = ivx / inorm, iy = ivy / inorm;
        const float th = bound_angle(atan2f(ny, nx) - atan2f(iy, ix) + 3.141592653589793f);
        const float k = (th > 0.0f) ? 1.0f : ((th < 0.0f) ? -1.0f : 0.0f);
        const float hx = -iy, hy = ix;
        const float F = p.gam * inorm;
        const float dv = (-vdx) * hx + (-vdy) * hy;
        const float e0 = p.Ei * expf(-dist / F);
        const float a1 = p.ns1 * F * th, a2 = p.ns * F * th;
        const float e1 = expf(-(a1 * a1)), e2 = expf(-(a2 * a2));
        fx -= e0 * (e1 * ix + k * e2 * hx) + p.k1 * m0 * ix + p.k2 * m0 * dv * hx;
        fy -= e0 * (e1 * iy + k * e2 * hy) + p.k1 * m0 * iy + p.k2 * m0 * dv * hy;
    }
}

// ------------------------------------------------------------------------------------------
// the fused SFM / HSFM step kernel
//   SOC    = type % 3  (0 Helbing, 1 Guo, 2 Moussaid)          forces_parallel.py:215
//   HEADED = type / 3  (0 SFM, 1 HSFM torque on desired force, 2 on total force)   :217
//   PEQ    = all_params_equal
//   MAXT   = 64 (one wavefront, floor(64/rows) worlds) or 1024 (one world per block)
// ------------------------------------------------------------------------------------------
template <int SOC, int HEADED, bool PEQ, int MAXT>
__global__ __launch_bounds__(MAXT) void k_sfm_step(const KArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int T = blockDim.x;
    float4* lds_pv = reinterpret_cast<float4*>(smem_raw);          // [2][T] x,y,vx,vy (stored velocity)
    float2* lds_vr = reinterpret_cast<float2*>(lds_pv + 2 * T);    // [2][T] velocity as refreshed in-place
    float* lds_rs = reinterpret_cast<float*>(lds_vr + 2 * T);      // [T] radius + safety
    float* lds_g0x = lds_rs + T;                                   // [T] respawn scratch
    int* lds_flag = reinterpret_cast<int*>(lds_g0x + T);           // [T] respawn scratch

    const int tid = threadIdx.x;
    const int rows = a.rows, n = a.n;
    const int lw = tid / rows;
    const int row = tid - lw * rows;
    const int w = blockIdx.x * a.wpb + lw;
    const bool valid = (lw < a.wpb) && (w < a.W);
    const bool robot_row = (a.flags & CS_ROBOT_ROW) != 0;
    const bool human = valid && row < n;
    const bool is_robot = valid && robot_row && row == n;
    const int base = lw * rows;
    const float dt = a.dt;
    const int obs_type = (a.type == 1 || a.type == 4 || a.type == 7) ? 1 : 0;

    // ---- load my row ------------------------------------------------------------------
    float px = 0, py = 0, th = 0, vx = 0, vy = 0, bvx = 0, bvy = 0, om = 0, r = 0, m = 1, gx = 0, gy = 0, vd = 0;
    float safety = 0;
    const long sidx = (long)w * rows + row;
    if (valid) {
        const float* s = a.Sin + sidx * a.in_as;
        if (is_robot && (a.mode & M_ROBOT_FROM_ARRAY)) {
            const float* rb = a.robot + (long)w * 13;
            px = rb[0]; py = rb[1]; th = rb[2]; vx = rb[3]; vy = rb[4]; bvx = rb[5]; bvy = rb[6]; om = rb[7];
            r = rb[8]; m = rb[9]; gx = rb[10]; gy = rb[11]; vd = rb[12];
        } else {
            const long fs = a.in_fs;
            px = s[0]; py = s[fs]; th = s[2 * fs]; vx = s[3 * fs]; vy = s[4 * fs]; bvx = s[5 * fs];
            bvy = s[6 * fs]; om = s[7 * fs]; r = s[8 * fs]; m = s[9 * fs]; gx = s[10 * fs]; gy = s[11 * fs];
            vd = s[12 * fs];
        }
        safety = a.safety[sidx];
    }
    // parameters: my own row of P for the single-agent forces; P[0] of my world for the pair loop
    // when all_params_equal (forces_parallel.py:220), else my own row (:261)
    float relax_t = 1, Aw = 0, Bw = 1, Cw = 0, Dw = 1, k1 = 0, k2 = 0, ko = 0, kd = 0, alpha = 1, klam = 0;
    SocP sp = {};
    float g0x = gx, g0y = gy;
    float* gi = nullptr;
    if (human) {
        const long pw = (a.flags & CS_PARAMS_SHARED) ? 0 : (long)w * n * 20;
        const float* P = a.params + pw + (long)row * 20;
        relax_t = P[0]; Aw = P[2]; Bw = P[4]; Cw = P[6]; Dw = P[8]; k1 = P[10]; k2 = P[11];
        ko = P[16]; kd = P[17]; alpha = P[18]; klam = P[19];
        sp = load_socp(PEQ ? (a.params + pw) : P);
        gi = a.goals + ((long)w * n + row) * a.G * 2;
        g0x = gi[0]; g0y = gi[1];
    }
    const float* obst = nullptr;
    if (a.O > 0) obst = a.obstacles + ((a.flags & CS_OBSTACLES_SHARED) ? 0 : (long)w * a.O * a.Smax * 4);

    const bool respawn_here = valid && (a.world_flags == nullptr || (a.world_flags[w] & 1));

    // robot action (held for the whole block, social_nav_gym.py:240-243)
    const bool robot_moves = a.action != nullptr;
    float ax = 0, ay = 0;
    if (valid && robot_moves) { ax = a.action[(long)w * 2]; ay = a.action[(long)w * 2 + 1]; }
    auto robot_step = [&]() { // robot_agent.py:114-136
        if (a.flags & CS_ROBOT_UNICYCLE) {
            const float c = cosf(th + ay), s = sinf(th + ay);
            px += c * ax * dt; py += s * ax * dt;
            th = fmodf(th + ay, 6.283185307179586f);
            if (th < 0) th += 6.283185307179586f;
            vx = cosf(th) * ax; vy = sinf(th) * ax;
        } else {
            px += ax * dt; py += ay * dt; vx = ax; vy = ay;
        }
    };

    // ---- prologue: publish substep-0 rows ----------------------------------------------
    if (is_robot && robot_moves) robot_step();
    if (valid) {
        lds_pv[tid] = make_float4(px, py, vx, vy);
        float rvx = vx, rvy = vy;
        if (HEADED > 0 && human) {
            const float c = cosf(th), s = sinf(th);
            rvx = c * bvx + (-s) * bvy;
            rvy = s * bvx + c * bvy;
        }
        lds_vr[tid] = make_float2(rvx, rvy);
        lds_rs[tid] = r + safety;
    }
    __syncthreads();

    int cur = 0;
    for (int sub = 0; sub < a.nsub; ++sub) {
        const int nxt = cur ^ 1;
        if (human) {
            // -- goal switch, forces_parallel.py:226-234 (on the incoming position)
            if (norm2(g0x - px, g0y - py) <= r) {
                int k = a.G;
                for (int g = 0; g < a.G; ++g)
                    if (isnan(gi[2 * g]) || isnan(gi[2 * g + 1])) { k = g; break; }
                if (a.mode & M_COMMIT_GOALS) {
                    const float r0 = gi[0], r1 = gi[1];
                    for (int g = 0; g + 1 < k; ++g) { gi[2 * g] = gi[2 * g + 2]; gi[2 * g + 1] = gi[2 * g + 3]; }
                    if (k > 0) { gi[2 * (k - 1)] = r0; gi[2 * (k - 1) + 1] = r1; }
                    g0x = gi[0]; g0y = gi[1];
                } else if (k > 1) {
                    g0x = gi[2]; g0y = gi[3];
                }
                gx = g0x; gy = g0y;
            }
            // -- rotation matrix and refreshed linear velocity, :254-256
            float c = 1.0f, s = 0.0f, cvx = vx, cvy = vy;
            if constexpr (HEADED > 0) {
                c = cosf(th); s = sinf(th);
                cvx = c * bvx + (-s) * bvy;
                cvy = s * bvx + c * bvy;
            }
            // -- desired force, :23-40
            float fdx = 0.0f, fdy = 0.0f;
            {
                const float dx = gx - px, dy = gy - py;
                const float dist = norm2(dx, dy);
                if (dist > r) {
                    const float ex = dx / dist, ey = dy / dist;
                    fdx = m * (ex * vd - cvx) / relax_t;
                    fdy = m * (ey * vd - cvy) / relax_t;
                }
            }
            // -- obstacle force: closest point per polygon :236-252, then :136-162
            float fox = 0.0f, foy = 0.0f;
            if (obst != nullptr) {
                for (int o = 0; o < a.O; ++o) {
                    float best = 0.0f, bxp = 0.0f, byp = 0.0f;
                    bool have = false;
                    for (int sg = 0; sg < a.Smax; ++sg) {
                        const float4 seg = *reinterpret_cast<const float4*>(obst + ((long)o * a.Smax + sg) * 4);
                        float d, hx = 0.0f, hy = 0.0f;
                        if (isnan(seg.x)) {
                            d = 9223372036854775807.0f;
                        } else {
                            const float ex = seg.z - seg.x, ey = seg.w - seg.y;
                            const float len = norm2(ex, ey);
                            const float t = ((px - seg.x) * ex + (py - seg.y) * ey) / (len * len);
                            float ts = t > 0.0f ? t : 0.0f;
                            ts = ts < 1.0f ? ts : 1.0f;
                            hx = seg.x + ts * ex; hy = seg.y + ts * ey;
                            d = norm2(hx - px, hy - py);
                        }
                        if (!have || d < best) { best = d; bxp = hx; byp = hy; have = true; }
                    }
                    const float dx = px - bxp, dy = py - byp;
                    const float dist = norm2(dx, dy);
                    const float nx = dx / dist, ny = dy / dist;
                    const float tx = -ny, ty = nx;
                    const float dv = -(cvx * tx + cvy * ty);
                    const float rd = r - dist + safety;
                    const float m0 = fmaxf(0.0f, rd);
                    const float fn = Aw * expf(rd / Bw) + k1 * m0;
                    if (obs_type == 0) {
                        const float ft = k2 * m0 * dv;
                        fox += fn * nx - ft * tx;
                        foy += fn * ny - ft * ty;
                    } else {
                        const float ft = (-Cw * expf(rd / Dw) - k2 * m0) * dv;
                        fox += fn * nx + ft * tx;
                        foy += fn * ny + ft * ty;
                    }
                }
                fox /= (float)a.O; foy /= (float)a.O;
            }
            // -- social force: O(N) partners broadcast from LDS, :87-133 / :43-84
            float fsx = 0.0f, fsy = 0.0f;
            {
                const float my_rs = r + safety;
                // all_params_equal: every row's stored velocity (the reference evaluates all pairs
                // before any in-place refresh); else: my refreshed velocity, partner j<i refreshed,
                // j>i stored (prange == range order; identical from the 2nd fused substep on)
                const float vix = PEQ ? vx : cvx, viy = PEQ ? vy : cvy;
                const float4* pv = lds_pv + cur * T + base;
                const float2* vr = lds_vr + cur * T + base;
                const float* rs = lds_rs + base;
                for (int j = 0; j < rows; ++j) {
                    if (j == row) continue;
                    const float4 q = pv[j];
                    float vjx = q.z, vjy = q.w;
                    if constexpr (!PEQ && HEADED > 0) {
                        if (j < row) { const float2 t2 = vr[j]; vjx = t2.x; vjy = t2.y; }
                    }
                    pair_force<SOC>(sp, px, py, vix, viy, q.x, q.y, vjx, vjy, my_rs + rs[j], fsx, fsy);
                }
            }
            // -- total force, body frame, torque  :262-271, :165-182
            const float fix = fdx + fox + fsx, fiy = fdy + foy + fsy;
            float gfx = fix, gfy = fiy, torque = 0.0f, inertia = 1.0f;
            if constexpr (HEADED > 0) {
                inertia = 0.5f * m * r * r;
                const float drx = (HEADED == 1) ? fdx : fix, dry = (HEADED == 1) ? fdy : fiy;
                const float fnorm = norm2(drx, dry);
                const float k_theta = inertia * klam * fnorm;
                const float k_omega = inertia * (1.0f + alpha) * sqrtf((klam * fnorm) / alpha);
                torque = -k_theta * bound_angle(th - atan2f(dry, drx)) - k_omega * om;
                gfx = fix * c + fiy * s;
                gfy = ko * ((fox + fsx) * (-s) + (foy + fsy) * c) - kd * bvy;
            }
            // -- explicit Euler, :273-283 (position uses the velocity stored in the incoming row)
            const float in_vx = cvx, in_vy = cvy; // what the reference leaves in agents_state[i,3:5]
            px += vx * dt; py += vy * dt;
            if constexpr (HEADED > 0) {
                th = bound_angle(th + om * dt);
                bvx += (gfx / m) * dt; bvy += (gfy / m) * dt;
                const float nb = norm2(bvx, bvy);
                if (nb > vd) { bvx = (bvx / nb) * vd; bvy = (bvy / nb) * vd; }
                om += (torque / inertia) * dt;
                const float c2 = cosf(th), s2 = sinf(th);
                vx = c2 * bvx + (-s2) * bvy;
                vy = s2 * bvx + c2 * bvy;
            } else {
                vx += (gfx / m) * dt; vy += (gfy / m) * dt;
                const float nb = norm2(vx, vy);
                if (nb > vd) { vx = (vx / nb) * vd; vy = (vy / nb) * vd; }
            }
            if ((a.mode & M_MUTATE_INPUT) && sub == 0) {
                float* si = a.Sin + sidx * a.in_as;
                if (HEADED > 0) { si[3 * a.in_fs] = in_vx; si[4 * a.in_fs] = in_vy; }
                si[10 * a.in_fs] = gx; si[11 * a.in_fs] = gy;
            }
            lds_pv[nxt * T + tid] = make_float4(px, py, vx, vy);
            lds_vr[nxt * T + tid] = make_float2(vx, vy);
        } else if (is_robot) {
            // the robot's move of the NEXT substep happens before that substep's update_humans
            if (robot_moves && sub + 1 < a.nsub) robot_step();
            lds_pv[nxt * T + tid] = make_float4(px, py, vx, vy);
            lds_vr[nxt * T + tid] = make_float2(vx, vy);
        }
        __syncthreads();
        // -- parallel-traffic respawn, motion_model_manager.py:407-422 (sequential inside a world)
        if (a.flags & CS_RESPAWN) {
            const int flag = (human && respawn_here && norm2(px - g0x, py - g0y) < 3.0f) ? 1 : 0;
            if (__syncthreads_or(flag)) {
                lds_flag[tid] = flag;
                lds_g0x[tid] = g0x;
                __syncthreads();
                if (valid && row == 0) {
                    float4* pvn = lds_pv + nxt * T + base;
                    const float* rs = lds_rs + base;
                    for (int i = 0; i < n; ++i) {
                        if (!lds_flag[base + i]) continue;
                        float mx = pvn[0].x, mr = rs[0];
                        for (int j = 1; j < n; ++j) {
                            mx = fmaxf(mx, pvn[j].x);
                            mr = fmaxf(mr, rs[j]);
                        }
                        if (robot_row) { // consider_robot: the robot where it stands in THIS substep
                            mx = fmaxf(mx, lds_pv[cur * T + base + n].x);
                            mr = fmaxf(mr, rs[n]);
                        }
                        float4 q = pvn[i];
                        q.x = fmaxf(mx + mr * 2.0f, a.bx);
                        q.y = (q.y >= 0.0f) ? fminf(q.y, a.by) : fmaxf(q.y, -a.by);
                        pvn[i] = q;
                    }
                }
                __syncthreads();
                if (flag) {
                    const float4 q = lds_pv[nxt * T + tid];
                    px = q.x; py = q.y;
                    g0y = py;               // human.set_goals([[goals[0][0], position[1]]])   :418
                    bvy = g0x; om = g0y;    // states[i,6:8] = goal  (reference writes cols 6:8) :421
                    for (int g = 0; g < a.G; ++g) { gi[2 * g] = g0x; gi[2 * g + 1] = g0y; } //   :422
                }
            }
        }
        cur = nxt;
    }

    // ---- epilogue ---------------------------------------------------------------------------
    if (a.mode & M_PEEK) {
        if (human) {
            float* o = a.peek_out + ((long)w * n + row) * 8;
            o[0] = px; o[1] = py; o[2] = th; o[3] = vx; o[4] = vy; o[5] = om; o[6] = gx; o[7] = gy;
        }
        return;
    }
    if (valid) {
        float* o = a.Sout + sidx * a.out_as;
        const long fs = a.out_fs;
        if (human || is_robot) {
            o[0] = px; o[fs] = py; o[2 * fs] = th; o[3 * fs] = vx; o[4 * fs] = vy; o[5 * fs] = bvx;
            o[6 * fs] = bvy; o[7 * fs] = om; o[10 * fs] = gx; o[11 * fs] = gy;
            if (a.Sout != a.Sin || is_robot) { o[8 * fs] = r; o[9 * fs] = m; o[12 * fs] = vd; }
        }
        if (is_robot && robot_moves && a.robot != nullptr) {
            float* rb = a.robot + (long)w * 13;
            rb[0] = px; rb[1] = py; rb[2] = th; rb[3] = vx; rb[4] = vy;
        }
        // invisible robot: advanced by the lane of row 0 (it does not interact with the crowd)
        if (!robot_row && row == 0 && robot_moves && a.robot != nullptr) {
            float* rb = a.robot + (long)w * 13;
            float qx = rb[0], qy = rb[1], qt = rb[2], qvx = rb[3], qvy = rb[4];
            for (int sub = 0; sub < a.nsub; ++sub) {
                if (a.flags & CS_ROBOT_UNICYCLE) {
                    const float c = cosf(qt + ay), s = sinf(qt + ay);
                    qx += c * ax * dt; qy += s * ax * dt;
                    qt = fmodf(qt + ay, 6.283185307179586f);
                    if (qt < 0) qt += 6.283185307179586f;
                    qvx = cosf(qt) * ax; qvy = sinf(qt) * ax;
                } else {
                    qx += ax * dt; qy += ay * dt; qvx = ax; qvy = ay;
                }
            }
            rb[0] = qx; rb[1] = qy; rb[2] = qt; rb[3] = qvx; rb[4] = qvy;
        }
    }
}

// ------------------------------------------------------------------------------------------
// collision / reward: social_nav_sim.py:949-1029, utils.py:22-36.  One lane per world.
// ------------------------------------------------------------------------------------------
__global__ void k_collision_reward(int W, int n, int rows, const float* S, long as, long fs, const float* robot,
                                   const float* action, float T, const float* gtime, float time_limit,
                                   float success_reward, float collision_penalty, float discomfort_dist,
                                   float discomfort_factor, float* out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    const float* rb = robot + (long)w * 13;
    const float rpx = rb[0], rpy = rb[1], rr = rb[8], rgx = rb[10], rgy = rb[11];
    const float ax = action[(long)w * 2], ay = action[(long)w * 2 + 1];
    float dmin = INFINITY;
    int collision = 0;
    for (int i = 0; i < n; ++i) {
        const float* s = S + ((long)w * rows + i) * as;
        const float x1 = s[0] - rpx, y1 = s[fs] - rpy;
        const float x2 = x1 + (s[3 * fs] - ax) * T, y2 = y1 + (s[4 * fs] - ay) * T;
        const float dx = x2 - x1, dy = y2 - y1;
        float d;
        if (dx == 0.0f && dy == 0.0f) d = norm2(0.0f - x1, 0.0f - y1);
        else {
            float u = ((0.0f - x1) * dx + (0.0f - y1) * dy) / (dx * dx + dy * dy);
            if (u > 1.0f) u = 1.0f; else if (u < 0.0f) u = 0.0f;
            d = norm2(x1 + u * dx, y1 + u * dy);
        }
        const float closest = d - s[8 * fs] - rr;
        if (closest < 0.0f) { collision = 1; break; }
        else if (closest < dmin) dmin = closest;
    }
    const float ex = rpx + ax * T, ey = rpy + ay * T;
    const int reaching = norm2(ex - rgx, ey - rgy) < rr;
    float reward = 0.0f; int term = 0, trunc = 0, info = 0;
    if (gtime[w] >= time_limit - 1.0f) { trunc = 1; info = 4; }
    else if (collision) { reward = collision_penalty; term = 1; info = 3; }
    else if (reaching) { reward = success_reward; term = 1; info = 2; }
    else if (dmin < discomfort_dist) { reward = (dmin - discomfort_dist) * discomfort_factor * T; info = 1; }
    float* o = out + (long)w * 7;
    o[0] = (float)collision; o[1] = dmin; o[2] = (float)reaching; o[3] = reward;
    o[4] = (float)term; o[5] = (float)trunc; o[6] = (float)info;
}

__global__ void k_transpose_state(const float* src, float* dst, long total_rows, int to_soa)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; // one lane per (row, field)
    if (i >= total_rows * 13) return;
    if (to_soa) { // dst plane-major: coalesced writes
        const long f = i / total_rows, rrow = i - f * total_rows;
        dst[i] = src[rrow * 13 + f];
    } else {
        const long rrow = i / 13, f = i - rrow * 13;
        dst[i] = src[f * total_rows + rrow];
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
using kfn = void (*)(const KArgs);

template <int MAXT>
kfn pick_kernel(int type, bool peq)
{
#define CS_CASE(SOC, HD)                                                                          \
    return peq ? (kfn)k_sfm_step<SOC, HD, true, MAXT> : (kfn)k_sfm_step<SOC, HD, false, MAXT>;
    switch (type) {
        case 0: CS_CASE(0, 0) case 1: CS_CASE(1, 0) case 2: CS_CASE(2, 0)
        case 3: CS_CASE(0, 1) case 4: CS_CASE(1, 1) case 5: CS_CASE(2, 1)
        case 6: CS_CASE(0, 2) case 7: CS_CASE(1, 2) case 8: CS_CASE(2, 2)
    }
#undef CS_CASE
    return nullptr;
}

struct Geometry { int grid, block, wpb; };

int geometry(const cs_worlds* w, Geometry& g)
{
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    if (rows <= 0 || rows > 1024) return fail(CS_ERR_ARG, "rows per world must be in 1..1024");
    if (rows <= 64) { g.block = 64; g.wpb = 64 / rows; }
    else { g.block = ((rows + 63) / 64) * 64; g.wpb = 1; }
    g.grid = (w->W + g.wpb - 1) / g.wpb;
    return CS_OK;
}

int check_worlds(const cs_worlds* w)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (w->type < 0 || w->type > 8) return fail(CS_ERR_TYPE, "Type " + std::to_string(w->type) + " does not exist for this implementation");
    if (w->W <= 0 || w->n <= 0 || w->G <= 0) return fail(CS_ERR_ARG, "W, n, G must be positive");
    if (!w->d_state || !w->d_goals || !w->d_params || !w->d_safety) return fail(CS_ERR_ARG, "null device buffer in cs_worlds");
    if (w->O < 0 || (w->O > 0 && (!w->d_obstacles || w->Smax <= 0))) return fail(CS_ERR_ARG, "bad obstacle description");
    if (w->layout != CS_LAYOUT_AOS && w->layout != CS_LAYOUT_SOA) return fail(CS_ERR_ARG, "bad layout");
    return CS_OK;
}

void strides(const cs_worlds* w, int rows, long& as, long& fs)
{
    if (w->layout == CS_LAYOUT_AOS) { as = 13; fs = 1; }
    else { as = 1; fs = (long)w->W * rows; }
}

int launch_step(const cs_worlds* w, float dt, int nsub, int mode, float* d_out, const float* d_action,
                float* d_peek, hipStream_t stream)
{
    int rc = check_worlds(w);
    if (rc) return rc;
    Geometry g;
    rc = geometry(w, g);
    if (rc) return rc;
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = w->W; a.n = w->n; a.rows = rows; a.G = w->G; a.O = w->O; a.Smax = w->Smax;
    a.type = w->type; a.flags = w->flags; a.mode = mode; a.nsub = nsub; a.wpb = g.wpb; a.dt = dt;
    a.Sin = w->d_state; a.Sout = d_out ? d_out : w->d_state;
    strides(w, rows, a.in_as, a.in_fs);
    a.out_as = a.in_as; a.out_fs = a.in_fs;
    a.goals = w->d_goals; a.params = w->d_params; a.safety = w->d_safety; a.obstacles = w->d_obstacles;
    a.robot = w->d_robot; a.action = d_action; a.peek_out = d_peek;
    a.bx = w->respawn_bound_x; a.by = w->respawn_bound_y;
    a.world_flags = w->d_world_flags;
    const bool peq = (w->flags & CS_ALL_PARAMS_EQUAL) != 0;
    kfn fn = (g.block == 64) ? pick_kernel<64>(w->type, peq) : pick_kernel<1024>(w->type, peq);
    const size_t shmem = (size_t)g.block * (2 * sizeof(float4) + 2 * sizeof(float2) + 3 * sizeof(float));
    hipLaunchKernelGGL(fn, dim3(g.grid), dim3(g.block), shmem, stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* cs_last_error(void) { return g_err.c_str(); }
int cs_abi_version(void) { return 1; }

int cs_device_count(int* count)
{
    if (!count) return fail(CS_ERR_ARG, "null count");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return fail(CS_ERR_NO_DEVICE, hipGetErrorString(e)); }
    *count = c;
    return CS_OK;
}

int cs_set_device(int device) { HIP_TRY(hipSetDevice(device)); return CS_OK; }

int cs_device_name(int device, char* buf, size_t buflen)
{
    if (!buf || buflen == 0) return fail(CS_ERR_ARG, "null buffer");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    std::snprintf(buf, buflen, "%s (%s)", p.name, p.gcnArchName);
    return CS_OK;
}

int cs_malloc(void** d_ptr, size_t bytes) { if (!d_ptr) return fail(CS_ERR_ARG, "null out pointer"); HIP_TRY(hipMalloc(d_ptr, bytes)); return CS_OK; }
int cs_free(void* d_ptr) { HIP_TRY(hipFree(d_ptr)); return CS_OK; }
int cs_memcpy_h2d(void* d, const void* h, size_t bytes, void* s) { HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)s)); if (!s) HIP_TRY(hipStreamSynchronize(nullptr)); return CS_OK; }
int cs_memcpy_d2h(void* h, const void* d, size_t bytes, void* s) { HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)s)); HIP_TRY(hipStreamSynchronize((hipStream_t)s)); return CS_OK; }
int cs_memcpy_d2d(void* dd, const void* ds, size_t bytes, void* s) { HIP_TRY(hipMemcpyAsync(dd, ds, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s)); return CS_OK; }
int cs_memset(void* d, int value, size_t bytes, void* s) { HIP_TRY(hipMemsetAsync(d, value, bytes, (hipStream_t)s)); return CS_OK; }
int cs_stream_create(void** s) { if (!s) return fail(CS_ERR_ARG, "null out pointer"); hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); *s = st; return CS_OK; }
int cs_stream_destroy(void* s) { HIP_TRY(hipStreamDestroy((hipStream_t)s)); return CS_OK; }
int cs_stream_sync(void* s) { HIP_TRY(hipStreamSynchronize((hipStream_t)s)); return CS_OK; }
int cs_event_create(void** e) { if (!e) return fail(CS_ERR_ARG, "null out pointer"); hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); *e = ev; return CS_OK; }
int cs_event_destroy(void* e) { HIP_TRY(hipEventDestroy((hipEvent_t)e)); return CS_OK; }
int cs_event_record(void* e, void* s) { HIP_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); return CS_OK; }
int cs_event_elapsed_ms(void* a, void* b, float* ms)
{
    if (!ms) return fail(CS_ERR_ARG, "null ms");
    HIP_TRY(hipEventSynchronize((hipEvent_t)b));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return CS_OK;
}

int cs_update_humans_parallel(const cs_worlds* w, float dt, float* d_out, void* stream)
{
    if (!w || !d_out) return fail(CS_ERR_ARG, "null argument");
    int mode = M_COMMIT_GOALS;
    if (d_out != w->d_state) mode |= M_MUTATE_INPUT;
    cs_worlds ww = *w;
    ww.flags &= ~CS_RESPAWN;
    ww.d_robot = nullptr;
    return launch_step(&ww, dt, 1, mode, d_out, nullptr, nullptr, (hipStream_t)stream);
}

int cs_step(const cs_worlds* w, float dt, int n_substeps, const float* d_action, void* stream)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (n_substeps <= 0) return fail(CS_ERR_ARG, "n_substeps must be positive");
    if (d_action && !w->d_robot) return fail(CS_ERR_ARG, "robot action given but cs_worlds.d_robot is null");
    int mode = M_COMMIT_GOALS;
    if ((w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
    return launch_step(w, dt, n_substeps, mode, nullptr, d_action, nullptr, (hipStream_t)stream);
}

int cs_peek(const cs_worlds* w, float dt, float* d_next, void* stream)
{
    if (!w || !d_next) return fail(CS_ERR_ARG, "null argument");
    int mode = M_PEEK;
    if ((w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
    cs_worlds ww = *w;
    ww.flags &= ~CS_RESPAWN; // update_humans(0, dt, post_update=False), motion_model_manager.py:705
    return launch_step(&ww, dt, 1, mode, nullptr, nullptr, d_next, (hipStream_t)stream);
}

int cs_collision_reward(const cs_worlds* w, const float* d_action, float T, const float* d_global_time,
                        const float* reward_cfg, float* d_out, void* stream)
{
    int rc = check_worlds(w);
    if (rc) return rc;
    if (!d_action || !d_global_time || !reward_cfg || !d_out || !w->d_robot) return fail(CS_ERR_ARG, "null argument");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    long as, fs;
    strides(w, rows, as, fs);
    const int block = 64, grid = (w->W + block - 1) / block;
    hipLaunchKernelGGL(k_collision_reward, dim3(grid), dim3(block), 0, (hipStream_t)stream, w->W, w->n, rows,
                       (const float*)w->d_state, as, fs, (const float*)w->d_robot, d_action, T, d_global_time,
                       reward_cfg[0], reward_cfg[1], reward_cfg[2], reward_cfg[3], reward_cfg[4], d_out);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_state_aos_to_soa(const float* d_aos, float* d_soa, int W, int rows, void* stream)
{
    if (!d_aos || !d_soa || W <= 0 || rows <= 0) return fail(CS_ERR_ARG, "bad argument");
    const long total = (long)W * rows;
    const int block = 256;
    const long grid = (total * 13 + block - 1) / block;
    hipLaunchKernelGGL(k_transpose_state, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream, d_aos, d_soa, total, 1);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_state_soa_to_aos(const float* d_soa, float* d_aos, int W, int rows, void* stream)
{
    if (!d_aos || !d_soa || W <= 0 || rows <= 0) return fail(CS_ERR_ARG, "bad argument");
    const long total = (long)W * rows;
    const int block = 256;
    const long grid = (total * 13 + block - 1) / block;
    hipLaunchKernelGGL(k_transpose_state, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream, d_soa, d_aos, total, 0);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_launch_geometry(const cs_worlds* w, int* grid, int* block, int* worlds_per_block)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    Geometry g;
    int rc = geometry(w, g);
    if (rc) return rc;
    if (grid) *grid = g.grid;
    if (block) *block = g.block;
    if (worlds_per_block) *worlds_per_block = g.wpb;
    return CS_OK;
}

} // extern "C"
