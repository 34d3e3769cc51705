// social_momentum.hip -- the social-momentum crowd model ("remaining models") for gfx950.
//
// Restates the social-momentum branch of MotionModelManager.update_humans
//   /root/reference/social_gym/src/motion_model_manager.py:395-404   per human: update_goals ; filter ; reactive ; optimise,
//                                                                     then for all: p += v dt ; v = chosen action
//   :66-70     update_goals (strict <, whole list rotated)           :247-251  action set: n_actions = 20 unit vectors x v_d
//   /root/reference/social_gym/src/social_momentum.py:10-27          filter_action_set_for_collisions
//   :29-44     update_reactive_agents (angle <= pi: everybody the angle is defined for)
//   :46-75     optimize_momentum (efficiency 1/|goal - next| + LAMBDA x weighted pairwise angular momentum,
//                                 zeroed by the first pair whose momentum would change sign)
//   :407-422   parallel-traffic respawn
// for W worlds at once: lane = agent row, floor(64/rows) worlds per wavefront, (x, y, vx, vy) and radius + safety space of
// every row in LDS (the partners of a lane are read as wave-wide broadcasts), all substeps of a launch fused.
// The model is a discrete arg-max over 20 actions per human: O(20 N) pair terms per human and substep, float32.
// Entities = humans, plus the robot as the last row when it is visible (the robot row holds the TRUE robot, moved by the
// action before every substep like social_nav_gym.py:240-245 does).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "common.h"
#include "robotstep.h"
#include "respawnx.h"

#pragma clang fp contract(off)

namespace {

using csimpl::fail;

constexpr int SM_MAX_ACTIONS = 64;
constexpr float SM_LAMBDA = 0.11f; // social_momentum.py:8

struct MArgs {
    int W, n, rows, G, flags, nsub, wpb, A;
    float dt, bx, by;
    float* S; long as, fs;
    float* goals;
    const float* safety;
    float* robot;
    const float* action;
    float* peek_out;
    const int* world_flags;
};

__global__ __launch_bounds__(64) void k_sm_step(const MArgs a)
{
    __shared__ float4 lds_pv[2][64];   // px, py, vx, vy of every row, double-buffered over substeps
    __shared__ float lds_rs[64];       // radius + safety space
    __shared__ float2 lds_unit[SM_MAX_ACTIONS]; // unit action directions
    const int tid = threadIdx.x, T = 64;
    const int rows = a.rows, n = a.n;
    const int lw = tid / rows, row = tid - lw * rows;
    const int w = blockIdx.x * a.wpb + lw;
    const bool valid = lw < a.wpb && w < a.W;
    const bool robot_row = (a.flags & CS_ROBOT_ROW) != 0;
    const bool human = valid && row < n;
    const bool is_robot = valid && robot_row && row == n;
    const int base = lw * rows;
    const float dt = a.dt;

    // action directions: [cos, sin]((2 pi / A) k) evaluated in double like the reference (:249-250), rounded once
    if (tid < a.A) {
        const double ang = ((2.0 * 3.141592653589793) / (double)a.A) * (double)tid;
        lds_unit[tid] = make_float2((float)cos(ang), (float)sin(ang));
    }

    float px = 0, py = 0, th = 0, vx = 0, vy = 0, om = 0, r = 0, vd = 0, rs = 0;
    float* srow = nullptr;
    if (valid) {
        srow = a.S + ((long)w * rows + row) * a.as;
        const long fs = a.fs;
        px = srow[0]; py = srow[fs]; th = srow[2 * fs]; vx = srow[3 * fs]; vy = srow[4 * fs]; om = srow[7 * fs];
        r = srow[8 * fs]; vd = srow[12 * fs];
        if (is_robot && a.robot != nullptr) { // the true robot
            const float* rb = a.robot + (long)w * 13;
            px = rb[0]; py = rb[1]; th = rb[2]; vx = rb[3]; vy = rb[4]; r = rb[8];
        }
        rs = r + a.safety[(long)w * rows + row];
    }
    float g0x = 0, g0y = 0;
    float* gi = nullptr;
    if (human) { gi = a.goals + ((long)w * n + row) * a.G * 2; g0x = gi[0]; g0y = gi[1]; }
    const bool respawn_here = valid && (a.world_flags == nullptr || (a.world_flags[w] & 1));
    const bool robot_moves = a.action != nullptr;
    float ax = 0, ay = 0;
    if (valid && robot_moves) { ax = a.action[(long)w * 2]; ay = a.action[(long)w * 2 + 1]; }

    if (is_robot && robot_moves) csimpl::robot_action_step(a.flags, px, py, th, vx, vy, ax, ay, dt); // robot.step before the first update (robotstep.h)
    if (valid) { lds_pv[0][tid] = make_float4(px, py, vx, vy); lds_rs[tid] = rs; }
    __syncthreads();

    int cur = 0;
    for (int sub = 0; sub < a.nsub; ++sub) {
        const int nxt = cur ^ 1;
        if (human) {
            // ---- update_goals (:66-70): strict <, on the incoming position; the whole non-NaN prefix rotates
            {
                const float ddx = g0x - px, ddy = g0y - py;
                if (sqrtf(ddx * ddx + ddy * ddy) < r) {
                    int k = a.G;
                    for (int g = 0; g < a.G; ++g) if (isnan(gi[2 * g])) { k = g; break; }
                    if (a.peek_out == nullptr) {
                        const float r0 = gi[0], r1 = gi[1];
                        for (int g = 0; g + 1 < k; ++g) { gi[2 * g] = gi[2 * g + 2]; gi[2 * g + 1] = gi[2 * g + 3]; }
                        if (k > 0) { gi[2 * (k - 1)] = r0; gi[2 * (k - 1) + 1] = r1; }
                        g0x = gi[0]; g0y = gi[1];
                    } else if (k > 1) { g0x = gi[2]; g0y = gi[3]; }
                }
            }
            const float4* pv = &lds_pv[cur][base];
            const float* rr = &lds_rs[base];
            // ---- reactive agents and their weights (:29-44, :49-52): everybody whose bearing angle is defined
            const float speed2 = vx * vx + vy * vy;
            float sumw = 0.0f;
            for (int j = 0; j < rows; ++j) {
                const float4 q = pv[j];
                const float dx = q.x - px, dy = q.y - py;
                const float d2 = dx * dx + dy * dy;
                const bool react = (j != row) && (speed2 == 0.0f || d2 > 0.0f); // NaN angle (coincident) -> not reactive
                sumw += react ? 1.0f / sqrtf(d2) : 0.0f;
            }
            // ---- arg-max over the collision-free actions, first maximum wins (:54-75)
            float best = -100000.0f, bax = 0.0f, bay = 0.0f;
            for (int k = 0; k < a.A; ++k) {
                const float2 u = lds_unit[k];
                const float acx = u.x * vd, acy = u.y * vd;
                const float nix = px + acx * dt, niy = py + acy * dt;
                bool free_ = true, broke = false;
                float mom = 0.0f;
                for (int j = 0; j < rows; ++j) {
                    const bool other_ = j != row; // predicate, not a branch: every lane skips a different row
                    const float4 q = pv[j];
                    // filter (:19-25): the other's constant-velocity next position against mine under this action
                    const float ex = (q.x + q.z * dt) - nix, ey = (q.y + q.w * dt) - niy;
                    const float thr = rs + rr[j];
                    free_ = free_ && !(other_ && sqrtf(ex * ex + ey * ey) < thr);
                    // pairwise angular momentum about the pair's centre of mass (:62-70)
                    const float dx = q.x - px, dy = q.y - py;
                    const float d2 = dx * dx + dy * dy;
                    const bool react = other_ && (speed2 == 0.0f || d2 > 0.0f);
                    const float cx = (px + q.x) / 2.0f, cy = (py + q.y) / 2.0f;
                    const float prx = px - cx, pry = py - cy, phx = q.x - cx, phy = q.y - cy;
                    const float other = phx * q.w - phy * q.z;
                    const float curm = prx * vy - pry * vx + other;
                    const float expm = prx * acy - pry * acx + other;
                    const bool keep = curm * expm > 0.0f;
                    const float term = (1.0f / sqrtf(d2)) * expm;
                    mom = (react && !broke) ? (keep ? mom + term : 0.0f) : mom;
                    broke = broke || (react && !keep);
                }
                const float gx_ = g0x - nix, gy_ = g0y - niy;
                float rew = 1.0f / sqrtf(gx_ * gx_ + gy_ * gy_);
                if (mom != 0.0f) rew = rew + SM_LAMBDA * (mom / sumw); // weights normalised by their sum (:52)
                if (free_ && rew > best) { best = rew; bax = acx; bay = acy; }
            }
            // ---- p += v dt with the incoming velocity, then v = chosen action (:401-403); no free action -> (0, 0)
            px += vx * dt; py += vy * dt;
            vx = bax; vy = bay;
            lds_pv[nxt][tid] = make_float4(px, py, vx, vy);
        } else if (is_robot) {
            if (robot_moves && sub + 1 < a.nsub) csimpl::robot_action_step(a.flags, px, py, th, vx, vy, ax, ay, dt); // next substep's robot.step
            lds_pv[nxt][tid] = make_float4(px, py, vx, vy);
        }
        __syncthreads();
        // ---- parallel-traffic respawn (:407-422): flagged humans of a world in index order, each behind everybody
        if (a.flags & CS_RESPAWN) {
            const float rdx = px - g0x, rdy = py - g0y;
            const bool flag = human && respawn_here && sqrtf(rdx * rdx + rdy * rdy) < 3.0f;
            if (flag) {
                const unsigned long long fm = __builtin_amdgcn_ballot_w64(true);
                const unsigned long long wm = (rows >= 64 ? ~0ull : ((1ull << rows) - 1ull)) << base;
                const int c = __builtin_popcountll(fm & wm & ((1ull << tid) - 1ull));
                const float4* pvn = &lds_pv[nxt][base];
                float mx = pvn[0].x, mr = lds_rs[base];
                for (int j = 1; j < n; ++j) { mx = fmaxf(mx, pvn[j].x); mr = fmaxf(mr, lds_rs[base + j]); }
                if (robot_row) { mx = fmaxf(mx, lds_pv[cur][base + n].x); mr = fmaxf(mr, lds_rs[base + n]); }
                px = csimpl::respawn_x(mx, mr, a.bx, c);   // (respawnx.h: the reference's float64 sum, rounded once)
                py = (py >= 0.0f) ? fminf(py, a.by) : fmaxf(py, -a.by);
                lds_pv[nxt][tid] = make_float4(px, py, vx, vy);
                g0y = py;
                if (a.peek_out == nullptr) for (int g = 0; g < a.G; ++g) { gi[2 * g] = g0x; gi[2 * g + 1] = g0y; }
            }
            __syncthreads();
        }
        cur = nxt;
    }

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
        if (human) { srow[10 * fs] = g0x; srow[11 * fs] = g0y; }
        if (is_robot && robot_moves && a.robot != nullptr) {
            float* rb = a.robot + (long)w * 13;
            rb[0] = px; rb[1] = py; rb[2] = th; rb[3] = vx; rb[4] = vy;
        }
        if (!robot_row && row == 0 && robot_moves && a.robot != nullptr) { // invisible robot: advanced by the lane of row 0
            float* rb = a.robot + (long)w * 13;
            float qx = rb[0], qy = rb[1], qt = rb[2], qvx = rb[3], qvy = rb[4];
            for (int sub = 0; sub < a.nsub; ++sub) csimpl::robot_action_step(a.flags, qx, qy, qt, qvx, qvy, ax, ay, dt);
            rb[0] = qx; rb[1] = qy; rb[2] = qt; rb[3] = qvx; rb[4] = qvy;
        }
    }
}

} // namespace

namespace csimpl {

int social_momentum_launch(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float* d_peek, hipStream_t stream)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (w->W <= 0 || w->n <= 0 || w->G <= 0) return fail(CS_ERR_ARG, "W, n, G must be positive");
    if (!w->d_state || !w->d_goals || !w->d_safety) return fail(CS_ERR_ARG, "null device buffer in cs_worlds");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    if (rows > 64) return fail(CS_ERR_ARG, "the social-momentum step supports up to 64 rows per world");
    const int A = w->sm_n_actions > 0 ? w->sm_n_actions : 20; // motion_model_manager.py:249
    if (A > SM_MAX_ACTIONS) return fail(CS_ERR_ARG, "sm_n_actions must be <= 64");
    MArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = w->W; a.n = w->n; a.rows = rows; a.G = w->G; a.flags = w->flags; a.nsub = n_substeps;
    a.wpb = 64 / rows; a.A = A;
    a.dt = dt; a.bx = w->respawn_bound_x; a.by = w->respawn_bound_y;
    a.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; } else { a.as = 1; a.fs = (long)w->W * rows; }
    a.goals = w->d_goals; a.safety = w->d_safety; a.robot = w->d_robot; a.action = d_action;
    a.peek_out = d_peek; a.world_flags = w->d_world_flags;
    if (d_peek) a.flags &= ~CS_RESPAWN;
    const int grid = (w->W + a.wpb - 1) / a.wpb;
    hipLaunchKernelGGL(k_sm_step, dim3(grid), dim3(64), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

} // namespace csimpl
