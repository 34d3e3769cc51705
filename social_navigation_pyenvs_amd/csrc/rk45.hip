// rk45.hip -- RK45 integration of the crowd, W worlds at once.
//
// Restates MotionModelManager(runge_kutta=True).update_humans(t, dt)
//   /root/reference/social_gym/src/motion_model_manager.py:374-384   solve_ivp(f_rk45_*, (t, t + dt), y0, method='RK45')
//   :500-550   f_rk45_headed / f_rk45_not_headed: the right-hand side writes the trial state into the agents (:88-103: speed
//              clamp, angle wrap), runs compute_forces (:437-460: goal switch at the TRIAL position, wall closest points,
//              linear velocity of headed agents, social forces) and returns [v, F/m] or [R bv, omega, F_body/m, torque/I]
//   forces.py:9-16, 27-53, 63-151, 153-218, 279-290   the SINGLE-AGENT force functions (stale desired force within one radius
//              of the goal, Guo's obstacle force not averaged, last-nearest wall point) and, when all humans share their
//              parameters, compute_all_social_forces: the pair force of the LOWER index mirrored onto the higher one
// and scipy's explicit Dormand-Prince pair with its step-size control (scipy/integrate/_ivp/rk.py RK45 + common.py
// select_initial_step, version 1.15.3: rtol 1e-3, atol 1e-6, SAFETY 0.9, factors 0.2 .. 10, RMS error norm over the whole
// world's state vector).  One block of one wavefront per world (rows <= 64): lane = human, trial rows exchanged through LDS,
// the error norm is a wavefront sum, so every lane takes the same accept / reject decision.  float32; not a hot path.
// gfx950 only.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>

#include "common.h"
#include "crowdstep.h"
#include "robot_model.h"

namespace {

using csimpl::fail;

struct KArgsRk {
    int W, n, rows, G, O, Smax, type, flags;
    float dt;
    float* S; long as, fs;
    float* goals;
    const float* params;
    const float* safety;
    const float* obstacles;
    float* memory;   // [W][n][2]
    int* nfev;       // [W] or null
    // complete_rk45_simulation (motion_model_manager.py:461-498): ONE solve over [0, t_final] with the solution at the n_eval times
    // k * eval_dt through scipy's dense output; dense [W][n_eval][n][4 | 6] (raw solution components), null for update_humans
    float t_final, eval_dt;
    int n_eval;
    float* dense;
};

constexpr float PI_F = 3.14159265358979323846f;
constexpr float TWO_PI_F = 6.28318530717958647692f;

__device__ __forceinline__ float bound_angle(float a)   // utils.py:7-13
{
    if (a >= TWO_PI_F) a = fmodf(a, TWO_PI_F);
    if (a <= -TWO_PI_F) a = fmodf(a, TWO_PI_F);
    if (a > PI_F) a -= TWO_PI_F;
    if (a < -PI_F) a += TWO_PI_F;
    return a;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// compute_pairwise_social_force(kind, agent1, agent2) (forces.py:63-128): force on agent 1; r1 / r2 = radius + safety_space
__device__ __forceinline__ void pair_force(int kind, const float* P, float p1x, float p1y, float v1x, float v1y, float r1, float p2x,
                                           float p2y, float v2x, float v2y, float r2, float& fx, float& fy)
{
    const float dx = p1x - p2x, dy = p1y - p2y;
    const float dn = sqrtf(dx * dx + dy * dy);
    const float nx = dx / dn, ny = dy / dn;
    const float rd = r1 + r2 - dn;
    const float comp = fmaxf(0.0f, rd);
    if (kind == 2) {
        const float ivx = P[12] * (v1x - v2x) - nx, ivy = P[12] * (v1y - v2y) - ny;
        const float inorm = sqrtf(ivx * ivx + ivy * ivy);
        const float ix = ivx / inorm, iy = ivy / inorm;
        const float th = bound_angle(atan2f(ny, nx) - atan2f(iy, ix) + PI_F);
        const float k = (th > 0.0f) ? 1.0f : ((th < 0.0f) ? -1.0f : 0.0f);
        const float hx = -iy, hy = ix;
        const float F = P[13] * inorm;
        const float dvh = (v2x - v1x) * hx + (v2y - v1y) * hy;
        const float e0 = P[9] * expf(-dn / F);
        const float t1 = P[15] * F * th, t2 = P[14] * F * th;
        const float e1 = expf(-(t1 * t1)), e2 = k * expf(-(t2 * t2));
        fx = -(e0 * (e1 * ix + e2 * hx) + P[10] * comp * ix + P[11] * comp * dvh * hx);
        fy = -(e0 * (e1 * iy + e2 * hy) + P[10] * comp * iy + P[11] * comp * dvh * hy);
    } else {
        const float tx = -ny, ty = nx;
        const float dv = (v2x - v1x) * tx + (v2y - v1y) * ty;
        const float fn = P[1] * expf(rd / P[3]) + P[10] * comp;
        float ft = P[11] * comp * dv;
        if (kind == 1) ft += P[5] * expf(rd / P[7]);
        fx = fn * nx + ft * tx;
        fy = fn * ny + ft * ty;
    }
}

// ---- scipy's explicit Dormand-Prince pair with its step-size control, shared by the crowd's and the robot's solve --------------
// (scipy/integrate/_ivp/rk.py RK45 + common.py select_initial_step, version 1.15.3: rtol 1e-3, atol 1e-6, SAFETY 0.9, factors
// 0.2 .. 10).  rhs(y_trial, out): the right-hand side (it does not depend on t); rms(v, scale): the RMS norm over the whole solve's
// state vector, identical in every lane; on_accept(t_old, t_new, h, y_old, K): called for every accepted step before K[0] takes
// f(t_new, y_new) -- the dense output of complete_rk45_simulation hangs there.  On return y = y(T).
template <int NS, class Rhs, class Rms, class OnAccept>
__device__ __forceinline__ void rk45_solve(float (&y)[NS], float T, Rhs&& rhs, Rms&& rms, OnAccept&& on_accept)
{
    const float rtol = 1e-3f, atol = 1e-6f;
    float K[7][NS];
    rhs(y, K[0]);
    // ---- select_initial_step (common.py)
    float h_abs;
    {
        float sc[NS], df[NS], y1[NS], f1[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) sc[c] = atol + fabsf(y[c]) * rtol;
        const float d0 = rms(y, sc), d1 = rms(K[0], sc);
        float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
        h0 = fminf(h0, T);
#pragma unroll
        for (int c = 0; c < NS; ++c) y1[c] = y[c] + h0 * K[0][c];
        rhs(y1, f1);
#pragma unroll
        for (int c = 0; c < NS; ++c) df[c] = f1[c] - K[0][c];
        const float d2 = rms(df, sc) / h0;
        float h1;
        if (d1 <= 1e-15f && d2 <= 1e-15f) h1 = fmaxf(1e-6f, h0 * 1e-3f);
        else h1 = powf(0.01f / fmaxf(d1, d2), 0.2f);
        h_abs = fminf(fminf(100.0f * h0, h1), T);
    }
    // ---- RK45 steps until t_bound (rk.py RungeKutta._step_impl); Dormand-Prince tableau
    constexpr float A10 = 1.0f / 5;
    constexpr float A20 = 3.0f / 40, A21 = 9.0f / 40;
    constexpr float A30 = 44.0f / 45, A31 = -56.0f / 15, A32 = 32.0f / 9;
    constexpr float A40 = 19372.0f / 6561, A41 = -25360.0f / 2187, A42 = 64448.0f / 6561, A43 = -212.0f / 729;
    constexpr float A50 = 9017.0f / 3168, A51 = -355.0f / 33, A52 = 46732.0f / 5247, A53 = 49.0f / 176, A54 = -5103.0f / 18656;
    constexpr float B0 = 35.0f / 384, B2 = 500.0f / 1113, B3 = 125.0f / 192, B4 = -2187.0f / 6784, B5 = 11.0f / 84;
    constexpr float E0 = -71.0f / 57600, E2 = 71.0f / 16695, E3 = -71.0f / 1920, E4 = 17253.0f / 339200, E5 = -22.0f / 525, E6 = 1.0f / 40;
    float t = 0.0f;
    int guard = 0;
    while (t < T && guard < 100000) {
        ++guard;
        const float min_step = 10.0f * (nextafterf(t, INFINITY) - t);
        if (h_abs < min_step) h_abs = min_step;
        bool rejected = false;
        for (;;) {
            float h = h_abs;
            float t_new = t + h;
            if (t_new - T > 0.0f) t_new = T;
            h = t_new - t;
            h_abs = fabsf(h);
            float yt[NS], ynew[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) yt[c] = y[c] + h * (A10 * K[0][c]);
            rhs(yt, K[1]);
#pragma unroll
            for (int c = 0; c < NS; ++c) yt[c] = y[c] + h * (A20 * K[0][c] + A21 * K[1][c]);
            rhs(yt, K[2]);
#pragma unroll
            for (int c = 0; c < NS; ++c) yt[c] = y[c] + h * (A30 * K[0][c] + A31 * K[1][c] + A32 * K[2][c]);
            rhs(yt, K[3]);
#pragma unroll
            for (int c = 0; c < NS; ++c) yt[c] = y[c] + h * (A40 * K[0][c] + A41 * K[1][c] + A42 * K[2][c] + A43 * K[3][c]);
            rhs(yt, K[4]);
#pragma unroll
            for (int c = 0; c < NS; ++c) yt[c] = y[c] + h * (A50 * K[0][c] + A51 * K[1][c] + A52 * K[2][c] + A53 * K[3][c] + A54 * K[4][c]);
            rhs(yt, K[5]);
#pragma unroll
            for (int c = 0; c < NS; ++c) ynew[c] = y[c] + h * (B0 * K[0][c] + B2 * K[2][c] + B3 * K[3][c] + B4 * K[4][c] + B5 * K[5][c]);
            rhs(ynew, K[6]);
            float err[NS], sc[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                sc[c] = atol + fmaxf(fabsf(y[c]), fabsf(ynew[c])) * rtol;
                err[c] = (E0 * K[0][c] + E2 * K[2][c] + E3 * K[3][c] + E4 * K[4][c] + E5 * K[5][c] + E6 * K[6][c]) * h;
            }
            const float error_norm = rms(err, sc);
            if (error_norm < 1.0f) {
                float factor = (error_norm == 0.0f) ? 10.0f : fminf(10.0f, 0.9f * powf(error_norm, -0.2f));
                if (rejected) factor = fminf(1.0f, factor);
                h_abs *= factor;
                on_accept(t, t_new, h, y, K);
                t = t_new;
#pragma unroll
                for (int c = 0; c < NS; ++c) { y[c] = ynew[c]; K[0][c] = K[6][c]; }
                break;
            }
            h_abs *= fmaxf(0.2f, 0.9f * powf(error_norm, -0.2f));
            rejected = true;
            if (h_abs < min_step || !(error_norm == error_norm)) { t = T; break; }   // TOO_SMALL_STEP / NaN: give up like a failed solve
        }
    }
}

// scipy's RkDenseOutput for RK45 (rk.py: Q = K^T P, y(t) = y_old + h Q [x, x^2, x^3, x^4], x = (t - t_old) / h): component c at x
template <int NS>
__device__ __forceinline__ float rk45_dense(const float (&K)[7][NS], int c, float y_old, float h, float x)
{
    constexpr float P[7][4] = {
        {1.0f, (float)(-8048581381.0 / 2820520608.0), (float)(8663915743.0 / 2820520608.0), (float)(-12715105075.0 / 11282082432.0)},
        {0.0f, 0.0f, 0.0f, 0.0f},
        {0.0f, (float)(131558114200.0 / 32700410799.0), (float)(-68118460800.0 / 10900136933.0), (float)(87487479700.0 / 32700410799.0)},
        {0.0f, (float)(-1754552775.0 / 470086768.0), (float)(14199869525.0 / 1410260304.0), (float)(-10690763975.0 / 1880347072.0)},
        {0.0f, (float)(127303824393.0 / 49829197408.0), (float)(-318862633887.0 / 49829197408.0), (float)(701980252875.0 / 199316789632.0)},
        {0.0f, (float)(-282668133.0 / 205662961.0), (float)(2019193451.0 / 616988883.0), (float)(-1453857185.0 / 822651844.0)},
        {0.0f, (float)(40617522.0 / 29380423.0), (float)(-110615467.0 / 29380423.0), (float)(69997945.0 / 29380423.0)}};
    float q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = 0.0f;
#pragma unroll
        for (int st = 0; st < 7; ++st) s += K[st][c] * P[st][j];
        q[j] = s;
    }
    const float x2 = x * x;
    return y_old + h * (q[0] * x + q[1] * x2 + q[2] * (x2 * x) + q[3] * (x2 * x2));
}

template <bool HEADED>
__global__ __launch_bounds__(64) void k_rk45_step(const KArgsRk a)
{
    constexpr int NS = HEADED ? 6 : 4;
    __shared__ float4 s_pv[64];   // trial x, y, vx, vy of every row
    __shared__ float s_rs[64];    // radius + safety_space
    const int w = blockIdx.x, lane = threadIdx.x;
    const int n = a.n, rows = a.rows;
    const bool human = lane < n;
    const bool all_equal = (a.flags & CS_ALL_PARAMS_EQUAL) != 0;
    const int kind = a.type % 3;
    const bool torque_new = a.type >= CS_HSFM_NEW;
    const int nent = rows;   // humans (+ the robot as the last entity)

    float* srow = a.S + ((long)w * rows + (lane < rows ? lane : 0)) * a.as;
    const long fs = a.fs;
    float y[NS] = {};
    float radius = 0.3f, mass = 1.0f, vd = 1.0f, vx = 0.0f, vy = 0.0f, th = 0.0f;
    if (lane < rows) {
        radius = srow[8 * fs]; mass = srow[9 * fs]; vd = srow[12 * fs];
        vx = srow[3 * fs]; vy = srow[4 * fs]; th = srow[2 * fs];
        if (HEADED) { y[0] = srow[0]; y[1] = srow[fs]; y[2] = th; y[3] = srow[5 * fs]; y[4] = srow[6 * fs]; y[5] = srow[7 * fs]; }
        else { y[0] = srow[0]; y[1] = srow[fs]; y[2] = vx; y[3] = vy; }
        s_rs[lane] = radius + a.safety[(long)w * rows + lane];
        s_pv[lane] = make_float4(srow[0], srow[fs], vx, vy);   // the robot row stays as it is for the whole call
    }
    const float rs = (lane < rows) ? radius + a.safety[(long)w * rows + lane] : 0.0f;
    const float* P = a.params + ((a.flags & CS_PARAMS_SHARED) ? (long)(human ? lane : 0) * 20 : ((long)w * n + (human ? lane : 0)) * 20);
    float* gi = a.goals + ((long)w * n + (human ? lane : 0)) * a.G * 2;
    float g0x = 0.0f, g0y = 0.0f;
    if (human) { g0x = gi[0]; g0y = gi[1]; }
    float* mem = a.memory + ((long)w * n + (human ? lane : 0)) * 2;
    float fdx = 0.0f, fdy = 0.0f;
    if (human) { fdx = mem[0]; fdy = mem[1]; }
    const float* ob = a.O > 0 ? a.obstacles + ((a.flags & CS_OBSTACLES_SHARED) ? 0 : (long)w * a.O * a.Smax * 4) : nullptr;
    const float inertia = 0.5f * mass * radius * radius;
    int nfev = 0;
    // agent attributes the right-hand side leaves behind (state of the LAST evaluation)
    float px = y[0], py = y[1], bvx = 0.0f, bvy = 0.0f, om = 0.0f;

    // f_rk45_*(t, yt): set the trial state, compute_forces, ydot
    auto rhs = [&](const float (&yt)[NS], float (&out)[NS]) {
        ++nfev;
        float sn = 0.0f, cs = 1.0f;
        if (human) {
            px = yt[0]; py = yt[1];
            if (HEADED) {
                th = bound_angle(yt[2]);
                bvx = yt[3]; bvy = yt[4];
                const float sp = sqrtf(bvx * bvx + bvy * bvy);
                if (sp > vd) { bvx = bvx / sp * vd; bvy = bvy / sp * vd; }
                om = yt[5];
            } else {
                vx = yt[2]; vy = yt[3];
                const float sp = sqrtf(vx * vx + vy * vy);
                if (sp > vd) { vx = vx / sp * vd; vy = vy / sp * vd; }
            }
            // update_goals at the trial position (:66-70, strict <): rotate the list
            const float ddx = g0x - px, ddy = g0y - py;
            if (sqrtf(ddx * ddx + ddy * ddy) < radius) {
                int k = a.G;
                for (int g = 0; g < a.G; ++g) if (isnan(gi[2 * g])) { k = g; break; }
                const float r0 = gi[0], r1 = gi[1];
                for (int g = 0; g + 1 < k; ++g) { gi[2 * g] = gi[2 * g + 2]; gi[2 * g + 1] = gi[2 * g + 3]; }
                if (k > 0) { gi[2 * (k - 1)] = r0; gi[2 * (k - 1) + 1] = r1; }
                g0x = gi[0]; g0y = gi[1];
            }
            if (HEADED) {
                sincosf(th, &sn, &cs);
                vx = cs * bvx - sn * bvy; vy = sn * bvx + cs * bvy;
            }
        }
        __syncthreads();   // the previous evaluation's readers are done
        if (human) s_pv[lane] = make_float4(px, py, vx, vy);
        __syncthreads();
        if (human) {
            // social force
            float fsx = 0.0f, fsy = 0.0f;
            for (int j = 0; j < nent; ++j) {
                if (j == lane) continue;
                const float4 q = s_pv[j];
                const float rj = s_rs[j];
                float fx, fy;
                if (all_equal && j < lane) {   // mirrored: minus the force the lower index feels from me
                    pair_force(kind, P, q.x, q.y, q.z, q.w, rj, px, py, vx, vy, rs, fx, fy);
                    fsx -= fx; fsy -= fy;
                } else {
                    pair_force(kind, P, px, py, vx, vy, rs, q.x, q.y, q.z, q.w, rj, fx, fy);
                    fsx += fx; fsy += fy;
                }
            }
            // desired force, kept within one radius of the goal
            {
                const float ddx = g0x - px, ddy = g0y - py;
                const float dist = sqrtf(ddx * ddx + ddy * ddy);
                if (dist > radius) {
                    fdx = mass * (ddx / dist * vd - vx) / P[0];
                    fdy = mass * (ddy / dist * vd - vy) / P[0];
                }
            }
            // obstacle force: the LAST nearest point of every polygon (obstacle.py:53-66)
            float fox = 0.0f, foy = 0.0f;
            for (int o = 0; o < a.O; ++o) {
                float bx = 0.0f, by = 0.0f, bd = 10000.0f;
                for (int sg = 0; sg < a.Smax; ++sg) {
                    const float* q = ob + ((long)o * a.Smax + sg) * 4;
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
                const float rd = rs - dn;
                const float comp = fmaxf(0.0f, rd);
                const float fn = P[2] * expf(rd / P[4]) + P[10] * comp;
                const float ft = (kind == 1) ? (-P[6] * expf(rd / P[8]) - P[11] * comp) * dv : -P[11] * comp * dv;
                fox += fn * nx + ft * tx;
                foy += fn * ny + ft * ty;
            }
            if (a.O > 0 && kind != 1) { fox /= (float)a.O; foy /= (float)a.O; }
            if (!HEADED) {
                out[0] = vx; out[1] = vy;
                out[2] = (fdx + fox + fsx) / mass; out[3] = (fdy + foy + fsy) / mass;
            } else {
                const float tx = torque_new ? fdx + fox + fsx : fdx, ty = torque_new ? fdy + foy + fsy : fdy;
                const float tn = sqrtf(tx * tx + ty * ty);
                const float k_theta = inertia * P[19] * tn;
                const float k_omega = inertia * (1.0f + P[18]) * sqrtf(P[19] * tn / P[18]);
                const float torque = -k_theta * bound_angle(th - atan2f(ty, tx)) - k_omega * om;
                const float g0 = (fdx + fox + fsx) * cs + (fdy + foy + fsy) * sn;
                const float g1 = P[16] * ((fox + fsx) * -sn + (foy + fsy) * cs) - P[17] * bvy;
                out[0] = cs * bvx - sn * bvy; out[1] = sn * bvx + cs * bvy; out[2] = om;
                out[3] = g0 / mass; out[4] = g1 / mass; out[5] = torque / inertia;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NS; ++c) out[c] = 0.0f;
        }
    };

    const float invN = 1.0f / (float)(n * NS);
    auto rms = [&](const float (&v)[NS], const float (&sc)[NS]) {
        float s = 0.0f;
        if (human) {
#pragma unroll
            for (int c = 0; c < NS; ++c) { const float q = v[c] / sc[c]; s += q * q; }
        }
        return sqrtf(wave_sum(s) * invN);
    };
    // update_humans: one solve over dt.  complete_rk45_simulation: one solve over t_final, the solution at the times k * eval_dt
    // taken from the dense output of the step that contains them (solve_ivp(..., t_eval=...): a time equal to a step's end belongs
    // to that step, t_eval[0] = t0 to the first one)
    const float T = a.dense != nullptr ? a.t_final : a.dt;
    int ke = 0;
    rk45_solve<NS>(y, T, rhs, rms, [&](float t_old, float t_new, float h, const float (&y_old)[NS], const float (&K)[7][NS]) {
        if (a.dense == nullptr) return;
        while (ke < a.n_eval && (float)ke * a.eval_dt <= t_new) {
            const float x = ((float)ke * a.eval_dt - t_old) / h;
            if (human) {
                float* o = a.dense + (((long)w * a.n_eval + ke) * n + lane) * NS;
#pragma unroll
                for (int c = 0; c < NS; ++c) o[c] = rk45_dense<NS>(K, c, y_old[c], h, x);
            }
            ++ke;
        }
    });
    // ---- set_new_*_state_from_rk45_solution(y[:, -1]) (+ the linear velocity of headed agents), :377-384
    if (human) {
        px = y[0]; py = y[1];
        if (HEADED) {
            th = bound_angle(y[2]);
            bvx = y[3]; bvy = y[4];
            const float sp = sqrtf(bvx * bvx + bvy * bvy);
            if (sp > vd) { bvx = bvx / sp * vd; bvy = bvy / sp * vd; }
            om = y[5];
            float sn, cs;
            sincosf(th, &sn, &cs);
            vx = cs * bvx - sn * bvy; vy = sn * bvx + cs * bvy;
            srow[2 * fs] = th; srow[5 * fs] = bvx; srow[6 * fs] = bvy; srow[7 * fs] = om;
        } else {
            vx = y[2]; vy = y[3];
            const float sp = sqrtf(vx * vx + vy * vy);
            if (sp > vd) { vx = vx / sp * vd; vy = vy / sp * vd; }
        }
        srow[0] = px; srow[fs] = py; srow[3 * fs] = vx; srow[4 * fs] = vy;
        srow[10 * fs] = g0x; srow[11 * fs] = g0y;
        mem[0] = fdx; mem[1] = fdy;
    }
    if (a.nfev != nullptr && lane == 0) a.nfev[w] = nfev;
}

// ---- the ROBOT under RK45 (motion_model_manager.py:631-640, 661-687): solve_ivp around f_rk45_robot_* -- the trial state written
// into the robot (:88-103), compute_robot_forces (:591-613: the single-agent force functions of robot_model.h, humans fixed for the
// whole solve), ydot = [v, F / m] or [R bv, omega, F_body / m, torque / I].  One wavefront per world: lane j hands in human j's
// term of the social force, every lane carries the robot's state and takes the same decisions.
struct KArgsRobotRk {
    int W, n, rows, O, Smax, type, write_row, obstacles_shared;
    float dt, robot_margin;
    float P[20];
    float* S; long as, fs;
    const float* hmargin;
    float* robot;
    float* memory;
    const float* obstacles;
    int* nfev;
};

template <bool HEADED>
__global__ __launch_bounds__(64) void k_robot_rk45(const KArgsRobotRk a)
{
    constexpr int NS = HEADED ? 6 : 4;
    __shared__ float2 s_term[64];
    const int w = blockIdx.x, lane = threadIdx.x;
    float* rb = a.robot + (long)w * 13;
    rmodel::RState r;
    r.px = rb[0]; r.py = rb[1]; r.yaw = rb[2]; r.vx = rb[3]; r.vy = rb[4]; r.bvx = rb[5]; r.bvy = rb[6]; r.om = rb[7];
    r.radius = rb[8]; r.mass = rb[9]; r.gx = rb[10]; r.gy = rb[11]; r.vd = rb[12];
    float* mem = a.memory + (long)w * 2;
    r.fdx = mem[0]; r.fdy = mem[1];
    const int soc = a.type % 3;
    const float rme = r.radius + a.robot_margin;
    // the humans stand still during the robot's solve: one human per lane (rows <= 64)
    float hx = 0.0f, hy = 0.0f, hvx = 0.0f, hvy = 0.0f, hrij = 0.0f;
    if (lane < a.n) {
        const float* s = a.S + ((long)w * a.rows + lane) * a.as;
        hx = s[0]; hy = s[a.fs]; hvx = s[3 * a.fs]; hvy = s[4 * a.fs];
        hrij = rme + s[8 * a.fs] + a.hmargin[(long)w * a.rows + lane];
    }
    const float* ob = a.O > 0 ? a.obstacles + (a.obstacles_shared ? 0 : (long)w * a.O * a.Smax * 4) : nullptr;
    int nfev = 0;
    float y[NS];
    if (HEADED) { y[0] = r.px; y[1] = r.py; y[2] = r.yaw; y[3] = r.bvx; y[4] = r.bvy; y[5] = r.om; }
    else { y[0] = r.px; y[1] = r.py; y[2] = r.vx; y[3] = r.vy; }

    auto set_state = [&](const float (&yt)[NS]) {   // set_new_*_state_from_rk45_solution(robot, y), :88-103
        r.px = yt[0]; r.py = yt[1];
        if (HEADED) {
            r.yaw = rmodel::bound_angle(yt[2]);
            r.bvx = yt[3]; r.bvy = yt[4];
            const float sp = sqrtf(r.bvx * r.bvx + r.bvy * r.bvy);
            if (sp > r.vd) { r.bvx = r.bvx / sp * r.vd; r.bvy = r.bvy / sp * r.vd; }
            r.om = yt[5];
        } else {
            r.vx = yt[2]; r.vy = yt[3];
            const float sp = sqrtf(r.vx * r.vx + r.vy * r.vy);
            if (sp > r.vd) { r.vx = r.vx / sp * r.vd; r.vy = r.vy / sp * r.vd; }
        }
    };
    auto rhs = [&](const float (&yt)[NS], float (&out)[NS]) {
        ++nfev;
        set_state(yt);
        float sn, cs;
        rmodel::refresh_velocity(r, HEADED, sn, cs);
        float tx = 0.0f, ty = 0.0f;
        if (lane < a.n) rmodel::pair_term(soc, a.P, r.px, r.py, r.vx, r.vy, hx, hy, hvx, hvy, hrij, tx, ty);
        asm volatile("" ::: "memory");
        s_term[lane] = make_float2(tx, ty);
        asm volatile("" ::: "memory");
        float fsx = 0.0f, fsy = 0.0f;
        for (int j = 0; j < a.n; ++j) { const float2 t = s_term[j]; fsx += t.x; fsy += t.y; }
        float fox, foy;
        rmodel::obstacle_force(ob, a.O, a.Smax, soc, a.P, r.px, r.py, r.vx, r.vy, rme, fox, foy);
        rmodel::derivative(r, a.type, a.P, fsx, fsy, fox, foy, sn, cs, out);
    };
    const float invN = 1.0f / (float)NS;
    auto rms = [&](const float (&v)[NS], const float (&sc)[NS]) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < NS; ++c) { const float q = v[c] / sc[c]; s += q * q; }
        return sqrtf(s * invN);
    };
    rk45_solve<NS>(y, a.dt, rhs, rms, [](float, float, float, const float (&)[NS], const float (&)[7][NS]) {});
    set_state(y);
    if (HEADED) {   // headed_agent_update_linear_velocity(robot) after the solve (:640)
        float sn, cs;
        rmodel::refresh_velocity(r, true, sn, cs);
    }
    if (lane != 0) return;
    rb[0] = r.px; rb[1] = r.py; rb[2] = r.yaw; rb[3] = r.vx; rb[4] = r.vy; rb[5] = r.bvx; rb[6] = r.bvy; rb[7] = r.om;
    mem[0] = r.fdx; mem[1] = r.fdy;
    if (a.write_row) {
        float* s = a.S + ((long)w * a.rows + a.n) * a.as;
        const long fs = a.fs;
        s[0] = r.px; s[fs] = r.py; s[2 * fs] = r.yaw; s[3 * fs] = r.vx; s[4 * fs] = r.vy; s[5 * fs] = r.bvx; s[6 * fs] = r.bvy; s[7 * fs] = r.om;
    }
    if (a.nfev != nullptr) a.nfev[w] = nfev;
}

int rk45_launch(const cs_worlds* w, float dt, float t_final, float eval_dt, int n_eval, float* d_memory, float* d_dense, int32_t* d_nfev, void* stream)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (w->type < 0 || w->type > 8) return fail(CS_ERR_TYPE, "Type " + std::to_string(w->type) + " does not exist for this implementation");
    if (w->W <= 0 || w->n <= 0 || w->G <= 0) return fail(CS_ERR_ARG, "W, n, G must be positive");
    if (!w->d_state || !w->d_goals || !w->d_params || !w->d_safety || !d_memory) return fail(CS_ERR_ARG, "null device buffer");
    if (w->O < 0 || (w->O > 0 && (!w->d_obstacles || w->Smax <= 0))) return fail(CS_ERR_ARG, "bad obstacle description");
    if (w->layout != CS_LAYOUT_AOS && w->layout != CS_LAYOUT_SOA) return fail(CS_ERR_ARG, "bad layout");
    if (!(dt > 0.0f)) return fail(CS_ERR_ARG, "dt must be positive");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    if (rows > 64) return fail(CS_ERR_ARG, "the RK45 step supports up to 64 rows per world");
    KArgsRk a;
    std::memset(&a, 0, sizeof(a));
    a.W = w->W; a.n = w->n; a.rows = rows; a.G = w->G; a.O = w->O; a.Smax = w->Smax; a.type = w->type; a.flags = w->flags;
    a.dt = dt; a.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; } else { a.as = 1; a.fs = (long)w->W * rows; }
    a.goals = w->d_goals; a.params = w->d_params; a.safety = w->d_safety; a.obstacles = w->d_obstacles;
    a.memory = d_memory; a.nfev = d_nfev;
    a.t_final = t_final; a.eval_dt = eval_dt; a.n_eval = n_eval; a.dense = d_dense;
    if (w->type >= CS_HSFM_FARINA) hipLaunchKernelGGL(k_rk45_step<true>, dim3(w->W), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_rk45_step<false>, dim3(w->W), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    // post_update of update_humans: the parallel-traffic respawn rule on the agents (motion_model_manager.py:407-422, non-parallel path)
    if (d_dense == nullptr && (w->flags & CS_RESPAWN))
        csimpl::big_respawn_launch(w->d_state, a.as, a.fs, w->W, w->n, rows, w->d_goals, w->G, w->d_safety, 2, w->respawn_bound_x, w->respawn_bound_y,
                                   w->d_world_flags, (hipStream_t)stream);
    return CS_OK;
}

} // namespace

extern "C" {

int cs_update_humans_rk45(const cs_worlds* w, float dt, float* d_memory, int32_t* d_nfev, void* stream)
{
    return rk45_launch(w, dt, 0.0f, 0.0f, 0, d_memory, nullptr, d_nfev, stream);
}

int cs_complete_rk45_simulation(const cs_worlds* w, float dt, float final_time, float* d_memory, float* d_human_states, int n_eval,
                                int32_t* d_nfev, void* stream)
{
    if (!d_human_states || n_eval <= 0) return fail(CS_ERR_ARG, "complete_rk45_simulation needs the output array and its number of evaluation times");
    if (!(final_time > 0.0f)) return fail(CS_ERR_ARG, "final_time must be positive");
    return rk45_launch(w, dt, final_time, dt, n_eval, d_memory, d_human_states, d_nfev, stream);
}

int cs_robot_model_rk45(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin, const float* d_human_margin,
                        float* d_robot_memory, float dt, int32_t* d_nfev, void* stream)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (w->W <= 0 || w->n <= 0 || !w->d_state || !w->d_robot) return fail(CS_ERR_ARG, "bad cs_worlds (a robot needs d_robot)");
    if (w->layout != CS_LAYOUT_AOS && w->layout != CS_LAYOUT_SOA) return fail(CS_ERR_ARG, "bad layout");
    const float* hm = d_human_margin ? d_human_margin : w->d_safety;
    if (!hm || !robot_params || !d_robot_memory) return fail(CS_ERR_ARG, "null argument");
    if (robot_type < 0 || robot_type > 8) return fail(CS_ERR_TYPE, "Runge-Kutta integration of the robot takes one of the nine SFM / HSFM models");
    if (w->O < 0 || (w->O > 0 && (!w->d_obstacles || w->Smax <= 0))) return fail(CS_ERR_ARG, "bad obstacle description");
    if (!(dt > 0.0f)) return fail(CS_ERR_ARG, "dt must be positive");
    if (w->n > 64) return fail(CS_ERR_ARG, "the robot's RK45 step supports up to 64 humans per world");
    KArgsRobotRk a;
    std::memset(&a, 0, sizeof(a));
    a.W = w->W; a.n = w->n; a.rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    // as robot_step_impl (robot_model.hip): an ORCA crowd's simulator sees the moved robot only after its own doStep (motion_model_manager.py:389)
    a.write_row = ((w->flags & CS_ROBOT_ROW) && w->type != CS_ORCA) ? 1 : 0;
    a.O = w->O; a.Smax = w->Smax; a.type = robot_type; a.obstacles_shared = (w->flags & CS_OBSTACLES_SHARED) ? 1 : 0;
    a.dt = dt; a.robot_margin = robot_margin;
    std::memcpy(a.P, robot_params, sizeof(a.P));
    a.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; } else { a.as = 1; a.fs = (long)w->W * a.rows; }
    a.hmargin = hm; a.robot = w->d_robot; a.memory = d_robot_memory; a.obstacles = w->d_obstacles; a.nfev = d_nfev;
    if (robot_type >= CS_HSFM_FARINA) hipLaunchKernelGGL(k_robot_rk45<true>, dim3(w->W), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_robot_rk45<false>, dim3(w->W), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

} // extern "C"
