// robot_model.hip -- the robot driven by a human motion model (imitation learning), W worlds at once.
//
// Restates MotionModelManager.update_robot(t, dt) with a robot model set by set_robot_motion_model
//   /root/reference/social_gym/src/motion_model_manager.py:552-653   (compute_robot_forces :591-613)
// for the nine SFM / HSFM titles, with the SINGLE-AGENT force functions the reference uses for the robot
//   /root/reference/social_gym/src/forces.py:9-16    desired force (kept from the previous substep within one radius of
//                                                     the goal: the `else` branch does not touch agent.desired_force)
//   forces.py:27-53     obstacle force, Helbing (mean over the walls) / Guo (plain sum)
//   forces.py:153-218   social force Helbing / Guo / Moussaid over the humans in index order (consider_robot = False)
//   forces.py:279-290   torque Farina (desired force) / "new" (total force)
// and the Euler updates motion_model_manager.py:72-86 (position first, then velocity, speed clamp; headed: yaw, body
// velocity, angular velocity, linear velocity from the NEW yaw).  These are not the parallel functions of the crowd step.
// The robot's ORCA model (a second RVO simulator, :580-589, :641-653) lives in orca.hip (orca_robot_launch).
//
// One wavefront per world: lane j sums the social force of humans j, j + 64, ...; a butterfly reduction; lane 0 does the
// walls (closest point of every polygon, obstacle.py:53-66), the torque and the Euler update and writes the robot row
// (w->d_robot, and the last state row when the robot is visible).  Not a hot path: W x (n + O*Smax) pair terms per substep.
// cs_actual_collision_reward restates check_actual_collisions_and_goal + compute_reward_and_infos
//   /root/reference/social_gym/social_nav_gym.py:107-118, social_nav_sim.py:986-1029.
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

struct RArgs {
    int W, n, rows, O, Smax, type, robot_row, write_row, obstacles_shared;
    float dt, robot_margin;
    float P[20];             // agent.py:269 slots: 0 relaxation_time 1 Ai 2 Aw 3 Bi 4 Bw 5 Ci 6 Cw 7 Di 8 Dw 9 Ei 10 k1 11 k2
                             // 12 lambda 13 gamma 14 ns 15 ns1 16 ko 17 kd 18 alpha 19 k_lambda
    float* S; long as, fs;
    const float* hmargin;    // [W][rows]
    float* robot;            // [W][13]
    float* memory;           // [W][2] desired force of the previous substep
    const float* obstacles;
    int just_velocities;     // update_robot(just_velocities=True): velocities integrate, position and yaw stay (:72-85)
    const float4* snap;      // imitation block: [nsub][W][n] (x, y, vx, vy) of the humans at the start of every substep (else nullptr)
    int nsub;
};

using rmodel::PI_F;
using rmodel::TWO_PI_F;
using rmodel::bound_angle;

// sum over the 64 lanes, left in every lane: four DPP rotate-and-add steps inside each 16-lane row, then the four row sums through
// scalar registers (v_readlane) -- a few dozen cycles instead of six dependent LDS-crossbar shuffles (~120 cycles each)
template <int N> __device__ __forceinline__ float row_ror_add(float x)
{
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v = row_ror_add<8>(v); v = row_ror_add<4>(v); v = row_ror_add<2>(v); v = row_ror_add<1>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

__global__ __launch_bounds__(256) void k_robot_model_step(const RArgs a)
{
    __shared__ float2 s_term[4][64];          // the humans' terms of the robot's social force, summed in index order (n <= 64)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * (blockDim.x >> 6) + wv;
    if (w >= a.W) return;
    float* rb = a.robot + (long)w * 13;
    rmodel::RState r;
    r.px = rb[0]; r.py = rb[1]; r.yaw = rb[2]; r.vx = rb[3]; r.vy = rb[4]; r.bvx = rb[5]; r.bvy = rb[6]; r.om = rb[7];
    r.radius = rb[8]; r.mass = rb[9]; r.gx = rb[10]; r.gy = rb[11]; r.vd = rb[12];
    const bool headed = a.type >= CS_HSFM_FARINA;
    const int soc = a.type % 3;                 // 0 Helbing, 1 Guo, 2 Moussaid
    const float* P = a.P;
    const float rme = r.radius + a.robot_margin;
    float* mem = a.memory + (long)w * 2;
    r.fdx = mem[0]; r.fdy = mem[1];
    // one substep (cs_robot_model_step), or all substeps of an imitation block against the crowd's snapshots: every lane carries
    // the robot's state and integrates it identically (same inputs, same operations), lane 0 writes it back at the end
    // imitation block: the humans' records of the NEXT substep are requested while this one is integrated (a substep is a few hundred
    // instructions, a dependent global load about as long)
    const bool pre = a.snap != nullptr && a.n <= 64;
    float4 qpre = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (pre && lane < a.n) qpre = a.snap[(long)w * a.n + lane];
    // radius + margins of "my" human never change during the launch (worlds of up to 64 humans: one per lane): loaded once, not in every
    // substep behind a dependent global load
    float rij_lane = 0.0f;
    if (a.n <= 64 && lane < a.n) rij_lane = rme + a.S[((long)w * a.rows + lane) * a.as + 8 * a.fs] + a.hmargin[(long)w * a.rows + lane];
    for (int sub = 0; sub < a.nsub; ++sub) {
    float4 qnext = qpre;
    if (pre && lane < a.n && sub + 1 < a.nsub) qnext = a.snap[((long)(sub + 1) * a.W + w) * a.n + lane];
    float sn, cs;
    rmodel::refresh_velocity(r, headed, sn, cs);                         // headed_agent_update_linear_velocity (:143-145)

    // ---- social force: one term per human (rmodel::pair_term), summed in index order -- the order the fused crowd launch
    //      (k_sfm_step<..., LEAN = 4>) uses too; worlds of more than 64 humans: strided partial sums and a butterfly
    float fsx = 0.0f, fsy = 0.0f;
    for (int j = lane; j < a.n; j += 64) {
        const float* s = a.S + ((long)w * a.rows + j) * a.as;
        float hx, hy, hvx, hvy;
        if (pre) { hx = qpre.x; hy = qpre.y; hvx = qpre.z; hvy = qpre.w; }
        else if (a.snap != nullptr) { const float4 q = a.snap[((long)sub * a.W + w) * a.n + j]; hx = q.x; hy = q.y; hvx = q.z; hvy = q.w; }
        else { hx = s[0]; hy = s[a.fs]; hvx = s[3 * a.fs]; hvy = s[4 * a.fs]; }
        const float rij = a.n <= 64 ? rij_lane : rme + s[8 * a.fs] + a.hmargin[(long)w * a.rows + j];
        float tx, ty;
        rmodel::pair_term(soc, P, r.px, r.py, r.vx, r.vy, hx, hy, hvx, hvy, rij, tx, ty);
        fsx += tx; fsy += ty;
    }
    if (a.n <= 64) {
        s_term[wv][lane] = make_float2(fsx, fsy);        // (lanes >= n hold zeros and are not read)
        asm volatile("" ::: "memory");                    // one wavefront: its LDS operations execute in order
        float sx = 0.0f, sy = 0.0f;
#pragma unroll 8
        for (int j = 0; j < a.n; ++j) { const float2 t = s_term[wv][j]; sx += t.x; sy += t.y; }   // (eight terms requested per trip; summed in index order)
        asm volatile("" ::: "memory");
        fsx = sx; fsy = sy;
    } else {
        fsx = wave_sum(fsx);
        fsy = wave_sum(fsy);
    }

    // ---- obstacle force: one closest point per polygon (the LAST nearest segment point, `<=`)
    float fox = 0.0f, foy = 0.0f;
    if (a.O > 0) {
        const float px = r.px, py = r.py, vx = r.vx, vy = r.vy;
        const float* ob = a.obstacles + (a.obstacles_shared ? 0 : (long)w * a.O * a.Smax * 4);
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
            const float rd = rme - dn;
            const float comp = fmaxf(0.0f, rd);
            const float fn = P[2] * expf(rd / P[4]) + P[10] * comp;
            float ft;
            if (soc == 1) ft = (-P[6] * expf(rd / P[8]) - P[11] * comp) * dv;
            else ft = -P[11] * comp * dv;
            fox += fn * nx + ft * tx;
            foy += fn * ny + ft * ty;
        }
        if (soc != 1) { fox /= (float)a.O; foy /= (float)a.O; }   // Guo's single-agent obstacle force is not averaged
    }
    rmodel::integrate(r, a.type, P, fsx, fsy, fox, foy, sn, cs, a.dt, a.just_velocities);
    qpre = qnext;
    }   // substeps
    if (lane != 0) return;
    rb[0] = r.px; rb[1] = r.py; rb[2] = r.yaw; rb[3] = r.vx; rb[4] = r.vy; rb[5] = r.bvx; rb[6] = r.bvy; rb[7] = r.om;
    mem[0] = r.fdx; mem[1] = r.fdy;
    if (a.write_row) {
        float* s = a.S + ((long)w * a.rows + a.n) * a.as;
        const long fs = a.fs;
        s[0] = r.px; s[fs] = r.py; s[2 * fs] = r.yaw; s[3 * fs] = r.vx; s[4 * fs] = r.vy; s[5 * fs] = r.bvx; s[6 * fs] = r.bvy; s[7 * fs] = r.om;
    }
}

// check_actual_collisions_and_goal (social_nav_gym.py:107-118) + compute_reward_and_infos (social_nav_sim.py:986-1029):
// distances at the CURRENT state (no swept test, `<=` for the collision), one lane per world.
__global__ void k_actual_collision_reward(int W, int n, int rows, const float* S, long as, long fs, const float* robot, float T,
                                          const float* gtime, float time_limit, float success_reward, float collision_penalty,
                                          float discomfort_dist, float discomfort_factor, float* out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    const float* rb = robot + (long)w * 13;
    const float rpx = rb[0], rpy = rb[1], rr = rb[8], rgx = rb[10], rgy = rb[11];
    float dmin = 10000.0f;
    for (int i = 0; i < n; ++i) {
        const float* s = S + ((long)w * rows + i) * as;
        const float dx = s[0] - rpx, dy = s[fs] - rpy;
        const float d = sqrtf(dx * dx + dy * dy) - s[8 * fs] - rr;
        dmin = d < dmin ? d : dmin;
    }
    const int collision = dmin <= 0.0f;
    const float ex = rpx - rgx, ey = rpy - rgy;
    const int reaching = sqrtf(ex * ex + ey * ey) < rr;
    float reward = 0.0f; int term = 0, trunc = 0, info = 0;
    if (gtime[w] >= time_limit - 1.0f) { trunc = 1; info = 4; }
    else if (collision) { reward = collision_penalty; term = 1; info = 3; }
    else if (reaching) { reward = success_reward; term = 1; info = 2; }
    else if (dmin < discomfort_dist) { reward = (dmin - discomfort_dist) * discomfort_factor * T; info = 1; }
    float* o = out + (long)w * 7;
    o[0] = (float)collision; o[1] = dmin; o[2] = (float)reaching; o[3] = reward;
    o[4] = (float)term; o[5] = (float)trunc; o[6] = (float)info;
}

// Episode bookkeeping of a vectorised Gym loop, one lane per world: what SocialNavGym does on the host between two steps
// (social_nav_gym.py:244 global_time += time_step, time_step_factor times -- the float32 sums are tabulated in `clock`;
// :135-197 the next reset of a finished world draws the next unused seed) plus the typed copies of the reward kernel's
// output row, so that a step of BatchedSocialNavGym.step_device needs no element-wise torch op.
__global__ void k_gym_bookkeeping(int W, const float* out7, int* counter, unsigned* seeds, int* mask, float* gtime, const float* clock,
                                  int clock_len, int auto_reset, float* reward, unsigned char* terminated, unsigned char* truncated,
                                  int* info, unsigned stride)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    const float* o = out7 + (long)w * 7;
    const bool term = o[4] > 0.0f, trunc = o[5] > 0.0f;
    reward[w] = o[3]; terminated[w] = term ? 1 : 0; truncated[w] = trunc ? 1 : 0; info[w] = (int)o[6];
    int c = counter[w] + 1;
    if (auto_reset) {
        const bool done = term || trunc;
        mask[w] = done ? 1 : 0;
        if (done) { seeds[w] += stride; c = 0; }   // every world walks its own arithmetic sequence of seeds (stride = worlds of the whole job)
    }
    c = c < clock_len - 1 ? c : clock_len - 1;
    counter[w] = c;
    gtime[w] = clock[c];
}

// The same for Gymnasium's NEXT_STEP autoreset mode: a world whose episode ended in the previous step (prev != 0) is being reset during
// this step -- its results are the reset step's (reward 0, not terminated, not truncated, Nothing), its clock restarts -- and a world
// that ends now is only flagged (mask) and takes the next seed of its sequence; its replacement is generated beside the next step.
__global__ void k_gym_bookkeeping_next_step(int W, const float* out7, int* counter, unsigned* seeds, int* mask, const int* prev, float* gtime,
                                            const float* clock, int clock_len, float* reward, unsigned char* terminated,
                                            unsigned char* truncated, int* info, unsigned stride)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    if (prev[w]) {
        reward[w] = 0.0f; terminated[w] = 0; truncated[w] = 0; info[w] = 0;
        mask[w] = 0; counter[w] = 0; gtime[w] = clock[0];
        return;
    }
    const float* o = out7 + (long)w * 7;
    const bool term = o[4] > 0.0f, trunc = o[5] > 0.0f;
    reward[w] = o[3]; terminated[w] = term ? 1 : 0; truncated[w] = trunc ? 1 : 0; info[w] = (int)o[6];
    const bool done = term || trunc;
    mask[w] = done ? 1 : 0;
    if (done) seeds[w] += stride;
    int c = counter[w] + 1;
    c = c < clock_len - 1 ? c : clock_len - 1;
    counter[w] = c;
    gtime[w] = clock[c];
}

} // namespace

extern "C" {

int cs_gym_bookkeeping_next_step(int W, const float* d_out, int32_t* d_counter, uint32_t* d_seeds, int32_t* d_mask, const int32_t* d_prev_mask,
                                 float* d_global_time, const float* d_clock, int clock_len, float* d_reward, uint8_t* d_terminated,
                                 uint8_t* d_truncated, int32_t* d_info, uint32_t seed_stride, void* stream)
{
    if (W <= 0 || clock_len <= 0) return fail(CS_ERR_ARG, "W and clock_len must be positive");
    if (!d_out || !d_counter || !d_seeds || !d_mask || !d_prev_mask || !d_global_time || !d_clock || !d_reward || !d_terminated || !d_truncated || !d_info)
        return fail(CS_ERR_ARG, "null argument");
    const int block = 64;
    hipLaunchKernelGGL(k_gym_bookkeeping_next_step, dim3((W + block - 1) / block), dim3(block), 0, (hipStream_t)stream, W, d_out, d_counter,
                       d_seeds, d_mask, d_prev_mask, d_global_time, d_clock, clock_len, d_reward, d_terminated, d_truncated, d_info,
                       seed_stride ? seed_stride : (unsigned)W);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

static int robot_step_impl(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin,
                           const float* d_human_margin, float* d_robot_memory, float dt, int just_velocities, void* stream);

int cs_robot_model_step(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin,
                        const float* d_human_margin, float* d_robot_memory, float dt, void* stream)
{
    return robot_step_impl(w, robot_type, robot_params, robot_margin, d_human_margin, d_robot_memory, dt, 0, stream);
}

int cs_robot_model_velocities(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin,
                              const float* d_human_margin, float* d_robot_memory, float dt, void* stream)
{
    return robot_step_impl(w, robot_type, robot_params, robot_margin, d_human_margin, d_robot_memory, dt, 1, stream);
}

static int robot_step_impl(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin,
                           const float* d_human_margin, float* d_robot_memory, float dt, int just_velocities, void* stream)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (w->W <= 0 || w->n <= 0 || !w->d_state || !w->d_robot) return fail(CS_ERR_ARG, "bad cs_worlds (a robot needs d_robot)");
    if (w->layout != CS_LAYOUT_AOS && w->layout != CS_LAYOUT_SOA) return fail(CS_ERR_ARG, "bad layout");
    const float* hm = d_human_margin ? d_human_margin : w->d_safety;
    if (!hm) return fail(CS_ERR_ARG, "no human margins");
    if (robot_type == CS_ORCA) return csimpl::orca_robot_launch(w, robot_margin, hm, dt, (hipStream_t)stream, just_velocities);
    if (robot_type < 0 || robot_type > 8)
        return fail(CS_ERR_TYPE, "The robot motion model '" + std::to_string(robot_type) + "' does not exist");
    if (!robot_params || !d_robot_memory) return fail(CS_ERR_ARG, "null argument");
    if (w->O < 0 || (w->O > 0 && (!w->d_obstacles || w->Smax <= 0))) return fail(CS_ERR_ARG, "bad obstacle description");
    RArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = w->W; a.n = w->n; a.robot_row = (w->flags & CS_ROBOT_ROW) ? 1 : 0; a.rows = w->n + a.robot_row;
    a.write_row = a.robot_row && w->type != CS_ORCA;   // an ORCA crowd's simulator sees the moved robot only after its doStep (:389)
    a.O = w->O; a.Smax = w->Smax; a.type = robot_type; a.obstacles_shared = (w->flags & CS_OBSTACLES_SHARED) ? 1 : 0;
    a.dt = dt; a.robot_margin = robot_margin;
    std::memcpy(a.P, robot_params, sizeof(a.P));
    a.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; } else { a.as = 1; a.fs = (long)w->W * a.rows; }
    a.hmargin = hm; a.robot = w->d_robot; a.memory = d_robot_memory; a.obstacles = w->d_obstacles;
    a.snap = nullptr; a.nsub = 1; a.just_velocities = just_velocities;
    const int wpb = 4;
    hipLaunchKernelGGL(k_robot_model_step, dim3((w->W + wpb - 1) / wpb), dim3(64 * wpb), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_actual_collision_reward(const cs_worlds* w, float T, const float* d_global_time, const float* reward_cfg, float* d_out,
                               void* stream)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (w->W <= 0 || w->n <= 0 || !w->d_state) return fail(CS_ERR_ARG, "bad cs_worlds");
    if (w->layout != CS_LAYOUT_AOS && w->layout != CS_LAYOUT_SOA) return fail(CS_ERR_ARG, "bad layout");
    if (!d_global_time || !reward_cfg || !d_out || !w->d_robot) return fail(CS_ERR_ARG, "null argument");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    long as, fs;
    if (w->layout == CS_LAYOUT_AOS) { as = 13; fs = 1; } else { as = 1; fs = (long)w->W * rows; }
    const int block = 64;
    hipLaunchKernelGGL(k_actual_collision_reward, dim3((w->W + block - 1) / block), dim3(block), 0, (hipStream_t)stream, w->W, w->n,
                       rows, w->d_state, as, fs, w->d_robot, T, d_global_time, reward_cfg[0], reward_cfg[1], reward_cfg[2],
                       reward_cfg[3], reward_cfg[4], d_out);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_gym_bookkeeping(int W, const float* d_out, int32_t* d_counter, uint32_t* d_seeds, int32_t* d_mask, float* d_global_time,
                       const float* d_clock, int clock_len, int auto_reset, float* d_reward, uint8_t* d_terminated,
                       uint8_t* d_truncated, int32_t* d_info, uint32_t seed_stride, void* stream)
{
    if (W <= 0 || clock_len <= 0) return fail(CS_ERR_ARG, "W and clock_len must be positive");
    if (!d_out || !d_counter || !d_global_time || !d_clock || !d_reward || !d_terminated || !d_truncated || !d_info)
        return fail(CS_ERR_ARG, "null argument");
    if (auto_reset && (!d_seeds || !d_mask)) return fail(CS_ERR_ARG, "auto_reset needs the seeds and the mask");
    const int block = 64;
    hipLaunchKernelGGL(k_gym_bookkeeping, dim3((W + block - 1) / block), dim3(block), 0, (hipStream_t)stream, W, d_out, d_counter,
                       d_seeds, d_mask, d_global_time, d_clock, clock_len, auto_reset, d_reward, d_terminated, d_truncated, d_info,
                       seed_stride ? seed_stride : (unsigned)W);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

} // extern "C"

int csimpl::robot_block_launch(const cs_worlds* w, int robot_type, const float* robot_params, float robot_margin, const float* d_human_margin,
                               float* d_robot_memory, float dt, int n_substeps, const float4* d_snap, hipStream_t stream)
{
    const float* hm = d_human_margin ? d_human_margin : w->d_safety;
    if (!hm || !robot_params || !d_robot_memory || !d_snap) return fail(CS_ERR_ARG, "null argument");
    if (w->O < 0 || (w->O > 0 && (!w->d_obstacles || w->Smax <= 0))) return fail(CS_ERR_ARG, "bad obstacle description");
    RArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = w->W; a.n = w->n; a.robot_row = 0; a.rows = w->n; a.write_row = 0;
    a.O = w->O; a.Smax = w->Smax; a.type = robot_type; a.obstacles_shared = (w->flags & CS_OBSTACLES_SHARED) ? 1 : 0;
    a.dt = dt; a.robot_margin = robot_margin;
    std::memcpy(a.P, robot_params, sizeof(a.P));
    a.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; } else { a.as = 1; a.fs = (long)w->W * a.rows; }
    a.hmargin = hm; a.robot = w->d_robot; a.memory = d_robot_memory; a.obstacles = w->d_obstacles;
    a.snap = d_snap; a.nsub = n_substeps;
    const int wpb = 4;
    hipLaunchKernelGGL(k_robot_model_step, dim3((w->W + wpb - 1) / wpb), dim3(64 * wpb), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}
