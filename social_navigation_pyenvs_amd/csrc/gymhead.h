// gymhead.h -- the head of a vectorised Gym step: reward / termination of the state BEFORE the substeps and the episode bookkeeping
// behind it, shared by k_collision_reward_wave (crowdstep.hip: cs_collision_reward_gym, a launch of its own) and by the step kernel
// (sfmstep_kernel.h: cs_gym_step runs it in the prologue of the fused substeps -- one launch and one graph node less per Gym step).
//   collision_detection_and_reaching_goal   /root/reference/social_gym/social_nav_sim.py:949-984  (swept test: utils.py:22-36)
//   compute_reward_and_infos                social_nav_sim.py:986-1029
//   SocialNavGym.step                       social_nav_gym.py:227-250  (the reward belongs to the state before the substeps, :229-233)
// The arithmetic is written without contraction: the two call sites must give the same bits whatever surrounds them.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "worldcopy.h"

namespace cstep {

// cs_gym_step_staged: the take-over of the pre-staged episodes (cs_consume_staged_worlds, generate.hip) folded into the step launch.
// The head (prologue) decides which worlds take over their staged episode at the end of THIS launch, checks the slot's tag and leaves a
// code in `pending[w]`; the epilogue of the wavefront that stepped the world copies the slot over it instead of writing the stepped rows.
//   pending[w]  between launches: 1 = the world's episode is over but its slot was not staged yet (the take-over is DEFERRED: the world
//               is between episodes -- reward 0, no flags, as a NEXT_STEP world after its terminal step -- until a later launch finds
//               the slot), 0 otherwise.  Inside a launch: 0 = keep stepping, 1 = deferred, 2 + slot = copy that slot, -(2 + slot) = that slot's
//               episode could not be generated (status != 0): nothing is copied, the stepped rows are written, the slot's turn is used up.
// Ordering (no acquire / release pair, as in k_consume_staged): the tag is written LAST by the refill, behind a release of the slot's
// words; it is read here with a device-scope relaxed load, the words ~30 us later with device-scope relaxed loads under a control
// dependency on it (a slot whose tag is this world's next seed is not rewritten before the world's epoch moves); the epoch is stored
// after every load of the slot has returned (s_waitcnt vmcnt(0) behind the copy: program order of one wavefront) -- from then on the
// refill may overwrite the slot.
struct GymFold {
    int on, depth, W;
    const unsigned* staged_seed; const int* staged_status; unsigned* epoch; int* failed; int* pending;
    csimpl::CopyArgs copy;   // staging -> live (with the Gym's observation rows)
};

// episode bookkeeping of a vectorised Gym step (cs_gym_bookkeeping / _next_step, robot_model.hip), done by the lane that wrote the
// world's reward row (mode 0: none, 1: same-step rules, 2: NEXT_STEP rules)
struct GymBook {
    int mode, clock_len, auto_reset;
    unsigned stride;   // what a finished world's seed moves on by (cs_gym_book.seed_stride; the worlds of the whole job)
    int* counter; unsigned* seeds; int* mask; const int* prev; float* gtime; const float* clock;
    float* reward; unsigned char* terminated; unsigned char* truncated; int* info;
};

struct GymHead {
    int unicycle;            // the action rows are ActionRot (v, r) (CS_ROBOT_UNICYCLE): the robot's velocity over the swept test is v (cos, sin)(r + yaw)
    float* out;              // [W][7] collision, dmin, reaching_goal, reward, terminated, truncated, info; nullptr: no head in this launch
    const float* gtime;      // [W] global time read for the time limit (== bk.gtime when the bookkeeping runs)
    float T, time_limit, success_reward, collision_penalty, discomfort_dist, discomfort_factor;
    GymBook bk;
    GymFold fold;
};

// The robot's velocity over the swept test's horizon for its action (social_nav_sim.py:968-973): an ActionXY is the velocity; an ActionRot (v, r)
// gives v (cos, sin)(r + yaw) -- the reference reads a `robot.theta` no agent has there, the yaw is what it means; golden G17 holds
// the reference's values with that attribute provided.  Hardware sin / cos in revolutions (stepcommon.h sincos_fast: 2.7e-7), both call sites alike.
__device__ __forceinline__ void gym_action_velocity(int unicycle, float yaw, float& ax, float& ay)
{
#pragma clang fp contract(off)
    if (unicycle) {
        const float rev = (ay + yaw) * 0.15915494309189535f, v = ax;
        ax = v * __builtin_amdgcn_cosf(rev);
        ay = v * __builtin_amdgcn_sinf(rev);
    }
}

// closest approach of one human to the robot over [0, T] with both velocities held (utils.py:22-36), minus the two radii
__device__ __forceinline__ float gym_swept_closest(float hx, float hy, float hvx, float hvy, float hr, float rpx, float rpy, float rr,
                                                   float ax, float ay, float T)
{
#pragma clang fp contract(off)
    const float x1 = hx - rpx, y1 = hy - rpy;
    const float x2 = x1 + (hvx - ax) * T, y2 = y1 + (hvy - ay) * T;
    const float dx = x2 - x1, dy = y2 - y1;
    float qx, qy;
    if (dx == 0.0f && dy == 0.0f) { qx = 0.0f - x1; qy = 0.0f - y1; }
    else {
        float u = ((0.0f - x1) * dx + (0.0f - y1) * dy) / (dx * dx + dy * dy);
        if (u > 1.0f) u = 1.0f; else if (u < 0.0f) u = 0.0f;
        qx = x1 + u * dx; qy = y1 + u * dy;
    }
    return __builtin_amdgcn_sqrtf(qx * qx + qy * qy) - hr - rr;
}

// what the head reads of its world's episode state: loaded by the caller together with its other loads (one memory round trip for all)
struct GymPre { float gtime; int counter, prev, pending; unsigned seed, epoch; };
template <bool FOLD = true>
__device__ __forceinline__ GymPre gym_head_preload(const GymHead& g, int w)
{
    GymPre p;
    p.gtime = g.gtime[w]; p.counter = 0; p.prev = 0; p.pending = 0; p.seed = 0; p.epoch = 0;
    if (g.bk.mode != 0) {
        p.counter = g.bk.counter[w];
        if (g.bk.mode == 2) p.prev = g.bk.prev[w];
        if (FOLD && g.fold.on) { p.pending = g.fold.pending[w]; p.seed = g.bk.seeds[w]; p.epoch = g.fold.epoch[w]; }
    }
    return p;
}

// One lane per world: goes over the humans' swept distances closest[0 .. n) in index order -- the reference's loop with its early `break`,
// written without the break (dmin = the minimum over the humans BEFORE the first collision) so that the n LDS reads are independent of each
// other: round 4's first version waited for every one of them in turn --, writes the reward row and -- bk.mode != 0 -- does the world's bookkeeping.
// Returns true when the world takes over its staged episode at the end of this launch (cs_gym_step_staged; the separate
// cs_consume_staged_worlds launch reads the same decision from the masks).
__device__ __forceinline__ bool gym_head_world(const GymHead& g, int w, int n, const float* closest, float rpx, float rpy, float rr,
                                               float rgx, float rgy, float ax, float ay, const GymPre pre)
{
#pragma clang fp contract(off)
    const GymBook& bk = g.bk;
    // the clock entries the bookkeeping may write (the step after this one, or a fresh episode's): requested before the walk
    float clk0 = 0.0f, clk1 = 0.0f;
    int c1 = 0;
    if (bk.mode != 0) {
        c1 = pre.counter + 1 < bk.clock_len - 1 ? pre.counter + 1 : bk.clock_len - 1;
        clk0 = bk.clock[0]; clk1 = bk.clock[c1];
    }
    float dmin = INFINITY;
    int collision = 0;
    for (int j = 0; j < n; ++j) {
        const float c = closest[j];
        const bool hit = c < 0.0f;
        dmin = (!collision && !hit && c < dmin) ? c : dmin;
        collision |= hit ? 1 : 0;
    }
    const float ex = rpx + ax * g.T, ey = rpy + ay * g.T;
    const float gx = ex - rgx, gy = ey - rgy;
    const int reaching = __builtin_amdgcn_sqrtf(gx * gx + gy * gy) < rr;
    float reward = 0.0f; int term = 0, trunc = 0, info = 0;
    if (pre.gtime >= g.time_limit - 1.0f) { trunc = 1; info = 4; }
    else if (collision) { reward = g.collision_penalty; term = 1; info = 3; }
    else if (reaching) { reward = g.success_reward; term = 1; info = 2; }
    else if (dmin < g.discomfort_dist) { reward = (dmin - g.discomfort_dist) * g.discomfort_factor * g.T; info = 1; }
    float* o = g.out + (long)w * 7;
    o[0] = (float)collision; o[1] = dmin; o[2] = (float)reaching; o[3] = reward;
    o[4] = (float)term; o[5] = (float)trunc; o[6] = (float)info;
    if (bk.mode == 0) return false;
    // the same statements as k_gym_bookkeeping / k_gym_bookkeeping_next_step (robot_model.hip), on the values just written
    if ((bk.mode == 2 && pre.prev) || pre.pending) {   // between two episodes (NEXT_STEP's step after the terminal one; a deferred take-over)
        bk.reward[w] = 0.0f; bk.terminated[w] = 0; bk.truncated[w] = 0; bk.info[w] = 0;
        bk.mask[w] = 0; bk.counter[w] = 0; bk.gtime[w] = clk0;
        return true;
    }
    bk.reward[w] = reward; bk.terminated[w] = term ? 1 : 0; bk.truncated[w] = trunc ? 1 : 0; bk.info[w] = info;
    const bool done = term || trunc;
    int c = c1;                                      // min(counter + 1, clock_len - 1)
    if (bk.mode == 2) {
        bk.mask[w] = done ? 1 : 0;
        if (done) bk.seeds[w] += bk.stride;
    } else if (bk.auto_reset) {
        bk.mask[w] = done ? 1 : 0;
        if (done) { bk.seeds[w] += bk.stride; c = 0; }
    }
    bk.counter[w] = c;
    bk.gtime[w] = c == 0 ? clk0 : clk1;
    return bk.mode != 2 && bk.auto_reset && done;      // same-step rules: the world that ends NOW is replaced before the observation is returned
}

// The head's second half under cs_gym_step_staged, by the lane that ran gym_head_world: `take` = its verdict.  Returns the launch's code for the
// world (the wavefront's epilogue acts on it) and leaves it in pending[w] for the launches to come (1: deferred).
__device__ __forceinline__ int gym_fold_decide(const GymHead& g, int w, bool take, bool ended_now, const GymPre pre)
{
    const GymFold& f = g.fold;
    int code = 0;
    if (take) {
        const unsigned want = pre.seed + (ended_now ? g.bk.stride : 0u);              // the seed the bookkeeping has moved this world to
        const long slot = (long)((pre.epoch + 1u) & (unsigned)(f.depth - 1)) * f.W + w;
        const unsigned tag = __hip_atomic_load(f.staged_seed + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        code = tag == want ? (int)slot + 2 : 1;
        // a staged episode the generator could not build (its bounded rejection sampling gave up: status != 0, stored before the tag's release)
        // is NOT taken over: the world keeps its stepped rows and a fresh observation -- what cs_gym_step + cs_consume_staged_worlds leave -- and
        // only the slot's turn is consumed (epoch, failed).  Code -(slot + 2): the wavefront writes its rows as usual.
        if (code >= 2 && __hip_atomic_load(f.staged_status + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) code = -code;
    }
    if (code != 0 || pre.pending != 0) __hip_atomic_store(f.pending + w, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return code;   // (the launch's own epilogue takes it from here, through LDS: no round trip to pending[w])
}

} // namespace cstep
