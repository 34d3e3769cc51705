// crowdstep.hip -- hand-written CDNA4 (gfx950) kernels + the C ABI of include/crowdstep.h.
//
// Hot path: the reference's per-substep pedestrian update
//   update_humans_parallel            /root/reference/social_gym/src/forces_parallel.py:185-284
//   MotionModelManager.update_humans  /root/reference/social_gym/src/motion_model_manager.py:354-422
//   SocialNavGym.step substep loop    /root/reference/social_gym/social_nav_gym.py:240-245
// re-designed for MI355X: lane = agent row, floor(64/rows) independent worlds per wavefront, the
// interacting columns (px, py, r + safety; velocities where a model needs them) staged in LDS, every
// unordered pair evaluated once with the reaction handed over through in-order LDS read-modify-writes
// (or all partners per lane for per-agent parameters / worlds of more than 64 rows), all substeps of
// one Gym step fused in one launch with the state in registers, so HBM sees each state row once per
// launch.  No MFMA (there is no dense contraction on this path).  Design notes: DESIGN.md §4.1.
//
// gfx950 only: no portability macros, no CPU fallback.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>
#include <type_traits>
#include <utility>

#include "common.h"
#include "crowdstep.h"
#include "sfmstep_kernel.h"
#include "stepcommon.h"

namespace csimpl {
thread_local std::string g_err;

// Library-owned device scratch, keyed by (device, stream, use): two batches on different streams never share a block
// (work on ONE stream is ordered, so reuse there is safe).  Capture rules: nothing is allocated or freed while `stream` is
// capturing -- a block that would have to be allocated or grown then is refused with CS_ERR_ARG (reserve it first:
// cs_reserve_scratch, or one call of the entry point outside the capture); a block handed out during a capture is PINNED:
// a graph holds its address, so growing it later retires the old block instead of freeing it (cs_release_scratch frees all).
// Threads: ONE host thread per (device, stream) -- the mutex guards the table, not the block: two threads launching on the same
// stream could grow (free) a block the other is about to launch into.  Different streams never share a block.
namespace {
struct ScratchBlock { void* p = nullptr; size_t cap = 0; bool pinned = false; };
std::mutex g_scratch_mu;
std::map<std::tuple<int, hipStream_t, int>, ScratchBlock> g_scratch;
std::vector<std::pair<int, void*>> g_retired;
}

int scratch(void** out, size_t bytes, int slot, hipStream_t stream)
{
    *out = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(CS_ERR_HIP, "hipGetDevice failed");
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (stream != nullptr && hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
    const bool capturing = st == hipStreamCaptureStatusActive;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    ScratchBlock& b = g_scratch[std::make_tuple(dev, stream, slot)];
    if (b.p && b.cap >= bytes) { b.pinned = b.pinned || capturing; *out = b.p; return CS_OK; }
    if (capturing)
        return fail(CS_ERR_ARG, "crowdstep needs " + std::to_string(bytes) + " bytes of library scratch on this stream, which cannot be allocated while "
                                "the stream is capturing: call cs_reserve_scratch (or this entry point once) on the stream before the capture");
    if (b.p) {
        if (b.pinned) g_retired.emplace_back(dev, b.p);     // a captured graph may still replay into it
        else (void)hipFree(b.p);                            // (hipFree synchronises: earlier users of the old block are done)
        b = ScratchBlock{};
    }
    const hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) { b = ScratchBlock{}; return fail(CS_ERR_HIP, std::string("hipMalloc of crowdstep scratch: ") + hipGetErrorString(e)); }
    b.cap = bytes;
    *out = b.p;
    return CS_OK;
}

int scratch_release_all()
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    int dev0 = 0;
    (void)hipGetDevice(&dev0);
    // every device that holds a block finishes its work first: a kernel or a graph still running THERE may be using one
    // (cs_release_scratch used to synchronise the calling thread's current device only)
    std::set<int> devs;
    for (auto& kv : g_scratch) if (kv.second.p) devs.insert(std::get<0>(kv.first));
    for (auto& r : g_retired) devs.insert(r.first);
    for (int d : devs) { (void)hipSetDevice(d); (void)hipDeviceSynchronize(); }
    for (auto& kv : g_scratch)
        if (kv.second.p) { (void)hipSetDevice(std::get<0>(kv.first)); (void)hipFree(kv.second.p); }
    g_scratch.clear();
    for (auto& r : g_retired) { (void)hipSetDevice(r.first); (void)hipFree(r.second); }
    g_retired.clear();
    (void)hipSetDevice(dev0);
    return CS_OK;
}
// SIMDs of the current device (CUs x 4): what "waves per SIMD" of a grid is measured against (MI355X: 256 x 4; fewer on a
// partitioned device)
int device_simds()
{
    static thread_local int cached_dev = -1, cached = 1024;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024;
    if (dev != cached_dev) {
        int cus = 0;
        cached = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus * 4 : 1024;
        cached_dev = dev;
    }
    return cached;
}
#ifdef CS_STAMPS
unsigned long long* g_stamp_buf = nullptr;
#endif
}

namespace {

using csimpl::fail;
using csimpl::g_err;

using namespace cstep;

// ------------------------------------------------------------------------------------------
// collision / reward: social_nav_sim.py:949-1029, utils.py:22-36.  One lane per world.
// ------------------------------------------------------------------------------------------
__global__ void k_collision_reward(int W, int n, int rows, const float* S, long as, long fs, const float* robot,
                                   const float* action, float T, const float* gtime, float time_limit,
                                   float success_reward, float collision_penalty, float discomfort_dist,
                                   float discomfort_factor, float* out, int unicycle)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    const float* rb = robot + (long)w * 13;
    const float rpx = rb[0], rpy = rb[1], rr = rb[8], rgx = rb[10], rgy = rb[11];
    float ax = action[(long)w * 2], ay = action[(long)w * 2 + 1];
    gym_action_velocity(unicycle, rb[2], ax, ay);
    float dmin = INFINITY;
    int collision = 0;
    for (int i = 0; i < n; ++i) {
        const float* s = S + ((long)w * rows + i) * as;
        const float closest = gym_swept_closest(s[0], s[fs], s[3 * fs], s[4 * fs], s[8 * fs], rpx, rpy, rr, ax, ay, T);
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

// The same, one lane per (world, human) for n <= 64: the humans' swept distances are evaluated in parallel (the rows are
// read once, coalesced), then one lane per world walks them in index order with the reference's early `break`.
// episode bookkeeping of a vectorised Gym step (cs_gym_bookkeeping / _next_step, robot_model.hip), done by the lane that wrote the
// world's reward row when cs_collision_reward_gym asks for it (mode 0: none, 1: same-step rules, 2: NEXT_STEP rules)
__global__ __launch_bounds__(64) void k_collision_reward_wave(int W, int n, int rows, int wpb, const float* S, long as, long fs,
                                                              const float* robot, const float* action, const GymHead g)
{
    __shared__ float s_closest[64];
    const int tid = threadIdx.x;
    const int lw = tid / n, i = tid - lw * n;
    const int w = blockIdx.x * wpb + lw;
    const bool valid = lw < wpb && w < W;
    float rpx = 0, rpy = 0, rr = 0, rgx = 0, rgy = 0, ax = 0, ay = 0;
    float closest = INFINITY;
    cstep::GymPre pre = {0.0f, 0, 0, 0, 0u, 0u};
    if (valid) {
        if (i == 0) pre = gym_head_preload(g, w);
        const float* rb = robot + (long)w * 13;
        rpx = rb[0]; rpy = rb[1]; rr = rb[8]; rgx = rb[10]; rgy = rb[11];
        ax = action[(long)w * 2]; ay = action[(long)w * 2 + 1];
        gym_action_velocity(g.unicycle, rb[2], ax, ay);
        const float* s = S + ((long)w * rows + i) * as;
        closest = gym_swept_closest(s[0], s[fs], s[3 * fs], s[4 * fs], s[8 * fs], rpx, rpy, rr, ax, ay, g.T);
    }
    s_closest[tid] = closest;
    __syncthreads();
    if (!valid || i != 0) return;
    gym_head_world(g, w, n, s_closest + lw * n, rpx, rpy, rr, rgx, rgy, ax, ay, pre);
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
#ifdef CS_STAMPS
using csimpl::g_stamp_buf;
#endif

struct Geometry { int grid, block, wpb, ws; };

int geometry(const cs_worlds* w, Geometry& g)
{
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    if (rows <= 0) return fail(CS_ERR_ARG, "rows per world must be positive");
    if (rows > csimpl::big_world_min_rows(1024)) { g.block = 256; g.wpb = 1; g.ws = 0; g.grid = ((rows + 255) / 256) * w->W; return CS_OK; }   // grid path (bigworld.hip)
    if (rows <= 64) { g.block = 64; g.wpb = 64 / rows; }
    else { g.block = ((rows + 63) / 64) * 64; g.wpb = 1; }
    // world pitch in the doubled LDS buffers: 2 * rows, padded so that row index == lane (mod 32) in every world of the
    // wavefront (bank-conflict-free b64 / b96 / b128 accesses across the world boundary) when that fits the 2T rows
    g.ws = 2 * rows;
    if (g.block == 64) {
        const int padded = 2 * rows + (32 - rows % 32) % 32;
        if (g.wpb * padded <= 2 * g.block) g.ws = padded;
    }
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

// Which instantiation of k_sfm_step a launch runs.  One function decides it for launch_step and for cs_step_variant (the
// diagnostic entry the parity tests use to assert that every benched build is the one they compared with the oracle).
struct RobotModel { int type; const float* params; float margin; const float* d_human_margin; float* d_memory; };

Variant select_variant(const cs_worlds* w, int mode, const Geometry& g, bool need_snap = false, bool robot_model = false)
{
    const bool robot = (w->flags & CS_ROBOT_ROW) != 0;
    const int rows = w->n + (robot ? 1 : 0);
    const bool peq = (w->flags & CS_ALL_PARAMS_EQUAL) != 0;
    if (g.block != 64) return Variant{1024, 1, 0, 0, peq};
    // more than two one-wave blocks per SIMD -> the 4-waves-per-SIMD register budget pays (SIMD count asked from the device: a
    // partitioned (CPX) device has fewer)
    const bool crowded = g.grid > 2 * csimpl::device_simds();
    // The plain crowd batch (what Gym scenarios and the bench step): all_params_equal -> every pair once, <= 2 goal slots, committed
    // in place -> a LEAN build: 1 = no walls, no robot row; 2 = walls; 3 = no walls + the robot as the last row (a Gym with a
    // visible robot: cs_step hands the robot rows over through d_robot, M_ROBOT_FROM_ARRAY); 5 = walls + robot row.  Everything else (
    // per-agent parameters, longer goal lists, peek / out-of-place modes) runs the generic build.
    const bool lean_mode = robot ? (mode & ~(int)M_ROBOT_FROM_ARRAY) == M_COMMIT_GOALS : mode == M_COMMIT_GOALS;
    int kind = 0;
    if (peq && w->G <= 2 && lean_mode) kind = robot ? (w->O == 0 ? 3 : 5) : (w->O == 0 ? 1 : 2);
    if (robot_model) {   // the robot's own motion model inside the launch: the LEAN = 4 builds only (callers check fusable_imitation)
        if (kind != 3) return Variant{0, 0, 0, 0, peq};
        return Variant{64, rows == 26 ? 1 : 3, rows == 26 ? 26 : 0, 4, true};
    }
    // per-agent parameters, 25 rows, Helbing / Guo: the compile-time pair-once build with the partners' parameter rows in registers
    // (sfmstep_peragent.hip; one spill-free register budget for every grid)
    if (kind == 0 && !peq && rows == 25 && w->type % 3 != 2 && !need_snap) return Variant{64, 1, 25, 0, false};
    if (kind == 0) return Variant{64, 3, 0, 0, peq};
    // small plain worlds (10 humans: BASELINE.json configs[1]; 5: the reference's default environment): one world per 16-lane
    // DPP row, partners exchanged with row shifts instead of LDS (rowstep.hip).  CROWDSTEP_ROW16=0 keeps them on the LDS kernel.
    const char* row16_env = std::getenv("CROWDSTEP_ROW16");
    const bool row16_on = !(row16_env && row16_env[0] == '0');
    if (kind == 1 && row16_on && !need_snap && csimpl::row16_supports(rows)) return Variant{16, 2, rows, 1, true};
    // Compile-time row counts (partner groups laid out at compile time: no loop, no scalar branches): the BASELINE.json
    // configurations (10, 25, 50 humans) and the shapes around them -- 20 and 30 humans, and each with the visible robot's extra
    // row (6, 11, 26, 51).  Register budgets: 25 / 26 / 30 / 20 rows = OCC 1 (spill-free, ~136 VGPRs) while the grid holds at most
    // two waves per SIMD, OCC 4 (128 VGPRs, a few spills) beyond; 10 / 11 / 6 rows fit 128 VGPRs without spills; 50 / 51 rows
    // unrolled spill 30-47 VGPRs at the 4-wave budget and take the three-wave one (168 VGPRs).
    struct Row { int kind, rows, occ_sparse, occ_crowded; };
    static const Row table[] = {{1, 25, 1, 4}, {1, 30, 1, 4}, {1, 20, 1, 4}, {1, 10, 4, 4}, {1, 50, 3, 3}, {2, 50, 3, 3},
                                {3, 26, 1, 4}, {3, 6, 4, 4}, {3, 11, 4, 4}, {3, 51, 3, 3}};
    for (const Row& r : table)
        if (r.kind == kind && r.rows == rows) return Variant{64, crowded ? r.occ_crowded : r.occ_sparse, rows, kind, true};
    // any other row count: the run-time partner loop keeps the budget of THREE waves per SIMD on every grid: its 128-VGPR build
    // spills (7 VGPRs lean, 35 with walls) and measured 6-17 % slower on crowded grids (8192 x 50 + walls: 355 vs 304 us;
    // 16384 x 30 Moussaid: 484 vs 402 us); the walls build needs exactly 168 VGPRs, and a build that tips over to 169 runs at
    // two waves per SIMD and 18 % slower -- with the cap a future compiler spills a register instead
    return Variant{64, 3, 0, kind, true};
}

kfn variant_kernel(const Variant& v, int type)
{
    for (auto lookup : {sfm_builds_generic, sfm_builds_leanrt, sfm_builds_lean25, sfm_builds_lean30, sfm_builds_small, sfm_builds_lean50, sfm_builds_robot26, sfm_builds_robotx, sfm_builds_imit, sfm_builds_peragent})
        if (kfn fn = lookup(v, type)) return fn;
    return nullptr;
}

// the Gym head's arguments from the C ABI's (book == nullptr: reward row only)
GymHead gym_head(float* d_out, const float* d_global_time, float T, const float* reward_cfg, const cs_gym_book* book, int W, int flags)
{
    GymHead g;
    std::memset(&g, 0, sizeof(g));
    g.unicycle = (flags & CS_ROBOT_UNICYCLE) ? 1 : 0;
    g.out = d_out; g.gtime = d_global_time; g.T = T;
    g.time_limit = reward_cfg[0]; g.success_reward = reward_cfg[1]; g.collision_penalty = reward_cfg[2];
    g.discomfort_dist = reward_cfg[3]; g.discomfort_factor = reward_cfg[4];
    if (book) {
        GymBook& bk = g.bk;
        bk.mode = book->d_prev_mask ? 2 : 1; bk.clock_len = book->clock_len; bk.auto_reset = book->auto_reset;
        bk.stride = book->seed_stride ? book->seed_stride : (unsigned)W;
        bk.counter = book->d_counter; bk.seeds = book->d_seeds; bk.mask = book->d_mask; bk.prev = book->d_prev_mask; bk.gtime = const_cast<float*>(d_global_time);   // (with a book the caller's buffer is writable: cs_collision_reward_gym)
        bk.clock = book->d_clock; bk.reward = book->d_reward; bk.terminated = book->d_terminated; bk.truncated = book->d_truncated;
        bk.info = book->d_info;
    }
    return g;
}

// Dynamic LDS of a k_sfm_step block, and the two kernel arguments that follow from it (one function for the launch and for
// cs_step_variant, which reports the figure: the blocks a CU holds are decided by it -- sixteen of the 25-row build, twelve of the 50-row
// wall build -- and a region added for one build on every launch costs the others a block without any test noticing; tests/test_gpu_parity.py
// asserts the figures of the benched builds).
// independent one-wavefront blocks per workgroup of the step kernels (sfmstep_kernel.h); CROWDSTEP_WG_WAVES = 1 / 2 / 4 for A/B, read once
int step_wg_waves();
// ... as launched for a block of `shmem` bytes of dynamic LDS: halved until the workgroup stays within the default 64 KB dynamic-LDS limit
// (one helper for launch_step and cs_step_variant: the reported width is the launched one)
int step_wg_waves_for(size_t shmem)
{
    int wg = step_wg_waves();
    const size_t per_wave = (shmem + 15) & ~(size_t)15;
    while (wg > 1 && per_wave * (size_t)wg > 64 * 1024) wg /= 2;
    return wg;
}
int step_wg_waves()
{
    static const int v = []{ const char* e = std::getenv("CROWDSTEP_WG_WAVES"); const int x = e ? std::atoi(e) : 4; return (x == 1 || x == 2 || x == 4) ? x : 4; }();
    return v;
}

size_t step_lds_bytes(const cs_worlds* w, const Geometry& g, bool peq, int* seg_tab_out, int* wall_pairs_out)
{
    struct { int seg_tab, wall_pairs; } a = {0, 0};
    // lds_p [2][2T+PADR] float4, lds_v [2][2T+PADR] float2, lds_vr [2][T] float2, respawn scratch 2 x [T] x 4 B,
    // reaction accumulators [UA][2T] float2 (pair-once loop: all_params_equal, block of one wavefront)
    size_t shmem = (size_t)g.block * (4 * sizeof(float4) + 4 * sizeof(float2) + 2 * sizeof(float2) + 2 * sizeof(float)) +
                   2 * PADR * (sizeof(float4) + sizeof(float2)) +
                   ((peq && g.block == 64) ? (size_t)UA * ACC_PITCH * sizeof(float2) : 0);
    // per-agent parameters on the pair-once loop (Helbing / Guo, block of one wavefront): reaction accumulators + the partners' parameter rows
    if (!peq && g.block == 64 && w->type % 3 != 2)
        shmem += (size_t)UA * ACC_PITCH * sizeof(float2) + (size_t)(2 * g.block + PADR) * sizeof(float4);
    // wall segment table (x1, y1, e, 1/|e|^2), shared or one per world of the block, when it is small enough
    const long seg_tab = (long)w->O * w->Smax * ((w->flags & CS_OBSTACLES_SHARED) ? 1 : g.wpb);
    a.seg_tab = (seg_tab > 0 && seg_tab * 20 <= 16 * 1024) ? (int)seg_tab : 0;
    shmem += (size_t)a.seg_tab * (sizeof(float4) + sizeof(float)) + 16 + (size_t)(a.seg_tab > 0 ? a.seg_tab / w->Smax : 0) * sizeof(float4);
    {
        // one (agent, polygon) pair per lane: Helbing-type walls (no tangential term outside a contact), at most 4 polygons staged in LDS, no
        // respawn rule (the only way an agent jumps); CROWDSTEP_WALL_PAIRS=0 keeps every launch on the all-lanes pass (A/B)
        static const bool wp_env = []{ const char* e = std::getenv("CROWDSTEP_WALL_PAIRS"); return !(e && e[0] == '0'); }();
        const bool guo_walls = w->type == 1 || w->type == 4 || w->type == 7;
        a.wall_pairs = (wp_env && peq && a.seg_tab > 0 && a.seg_tab < 4096 && w->O > 0 && w->O <= 4 && !guo_walls && !(w->flags & CS_RESPAWN) && g.block == 64) ? 1 : 0;
    }
    // wall pairs (sfmstep_kernel.h): the pairs' records and forces, the wall law -- the LAST region of the block's LDS, only where it is used
    // (on every launch it cost the 25-row build its sixteenth block per CU: 8192 worlds 51 -> 60 us, 32768 worlds 160 -> 181 us)
    if (a.wall_pairs) shmem += 128 * sizeof(int) + 130 * sizeof(float2) + sizeof(float4);
    if (seg_tab_out) *seg_tab_out = a.seg_tab;
    if (wall_pairs_out) *wall_pairs_out = a.wall_pairs;
    return shmem;
}

int launch_step(const cs_worlds* w, float dt, int nsub, int mode, float* d_out, const float* d_action,
                float* d_peek, hipStream_t stream, float4* d_snap = nullptr, float* d_trace = nullptr, const RobotModel* rm = nullptr,
                float* d_obs = nullptr, int obs_cols = 0, const GymHead* gym = nullptr)
{
    int rc = check_worlds(w);
    if (rc) return rc;
    Geometry g;
    rc = geometry(w, g);
    if (rc) return rc;
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    if (rows > csimpl::big_world_min_rows(1024))   // worlds beyond one block: partners through a uniform grid in HBM (bigworld.hip)
        return csimpl::sfm_big_launch(w, dt, nsub, d_out ? d_out : w->d_state, (mode & M_MUTATE_INPUT) ? 1 : 0,
                                      (mode & M_ROBOT_FROM_ARRAY) != 0, d_action, (mode & M_PEEK) ? d_peek : nullptr, stream, d_trace);
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = w->W; a.n = w->n; a.rows = rows; a.G = w->G; a.O = w->O; a.Smax = w->Smax;
    a.type = w->type; a.flags = w->flags; a.mode = mode; a.nsub = nsub; a.wpb = g.wpb; a.ws = g.ws; a.dt = dt;
    a.Sin = w->d_state; a.Sout = d_out ? d_out : w->d_state;
    strides(w, rows, a.in_as, a.in_fs);
    a.out_as = a.in_as; a.out_fs = a.in_fs;
    a.goals = w->d_goals; a.params = w->d_params; a.safety = w->d_safety; a.obstacles = w->d_obstacles;
    a.robot = w->d_robot; a.action = d_action; a.peek_out = d_peek;
    a.bx = w->respawn_bound_x; a.by = w->respawn_bound_y;
    a.world_flags = w->d_world_flags;
    {   // The reach of a wall's force: beyond 22 e-folding lengths it is below |A| e^-22 = 5.6e-7 N for the reference's A = 2000 N -- less than
        // the float32 ulp of a 10 N force sum (9.5e-7 N; an agent near a wall carries tens of newtons), 9e-11 m/s on a velocity per substep
        // (dt / m = 1.7e-4 s/kg) against north_star's 1e-5.  Rounds 2 - 4 used 36 (5e-13 N); in cfg5's bench window the crowd stands among
        // the polygons and 36 lengths = 2.9 m reach two thirds of all (agent, polygon) pairs, 22 = 1.8 m about half
        // (profiles/r5e_wall_pairs_ab.txt).  CROWDSTEP_WALL_EFOLDS overrides it.
        const char* e = std::getenv("CROWDSTEP_WALL_EFOLDS");
        a.wall_efolds = e ? (float)std::atof(e) : 22.0f;
    }
#ifdef CS_STAMPS
    a.stamps = g_stamp_buf;
#endif
    a.snap = d_snap;
    a.trace = d_trace;
    a.obs = d_obs; a.obs_cols = obs_cols;
    const Variant v = select_variant(w, mode, g, d_snap != nullptr, rm != nullptr);
    if (gym) {
        if (v.maxt != 64) return fail(CS_ERR_ARG, "the Gym head runs inside the step launch of blocks of one wavefront only");
        a.gym = *gym;
    }
    if (rm) {
        if (v.lean != 4) return fail(CS_ERR_ARG, "the robot's motion model runs inside the crowd's launch only for the plain crowd batch with a robot row");
        a.rm_type = rm->type; a.rm_margin = rm->margin; a.rm_hmargin = rm->d_human_margin; a.rm_memory = rm->d_memory;
        std::memcpy(a.rm_P, rm->params, sizeof(a.rm_P));
    }
    const bool peq = v.peq;
    if (v.maxt == 16) return csimpl::row16_launch(a, stream);
    const kfn fn = variant_kernel(v, w->type);
    if (!fn) return fail(CS_ERR_ARG, "no kernel build for this variant");
    const size_t shmem = step_lds_bytes(w, g, peq, &a.seg_tab, &a.wall_pairs);
    if (shmem > 64 * 1024) // one world per block with > ~600 rows
        HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    // two one-wavefront blocks per SIMD (the benchmark's 4096 x 25): the second half of the grid shares each SIMD with an older wavefront
    a.young_from = (g.block == 64 && g.grid == 2 * csimpl::device_simds()) ? g.grid / 2 : 0x7fffffff;
    a.wg_waves = 1; a.lds_per_wave = 0;
    if (g.block == 64) {
        // one-wavefront builds: wg_waves independent wavefronts per workgroup (sfmstep_kernel.h)
        a.lds_per_wave = (int)((shmem + 15) & ~(size_t)15);
        a.wg_waves = step_wg_waves_for(shmem);
        hipLaunchKernelGGL(fn, dim3((g.grid + a.wg_waves - 1) / a.wg_waves), dim3(64 * a.wg_waves), (size_t)a.lds_per_wave * a.wg_waves, stream, a);
        HIP_TRY(hipGetLastError());
        return CS_OK;
    }
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
int cs_abi_version(void) { return CS_ABI_VERSION; }   // 2: seed_stride in cs_gym_book / cs_gym_bookkeeping*, cs_stage_book, cs_event_query, cs_device_pci_bus_id; 3: cs_orca_set_math / cs_orca_get_math; 4: cs_worlds.orca_math replaces them (cs_orca_default_math)

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

int cs_device_pci_bus_id(int device, char* buf, size_t buflen)
{
    if (!buf || buflen < 16) return fail(CS_ERR_ARG, "buffer of at least 16 bytes needed");
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int)buflen, device));
    return CS_OK;
}

int cs_malloc(void** d_ptr, size_t bytes) { if (!d_ptr) return fail(CS_ERR_ARG, "null out pointer"); HIP_TRY(hipMalloc(d_ptr, bytes)); return CS_OK; }
int cs_free(void* d_ptr) { HIP_TRY(hipFree(d_ptr)); return CS_OK; }
int cs_memcpy_h2d(void* d, const void* h, size_t bytes, void* s) { HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)s)); if (!s) HIP_TRY(hipStreamSynchronize(nullptr)); return CS_OK; }
int cs_memcpy_d2h(void* h, const void* d, size_t bytes, void* s) { HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)s)); HIP_TRY(hipStreamSynchronize((hipStream_t)s)); return CS_OK; }
int cs_memcpy_d2d(void* dd, const void* ds, size_t bytes, void* s) { HIP_TRY(hipMemcpyAsync(dd, ds, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s)); return CS_OK; }
int cs_memset(void* d, int value, size_t bytes, void* s) { HIP_TRY(hipMemsetAsync(d, value, bytes, (hipStream_t)s)); return CS_OK; }
int cs_stream_create(void** s) { if (!s) return fail(CS_ERR_ARG, "null out pointer"); hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); *s = st; return CS_OK; }
int cs_stream_create_with_priority(void** s, int priority)
{
    if (!s) return fail(CS_ERR_ARG, "null out pointer");
    int least = 0, greatest = 0;   // (numerically: least priority = the larger number)
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int p = priority == 0 ? 0 : (priority < 0 ? greatest : least);
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, p));
    *s = st;
    return CS_OK;
}
int cs_stream_destroy(void* s) { HIP_TRY(hipStreamDestroy((hipStream_t)s)); return CS_OK; }
int cs_stream_sync(void* s) { HIP_TRY(hipStreamSynchronize((hipStream_t)s)); return CS_OK; }
int cs_event_create(void** e) { if (!e) return fail(CS_ERR_ARG, "null out pointer"); hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); *e = ev; return CS_OK; }
int cs_event_destroy(void* e) { HIP_TRY(hipEventDestroy((hipEvent_t)e)); return CS_OK; }
int cs_event_query(void* e, int* done)
{
    if (!e || !done) return fail(CS_ERR_ARG, "null argument");
    const hipError_t rc = hipEventQuery((hipEvent_t)e);
    if (rc != hipSuccess && rc != hipErrorNotReady) HIP_TRY(rc);
    *done = rc == hipSuccess ? 1 : 0;
    return CS_OK;
}
int cs_event_record(void* e, void* s) { HIP_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); return CS_OK; }
int cs_stream_wait_event(void* s, void* e) { HIP_TRY(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0)); return CS_OK; }
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
    if (w->type == CS_ORCA) return csimpl::orca_launch(w, dt, n_substeps, d_action, nullptr, (hipStream_t)stream);
    if (w->type == CS_SOCIAL_MOMENTUM) return csimpl::social_momentum_launch(w, dt, n_substeps, d_action, nullptr, (hipStream_t)stream);
    int mode = M_COMMIT_GOALS;
    if ((w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
    return launch_step(w, dt, n_substeps, mode, nullptr, d_action, nullptr, (hipStream_t)stream);
}

int cs_step_observe(const cs_worlds* w, float dt, int n_substeps, const float* d_action, int theta_and_omega_visible, float* d_obs,
                    void* stream)
{
    if (!w || !d_obs) return fail(CS_ERR_ARG, "null argument");
    if (n_substeps <= 0) return fail(CS_ERR_ARG, "n_substeps must be positive");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    // the SFM / HSFM kernels of one block write the observation from their registers; everything else steps, then reads it back
    if (w->type < 0 || w->type > 8 || rows > csimpl::big_world_min_rows(1024)) {
        const int rc = cs_step(w, dt, n_substeps, d_action, stream);
        return rc ? rc : cs_gym_observe(w, theta_and_omega_visible, d_obs, stream);
    }
    if (d_action && !w->d_robot) return fail(CS_ERR_ARG, "robot action given but cs_worlds.d_robot is null");
    int mode = M_COMMIT_GOALS;
    if ((w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
    return launch_step(w, dt, n_substeps, mode, nullptr, d_action, nullptr, (hipStream_t)stream, nullptr, nullptr, nullptr, d_obs,
                       theta_and_omega_visible ? 7 : 5);
}

int cs_gym_step_is_one_launch(const cs_worlds* w)
{
    // ONE launch where the step kernel is the LDS kernel of one wavefront per block (SFM / HSFM worlds of up to 64 rows); the two
    // launches everywhere else (ORCA, social momentum, the DPP-row kernel's small worlds keep their own launch: same results)
    if (!w) return 0;
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    bool fused = w->type >= 0 && w->type <= 8 && rows <= 64 && w->W > 0 && w->n > 0;
    if (fused) {
        if (check_worlds(w)) fused = false;
        else {
            Geometry g;
            if (geometry(w, g)) fused = false;
            else {
                int mode = M_COMMIT_GOALS;
                if ((w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
                fused = select_variant(w, mode, g).maxt == 64;   // (small plain worlds keep the DPP-row kernel and its own reward launch)
            }
        }
    }
    if (!fused) return 0;
    // 2: cs_gym_step_staged as well (the take-over in the epilogue is compiled into the builds without walls: sfmstep_kernel.h FOLD)
    return w->O > 0 ? 1 : 2;
}

int cs_gym_step(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float T, float* d_global_time, const float* reward_cfg,
                float* d_out, const cs_gym_book* book, int theta_and_omega_visible, float* d_obs, void* stream)
{
    if (!w || !book || !d_obs) return fail(CS_ERR_ARG, "null argument");
    if (n_substeps <= 0) return fail(CS_ERR_ARG, "n_substeps must be positive");
    const bool fused = cs_gym_step_is_one_launch(w) != 0;
    if (!fused) {
        const int rc = cs_collision_reward_gym(w, d_action, T, d_global_time, reward_cfg, d_out, book, stream);
        return rc ? rc : cs_step_observe(w, dt, n_substeps, d_action, theta_and_omega_visible, d_obs, stream);
    }
    if (!d_action || !d_global_time || !reward_cfg || !d_out || !w->d_robot) return fail(CS_ERR_ARG, "null argument");
    if (book->clock_len <= 0 || !book->d_counter || !book->d_seeds || !book->d_mask || !book->d_clock || !book->d_reward ||
        !book->d_terminated || !book->d_truncated || !book->d_info)
        return fail(CS_ERR_ARG, "null buffer in cs_gym_book");
    int mode = M_COMMIT_GOALS;
    if ((w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
    const GymHead gh = gym_head(d_out, d_global_time, T, reward_cfg, book, w->W, w->flags);
    return launch_step(w, dt, n_substeps, mode, nullptr, d_action, nullptr, (hipStream_t)stream, nullptr, nullptr, nullptr, d_obs,
                       theta_and_omega_visible ? 7 : 5, &gh);
}

int cs_gym_step_staged(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float T, float* d_global_time, const float* reward_cfg,
                       float* d_out, const cs_gym_book* book, int theta_and_omega_visible, float* d_obs, const cs_generator* gen,
                       const cs_worlds* staging, const cs_stage_book* stage_book, void* stream)
{
    if (!w || !book || !d_obs || !gen || !staging || !stage_book) return fail(CS_ERR_ARG, "null argument");
    if (n_substeps <= 0) return fail(CS_ERR_ARG, "n_substeps must be positive");
    if (!stage_book->d_pending || !stage_book->d_failed) return fail(CS_ERR_ARG, "cs_gym_step_staged needs cs_stage_book.d_pending and d_failed");
    if (!book->auto_reset && !book->d_prev_mask) return fail(CS_ERR_ARG, "cs_gym_step_staged is the auto-reset step (cs_gym_book.auto_reset or the NEXT_STEP masks)");
    if (cs_gym_step_is_one_launch(w) != 2) return fail(CS_ERR_ARG, "cs_gym_step_staged: these worlds take the two launches (cs_gym_step, then cs_consume_staged_worlds)");
    if (!d_action || !d_global_time || !reward_cfg || !d_out || !w->d_robot) return fail(CS_ERR_ARG, "null argument");
    if (book->clock_len <= 0 || !book->d_counter || !book->d_seeds || !book->d_mask || !book->d_clock || !book->d_reward ||
        !book->d_terminated || !book->d_truncated || !book->d_info)
        return fail(CS_ERR_ARG, "null buffer in cs_gym_book");
    if (book->d_seeds != stage_book->d_seeds) return fail(CS_ERR_ARG, "cs_gym_book.d_seeds and cs_stage_book.d_seeds must be one buffer");
    int mode = M_COMMIT_GOALS;
    if ((w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
    GymHead gh = gym_head(d_out, d_global_time, T, reward_cfg, book, w->W, w->flags);
    const int rc = csimpl::stage_fold(gen, staging, w, stage_book, theta_and_omega_visible ? 7 : 5, d_obs, gh.fold);
    if (rc) return rc;
    return launch_step(w, dt, n_substeps, mode, nullptr, d_action, nullptr, (hipStream_t)stream, nullptr, nullptr, nullptr, d_obs,
                       theta_and_omega_visible ? 7 : 5, &gh);
}

int cs_step_trace(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float* d_trace, void* stream)
{
    if (!w || !d_trace) return fail(CS_ERR_ARG, "null argument");
    if (n_substeps <= 0) return fail(CS_ERR_ARG, "n_substeps must be positive");
    if (d_action && !w->d_robot) return fail(CS_ERR_ARG, "robot action given but cs_worlds.d_robot is null");
    if (w->type < 0 || w->type > 8) return fail(CS_ERR_ARG, "cs_step_trace covers the SFM / HSFM models (types 0..8)");
    int mode = M_COMMIT_GOALS;
    if ((w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
    return launch_step(w, dt, n_substeps, mode, nullptr, d_action, nullptr, (hipStream_t)stream, nullptr, d_trace);
}

int cs_peek(const cs_worlds* w, float dt, float* d_next, void* stream)
{
    if (!w || !d_next) return fail(CS_ERR_ARG, "null argument");
    if (w->type == CS_ORCA) return csimpl::orca_launch(w, dt, 1, nullptr, d_next, (hipStream_t)stream);
    if (w->type == CS_SOCIAL_MOMENTUM) return csimpl::social_momentum_launch(w, dt, 1, nullptr, d_next, (hipStream_t)stream);
    int mode = M_PEEK;
    if ((w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
    cs_worlds ww = *w;
    ww.flags &= ~CS_RESPAWN; // update_humans(0, dt, post_update=False), motion_model_manager.py:705
    return launch_step(&ww, dt, 1, mode, nullptr, nullptr, d_next, (hipStream_t)stream);
}

int cs_collision_reward(const cs_worlds* w, const float* d_action, float T, const float* d_global_time,
                        const float* reward_cfg, float* d_out, void* stream)
{
    // the swept test only reads positions, velocities and radii: every crowd model (0..8, ORCA, social momentum) qualifies
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (w->W <= 0 || w->n <= 0 || !w->d_state) return fail(CS_ERR_ARG, "bad cs_worlds");
    if (w->layout != CS_LAYOUT_AOS && w->layout != CS_LAYOUT_SOA) return fail(CS_ERR_ARG, "bad layout");
    if (!d_action || !d_global_time || !reward_cfg || !d_out || !w->d_robot) return fail(CS_ERR_ARG, "null argument");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    long as, fs;
    strides(w, rows, as, fs);
    if (w->n <= 64) {
        const int wpb = 64 / w->n, grid = (w->W + wpb - 1) / wpb;
        hipLaunchKernelGGL(k_collision_reward_wave, dim3(grid), dim3(64), 0, (hipStream_t)stream, w->W, w->n, rows, wpb,
                           (const float*)w->d_state, as, fs, (const float*)w->d_robot, d_action,
                           gym_head(d_out, d_global_time, T, reward_cfg, nullptr, w->W, w->flags));
    } else {
        const int block = 64, grid = (w->W + block - 1) / block;
        hipLaunchKernelGGL(k_collision_reward, dim3(grid), dim3(block), 0, (hipStream_t)stream, w->W, w->n, rows,
                           (const float*)w->d_state, as, fs, (const float*)w->d_robot, d_action, T, d_global_time,
                           reward_cfg[0], reward_cfg[1], reward_cfg[2], reward_cfg[3], reward_cfg[4], d_out, (w->flags & CS_ROBOT_UNICYCLE) ? 1 : 0);
    }
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_collision_reward_gym(const cs_worlds* w, const float* d_action, float T, float* d_global_time, const float* reward_cfg,
                            float* d_out, const cs_gym_book* book, void* stream)
{
    if (!w || !book) return fail(CS_ERR_ARG, "null argument");
    if (w->W <= 0 || w->n <= 0 || !w->d_state) return fail(CS_ERR_ARG, "bad cs_worlds");
    if (w->layout != CS_LAYOUT_AOS && w->layout != CS_LAYOUT_SOA) return fail(CS_ERR_ARG, "bad layout");
    if (!d_action || !d_global_time || !reward_cfg || !d_out || !w->d_robot) return fail(CS_ERR_ARG, "null argument");
    if (book->clock_len <= 0 || !book->d_counter || !book->d_seeds || !book->d_mask || !book->d_clock || !book->d_reward ||
        !book->d_terminated || !book->d_truncated || !book->d_info)
        return fail(CS_ERR_ARG, "null buffer in cs_gym_book");
    if (w->n > 64) {   // the lane-per-world reward kernel: the two launches
        const int rc = cs_collision_reward(w, d_action, T, d_global_time, reward_cfg, d_out, stream);
        if (rc) return rc;
        if (book->d_prev_mask)
            return cs_gym_bookkeeping_next_step(w->W, d_out, book->d_counter, book->d_seeds, book->d_mask, book->d_prev_mask, d_global_time,
                                                book->d_clock, book->clock_len, book->d_reward, book->d_terminated, book->d_truncated,
                                                book->d_info, book->seed_stride, stream);
        return cs_gym_bookkeeping(w->W, d_out, book->d_counter, book->d_seeds, book->d_mask, d_global_time, book->d_clock, book->clock_len,
                                  book->auto_reset, book->d_reward, book->d_terminated, book->d_truncated, book->d_info, book->seed_stride, stream);
    }
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    long as, fs;
    strides(w, rows, as, fs);
    const int wpb = 64 / w->n, grid = (w->W + wpb - 1) / wpb;
    hipLaunchKernelGGL(k_collision_reward_wave, dim3(grid), dim3(64), 0, (hipStream_t)stream, w->W, w->n, rows, wpb,
                       (const float*)w->d_state, as, fs, (const float*)w->d_robot, d_action,
                       gym_head(d_out, d_global_time, T, reward_cfg, book, w->W, w->flags));
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

#ifdef CS_STAMPS
int cs_debug_set_stamp_buffer(void* d_buf) { g_stamp_buf = (unsigned long long*)d_buf; return CS_OK; }
#endif

int cs_imitation_block(const cs_worlds* w, int32_t robot_type, const float* robot_params, float robot_margin, const float* d_human_margin,
                       float* d_robot_memory, float dt, int n_substeps, void* stream)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (n_substeps <= 0) return fail(CS_ERR_ARG, "n_substeps must be positive");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    // An invisible robot does not act on the crowd: the crowd's n substeps fuse into ONE launch that leaves what the robot's integrator
    // sees of it at every substep in a snapshot, and the robot's n substeps are ONE launch behind it -- 2 launches instead of 2 n.
    const bool fusable = !(w->flags & CS_ROBOT_ROW) && w->type >= 0 && w->type <= 8 && robot_type >= 0 && robot_type <= 8 &&
                         rows <= 64 && w->d_robot != nullptr;
    if (fusable) {
        float4* snap = nullptr;
        int rc = csimpl::scratch((void**)&snap, csimpl::imitation_scratch_bytes(w, n_substeps), csimpl::SCRATCH_IMITATION, (hipStream_t)stream);
        if (rc) return rc;
        rc = launch_step(w, dt, n_substeps, M_COMMIT_GOALS, nullptr, nullptr, nullptr, (hipStream_t)stream, snap);
        if (rc) return rc;
        return csimpl::robot_block_launch(w, robot_type, robot_params, robot_margin, d_human_margin, d_robot_memory, dt, n_substeps, snap,
                                          (hipStream_t)stream);
    }
    // A VISIBLE robot and the crowd act on each other in every substep: update_robot runs INSIDE the crowd's substep loop, the robot
    // being the last row of every world (k_sfm_step<..., LEAN = 4>, robot_model.h) -- one launch, bit-identical to the alternation below.
    // Needs the plain crowd batch (all_params_equal, <= 2 goal slots, no walls) and SFM / HSFM models on both sides.
    const bool fusable_visible = (w->flags & CS_ROBOT_ROW) && (w->flags & CS_ALL_PARAMS_EQUAL) && w->type >= 0 && w->type <= 8 && robot_type >= 0 &&
                                 robot_type <= 8 && rows <= 64 && w->d_robot != nullptr && w->O == 0 && w->G <= 2 && robot_params && d_robot_memory &&
                                 !(std::getenv("CROWDSTEP_IMITATION_FUSED") && std::getenv("CROWDSTEP_IMITATION_FUSED")[0] == '0');
    if (fusable_visible) {
        const RobotModel rm{robot_type, robot_params, robot_margin, d_human_margin ? d_human_margin : w->d_safety, d_robot_memory};
        return launch_step(w, dt, n_substeps, M_COMMIT_GOALS | M_ROBOT_FROM_ARRAY, nullptr, nullptr, nullptr, (hipStream_t)stream, nullptr, nullptr, &rm);
    }
    for (int k = 0; k < n_substeps; ++k) {   // ORCA on either side, walls with a visible robot: the reference's strict alternation
        int rc = cs_robot_model_step(w, robot_type, robot_params, robot_margin, d_human_margin, d_robot_memory, dt, stream);
        if (rc) return rc;
        rc = cs_step(w, dt, 1, nullptr, stream);
        if (rc) return rc;
    }
    return CS_OK;
}

int cs_reserve_scratch(const cs_worlds* w, int n_substeps, void* stream)
{
    if (!w) return fail(CS_ERR_ARG, "null cs_worlds");
    if (n_substeps <= 0) return fail(CS_ERR_ARG, "n_substeps must be positive");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    void* p = nullptr;
    int rc = CS_OK;
    if (w->type == CS_ORCA) {
        if (csimpl::orca_uses_grid(w)) rc = csimpl::scratch(&p, csimpl::orca_big_scratch_bytes(w), csimpl::SCRATCH_ORCA_BIG, (hipStream_t)stream);
        return rc;
    }
    if (w->type < 0 || w->type > 8) return CS_OK;   // (social momentum keeps no library scratch)
    if (rows > csimpl::big_world_min_rows(1024)) return csimpl::scratch(&p, csimpl::sfm_big_scratch_bytes(w), csimpl::SCRATCH_SFM_BIG, (hipStream_t)stream);
    if (!(w->flags & CS_ROBOT_ROW) && rows <= 64 && w->d_robot != nullptr)   // what cs_imitation_block's fused form records
        rc = csimpl::scratch(&p, csimpl::imitation_scratch_bytes(w, n_substeps), csimpl::SCRATCH_IMITATION, (hipStream_t)stream);
    return rc;
}

int cs_release_scratch(void)
{
    HIP_TRY(hipDeviceSynchronize());
    return csimpl::scratch_release_all();
}

int cs_step_variant(const cs_worlds* w, int entry, char* buf, size_t buflen)
{
    if (!w || !buf || buflen == 0) return fail(CS_ERR_ARG, "null argument");
    if (entry < 0 || entry > 2) return fail(CS_ERR_ARG, "entry must be 0 (cs_step), 1 (cs_update_humans_parallel, out of place) or 2 (cs_peek)");
    if (w->type == CS_ORCA) return csimpl::orca_variant(w, buf, buflen);
    if (w->type == CS_SOCIAL_MOMENTUM) { std::snprintf(buf, buflen, "k_sm_step"); return CS_OK; }
    int rc = check_worlds(w);
    if (rc) return rc;
    Geometry g;
    rc = geometry(w, g);
    if (rc) return rc;
    int mode = entry == 2 ? (int)M_PEEK : (int)M_COMMIT_GOALS;
    if (entry == 1) mode |= M_MUTATE_INPUT;
    if (entry != 1 && (w->flags & CS_ROBOT_ROW) && w->d_robot) mode |= M_ROBOT_FROM_ARRAY;
    if (w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0) > csimpl::big_world_min_rows(1024)) {
        std::snprintf(buf, buflen, "k_bw_sfm_step<SOC=%d,HEADED=%d,PEQ=%d> grid=%d block=256 (uniform grid in HBM)", w->type % 3, w->type / 3,
                      (w->flags & CS_ALL_PARAMS_EQUAL) ? 1 : 0, g.grid);
        return CS_OK;
    }
    const Variant v = select_variant(w, mode, g);
    if (v.maxt == 16) {
        std::snprintf(buf, buflen, "k_sfm_step_row16<SOC=%d,HEADED=%d,ROWS=%d> grid=%d block=64 wpb=4", w->type % 3, w->type / 3, v.rows_ct, (w->W + 3) / 4);
        return CS_OK;
    }
    std::snprintf(buf, buflen, "k_sfm_step<SOC=%d,HEADED=%d,PEQ=%d,MAXT=%d,OCC=%d,ROWS_CT=%d,LEAN=%d> grid=%d block=%d wpb=%d lds=%d%s",
                  w->type % 3, w->type / 3, v.peq ? 1 : 0, v.maxt, v.occ, v.rows_ct, v.lean, g.grid, g.block, g.wpb,
                  (int)step_lds_bytes(w, g, v.peq, nullptr, nullptr),
                  g.block == 64 ? (step_wg_waves_for(step_lds_bytes(w, g, v.peq, nullptr, nullptr)) == 4 ? " wg=4" : (step_wg_waves_for(step_lds_bytes(w, g, v.peq, nullptr, nullptr)) == 2 ? " wg=2" : " wg=1")) : "");
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
