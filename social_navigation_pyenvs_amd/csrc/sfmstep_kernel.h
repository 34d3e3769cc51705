// sfmstep_kernel.h -- the fused SFM / HSFM step kernel template (k_sfm_step) of libcrowdstep.so, shared by the translation units that
// instantiate its builds (sfmstep_*.hip: one group of builds each, compiled in parallel) and by crowdstep.hip, which picks the build.
//
// Hot path: the reference's per-substep pedestrian update
//   update_humans_parallel            /root/reference/social_gym/src/forces_parallel.py:185-284
//   MotionModelManager.update_humans  /root/reference/social_gym/src/motion_model_manager.py:354-422
//   SocialNavGym.step substep loop    /root/reference/social_gym/social_nav_gym.py:240-245
// Design notes: DESIGN.md §4.1.  gfx950 only: no portability macros, no CPU fallback.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>
#include <utility>

#include "common.h"
#include "crowdstep.h"
#include "robot_model.h"
#include "stepcommon.h"

namespace cstep {

// ------------------------------------------------------------------------------------------
// the fused SFM / HSFM step kernel
//   SOC    = type % 3  (0 Helbing, 1 Guo, 2 Moussaid)          forces_parallel.py:215
//   HEADED = type / 3  (0 SFM, 1 HSFM torque on desired force, 2 on total force)   :217
//   PEQ    = all_params_equal
//   MAXT   = 64 (one wavefront, floor(64/rows) worlds) or 1024 (one world per block)
// ------------------------------------------------------------------------------------------
template <class F, int... I>
__device__ __forceinline__ void for_each_index(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}

// The pair-once loop hands values from lane to lane through LDS inside ONE wavefront.  The hardware executes a wave's LDS
// operations in order, so no s_barrier / s_waitcnt is needed -- but the compiler does not know that another lane's store
// feeds this lane's load, and with compile-time offsets it can prove "no alias" for the lane's own addresses and hoist the
// load above the store.  This compiler-only fence pins the program order of the LDS accesses around it.
#define LDS_ORDER_FENCE() asm volatile("" ::: "memory")

// reaction accumulator rows of the pair-once loop (independent LDS read-modify-write chains) = partners per group.  Measured at
// 4096 x 25 (same-box A/B): 6 rows 38.5 us, 4 rows 35.9, 3 rows 34.2, 2 rows 32.5, 1 row 32.8 -- every row costs a 1 KB zeroing store
// and two 512 B reads in the reaction sum per wavefront-substep, and the LDS pipe of a CU (128 B/clk, 8 wavefronts) is ~40 % busy
constexpr int UA = 2;
constexpr int ACC_PITCH = 128; // float2 slots per accumulator row (block = one wavefront: 2 x 64 doubled rows)

//   OCC    = waves per SIMD the register allocation must allow: 4 (<= 128 VGPRs, a few spills) when the grid holds more
//            than two wavefronts per SIMD, 1 (unconstrained, ~150 VGPRs, no spills) otherwise -- at 4096 x 25 there
//            are exactly two waves per SIMD and the spill-free build is 6 % faster; at 16384 x 25 the 4-wave build is 7 % faster
//   ROWS_CT = rows per world known at compile time (0: read from the arguments): the partner-group loop unrolls and
//            its ~9 scalar branches per substep (~24 cycles of wave latency each) disappear
//   LEAN   = 1 / 2 / 3: pair-once build for the plain crowd batch: no walls (1), walls kept (2), or no walls + a robot as the last
//            row (3: what a Gym with a VISIBLE robot steps; rows = humans + 1), goal lists of <= 2 entries, state committed in
//            place -- the wall / robot / goal-list-in-memory code and their branches are compiled out wherever the build
//            does not need them and the goal switch is predicated; 4 = 3 + the robot follows a HUMAN motion model of its own
//            (imitation learning, social_nav_gym.py:252-274): update_robot runs inside the substep loop (robot_model.h);
//            5 = 3 with the walls kept (a Gym with a visible robot in a walled scene)
template <int SOC, int HEADED, bool PEQ, int MAXT, int OCC, int ROWS_CT, int LEAN>
__global__ __launch_bounds__(MAXT, OCC) void k_sfm_step(const KArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // all_params_equal, whole worlds inside one wavefront: every unordered pair is evaluated ONCE (as the reference
    // does, forces_parallel.py:100-131: F[i,j] = f, F[j,i] = -f) and the reaction handed over through LDS.
    // Per-agent parameters (forces_parallel.py:43-84, :261: F_i = sum_j f(P_i; i, j), no antisymmetry) take the same loop for the Helbing /
    // Guo laws (PP): the lane that visits the pair (i, j) evaluates BOTH directions -- d, 1/d and the overlap are shared, each side's
    // A e^{./B} (+ C e^{./D}) comes from its own parameters, the partner's read from an LDS row beside its position -- 12 VALU + 3
    // transcendentals per unordered pair instead of 2 x (12 + 2) (the peragent bench entry: 443 -> 35x VALU per wavefront-substep).
    constexpr bool N3L = MAXT == 64 && (PEQ || SOC != 2);
    constexpr bool PP = N3L && !PEQ;
    const int T = blockDim.x;
    // Every world's rows are stored TWICE, back to back ([w][2][rows]): lane i then reads its partners
    // i+1 .. i+rows-1 at constant offsets from one base address -- no own-row slot, no modulo, no
    // per-partner compare (v_cmp + v_cndmask costs as much as a transcendental on this SIMD).
    const int TP = 2 * T + PADR;                                   // rows per position buffer (+ finite padding)
    float4* lds_p = reinterpret_cast<float4*>(smem_raw);           // [2][TP] x, y, radius+safety, -
    float2* lds_v = reinterpret_cast<float2*>(lds_p + 2 * TP);     // [2][TP] stored linear velocity (doubled rows like lds_p)
    float2* lds_vr = lds_v + 2 * TP;                               // [2][T] velocity as refreshed in-place
    float* lds_g0x = reinterpret_cast<float*>(lds_vr + 2 * T);     // [T] respawn scratch
    int* lds_flag = reinterpret_cast<int*>(lds_g0x + T);           // [T] respawn scratch
    float2* lds_acc = reinterpret_cast<float2*>(lds_flag + T);     // [UA][2T] reaction accumulators (N3L only)
    float4* lds_pp = reinterpret_cast<float4*>(lds_acc + (N3L ? UA * ACC_PITCH : 0));   // [2T + PADR] (PP only) partner parameters, doubled rows like lds_p:
                                                                                        // lA + cB rs, cB, lC + cD rs, cD  (its own radius + safety space folded in)
    float4* lds_seg = lds_pp + (PP ? 2 * T + PADR : 0);            // [seg_tab] x1, y1, ex, ey
    float* lds_sinv = reinterpret_cast<float*>(lds_seg + a.seg_tab);                   // [seg_tab] 1 / |e|^2, 0 = NaN slot
    float4* lds_poly = reinterpret_cast<float4*>(lds_sinv + ((a.seg_tab + 3) & ~3));   // [seg_tab / Smax] bounding circle cx, cy, R of every polygon
    // wall pairs (LEAN = 2; all_params_equal, no respawn rule): up to WP_MAX (agent, polygon) pairs per block.  The 50-row wall build runs
    // twelve blocks per CU at 10.5 KB each: 4 KB more per block cost it one of them (measured: 208 -> 260 us), so the agents' radii live in
    // lds_vr (written by the per-agent-parameter builds only) and the contact velocities in the respawn scratch (idle without the rule)
    constexpr int WP_MAX = 128;
    int* lds_wrec = reinterpret_cast<int*>(lds_poly + (a.Smax > 0 ? a.seg_tab / a.Smax : 0));   // [WP_MAX] pair: agent's row in lds_p | first segment << 8 | agent lane << 20
    float2* lds_wres = reinterpret_cast<float2*>(lds_wrec + WP_MAX);                             // [WP_MAX + 1] the pairs' forces ([WP_MAX]: the zero of "no pair")
    float4* lds_wlaw = reinterpret_cast<float4*>(lds_wres + WP_MAX + 2);                         // [1] the wall law A, log2 e / B, k1, k2 (all_params_equal)
    float2* lds_wrs = lds_vr;                                                                    // [WP_MAX] pair: the agent's radius, safety space
    float2* lds_wcv = reinterpret_cast<float2*>(lds_g0x);                                        // [T] the agents' refreshed linear velocities (contact terms)
    const int tid = threadIdx.x;
    static_assert(!LEAN || (PEQ && MAXT == 64), "the lean build is a pair-once build");
    constexpr bool LEAN_ROBOT = LEAN == 3 || LEAN == 4 || LEAN == 5;   // the robot is the last row (5: with walls)
    constexpr bool IMIT = LEAN == 4;                       // ... and follows its own human motion model
    constexpr bool NO_WALLS = LEAN == 1 || LEAN == 3 || LEAN == 4;     // wall code compiled out
    const int rows = ROWS_CT > 0 ? ROWS_CT : a.rows;
    const int n = LEAN_ROBOT ? rows - 1 : (LEAN ? rows : a.n);
    const int kmode = LEAN_ROBOT ? ((int)M_COMMIT_GOALS | (a.mode & (int)M_ROBOT_FROM_ARRAY)) : (LEAN ? (int)M_COMMIT_GOALS : a.mode);
    const int lw = tid / rows;
    const int row = tid - lw * rows;
    const int w = blockIdx.x * a.wpb + lw;
    const bool valid = (lw < a.wpb) && (w < a.W);
    const bool robot_row = LEAN_ROBOT ? true : (LEAN ? false : (a.flags & CS_ROBOT_ROW) != 0);
    const bool human = valid && row < n;
    const bool is_robot = valid && robot_row && row == n;
    const int base = lw * rows;        // first row of my world in the per-row arrays (lds_v, lds_vr, ...)
    const int pbase = lw * a.ws;       // first row of my world in the doubled position buffers
    const float dt = a.dt;
    const int obs_type = (a.type == 1 || a.type == 4 || a.type == 7) ? 1 : 0;

    // ---- load phase ---------------------------------------------------------------------
    // Every global load of the prologue is issued before the first result is used: ONE memory round trip.  (Until round 4 the row,
    // the parameter rows, the goal list -- a NaN scan that loaded and waited slot by slot --, the world's respawn flag and the action
    // were seven dependent round trips, ~3.2 us of a 33 us launch by the s_memtime stamps of tools/launch_floor.sh's probe.)
    // Straight-line code on purpose: a load inside a branch whose other arm is a constant gets its first use hoisted into the branch
    // (and a wait with it), so lanes without a world / a human row load world 0's / row 0's (never used) and an absent optional array is
    // replaced by a readable dummy address instead of being branched around.
    const long sidx = (long)w * rows + row;
    const int wc = valid ? w : 0, rowh = human ? row : 0;
    const long sidx_c = valid ? sidx : 0;
    const float* dummy = a.Sin;
    float px, py, th, vx, vy, bvx, bvy, om, r, m, gx, gy, vd;
    {
        const float* s = a.Sin + sidx_c * a.in_as;
        const long fs = a.in_fs;
        px = s[0]; py = s[fs]; th = s[2 * fs]; vx = s[3 * fs]; vy = s[4 * fs]; bvx = s[5 * fs];
        bvy = s[6 * fs]; om = s[7 * fs]; r = s[8 * fs]; m = s[9 * fs]; gx = s[10 * fs]; gy = s[11 * fs];
        vd = s[12 * fs];
    }
    float safety = a.safety[sidx_c];
    const bool has_wflags = a.world_flags != nullptr;
    const int wflag_raw = *(has_wflags ? a.world_flags + wc : reinterpret_cast<const int*>(dummy));   // bit 0: the respawn rule applies to my world
    const bool robot_moves = a.action != nullptr; // (lean build: only the invisible robot of the epilogue)
    float ax, ay;                                 // robot action (held for the whole block, social_nav_gym.py:240-243)
    {
        const float* ap = robot_moves ? a.action + (long)wc * 2 : dummy;
        ax = ap[0]; ay = ap[1];
    }
    // parameters: my own row of P for the single-agent forces; P[0] of my world for the pair loop
    // when all_params_equal (forces_parallel.py:220), else my own row (:261)
    const long pw = (a.flags & CS_PARAMS_SHARED) ? 0 : (long)wc * n * 20;
    const float* Prow = a.params + pw + (long)rowh * 20;
    float Pm[11];                                 // my P[0], [2], [4], [6], [8], [10], [11], [16], [17], [18], [19]
    Pm[0] = Prow[0]; Pm[1] = Prow[2]; Pm[2] = Prow[4]; Pm[3] = Prow[6]; Pm[4] = Prow[8]; Pm[5] = Prow[10]; Pm[6] = Prow[11];
    Pm[7] = Prow[16]; Pm[8] = Prow[17]; Pm[9] = Prow[18]; Pm[10] = Prow[19];
    const SocRaw sraw = load_socraw(PEQ ? a.params + pw : Prow);
    float* gi = a.goals + ((long)wc * n + rowh) * a.G * 2;   // my goal list (row 0's for a lane without a human)
    float gl[4];                                  // its first two slots
    gl[0] = gi[0]; gl[1] = gi[1];
    {
        const float* g2 = a.G >= 2 ? gi + 2 : gi;
        gl[2] = g2[0]; gl[3] = g2[1];
    }
    if (valid && is_robot && (kmode & M_ROBOT_FROM_ARRAY)) {   // (divergent: the robot's lane)
        const float* rb = a.robot + (long)w * 13;
        px = rb[0]; py = rb[1]; th = rb[2]; vx = rb[3]; vy = rb[4]; bvx = rb[5]; bvy = rb[6]; om = rb[7];
        r = rb[8]; m = rb[9]; gx = rb[10]; gy = rb[11]; vd = rb[12];
    }
    // (cs_gym_step: the robot as the head sees it and the action, see below)
    float hrb[5] = {0, 0, 0, 0, 0}, hact[2] = {0, 0};
    // cs_gym_step_staged's take-over in the epilogue (gymhead.h GymFold): compiled into the builds without walls only -- the 50-row wall build
    // has no register to spare (its shard went 191 -> 200 us with the fold compiled in), and a Gym with polygon walls keeps the two launches
    constexpr bool FOLD = MAXT == 64 && !(LEAN == 2 || LEAN == 5);
    GymPre hpre = {0.0f, 0, 0, 0, 0u, 0u};
    if constexpr (MAXT == 64) {
        if (a.gym.out != nullptr && valid) {
            const float* rb = a.robot + (long)w * 13;      // the robot BEFORE its move of substep 1
            hrb[0] = rb[0]; hrb[1] = rb[1]; hrb[2] = rb[8]; hrb[3] = rb[10]; hrb[4] = rb[11];
            hact[0] = a.action[(long)w * 2]; hact[1] = a.action[(long)w * 2 + 1];
            if (row == 0) hpre = gym_head_preload<MAXT == 64 && !(LEAN == 2 || LEAN == 5)>(a.gym, w);   // (last: the compiler waits for these right here)
        }
    }
    // imitation learning (LEAN = 4)
    float rm_hm = 0.0f, rm_fdx = 0.0f, rm_fdy = 0.0f;
    if constexpr (IMIT) {
        if (human) rm_hm = a.rm_hmargin[sidx];
        if (is_robot) { rm_fdx = a.rm_memory[(long)w * 2]; rm_fdy = a.rm_memory[(long)w * 2 + 1]; }
    }

    const float inv_O = a.O > 0 ? 1.0f / (float)a.O : 0.0f;
    const float* obst = nullptr;
    if (!NO_WALLS && a.O > 0) obst = a.obstacles + ((a.flags & CS_OBSTACLES_SHARED) ? 0 : (long)w * a.O * a.Smax * 4);
    // (their loads belong to the load phase: the LDS stores below are the first use of anything loaded so far)
    // wall segments are constant over the launch: stage (x1, y1, e, 1/|e|^2) in LDS once instead of re-loading and
    // re-deriving them in every substep (3 polygons x 5 segments cost as much as the whole 50-agent pair loop otherwise)
    const int nseg = NO_WALLS ? 0 : a.O * a.Smax;
    const int sbase = (a.flags & CS_OBSTACLES_SHARED) ? 0 : lw * nseg;
    for (int i = tid; i < (NO_WALLS ? 0 : a.seg_tab); i += T) {
        const int lwi = i / (nseg > 0 ? nseg : 1);
        const long wi = (long)blockIdx.x * a.wpb + lwi;
        // a NaN slot becomes a degenerate segment 1e18 m away: squared distance 2e36, never the polygon's minimum (the reference
        // stores the largest int64 as that slot's distance, forces_parallel.py:247) -- no per-slot test in the substep loop
        float4 e = make_float4(1.0e18f, 1.0e18f, 0.0f, 0.0f);
        float inv = 0.0f;
        if ((a.flags & CS_OBSTACLES_SHARED) || wi < a.W) {
            const float* src = a.obstacles + ((a.flags & CS_OBSTACLES_SHARED) ? (long)i * 4 : (wi * nseg + (i - lwi * nseg)) * 4);
            const float4 seg = *reinterpret_cast<const float4*>(src);
            if (!isnan(seg.x)) {
                e = make_float4(seg.x, seg.y, seg.z - seg.x, seg.w - seg.y);
                inv = rcp_fast(fmaf(e.z, e.z, e.w * e.w));
            }
        }
        lds_seg[i] = e;
        lds_sinv[i] = inv;
    }

    // ---- compute phase --------------------------------------------------------------------
    if (!valid) {   // (selects: what the rows of a lane without a world have always been)
        px = 0; py = 0; th = 0; vx = 0; vy = 0; bvx = 0; bvy = 0; om = 0; r = 0; m = 1; gx = 0; gy = 0; vd = 0;
        safety = 0; ax = 0; ay = 0;
    }
    if (!robot_moves) { ax = 0; ay = 0; }
    float m_tau = 0, Aw = 0, cBw = 0, Cw = 0, cDw = 0, k1 = 0, k2 = 0, ko = 0, kd = 0, alpha = 1, klam = 0;
    float dt_m = 0, inv_alpha = 1, inertia = 1, dt_inertia = 0, wall_cut = 0;
    SocP sp = {};
    float g0x = gx, g0y = gy, g1x = 0, g1y = 0;
    int gk = 0;          // length of the non-NaN prefix of my goal list
    bool gdirty = false; // a two-goal list rotated in registers, to be written back in the epilogue
    if (N3L && PEQ && valid) sp = make_socp(sraw);
    if (human) {
        m_tau = m / Pm[0];                      // m / relax_t              (:39)
        Aw = Pm[1]; cBw = LOG2E / Pm[2]; Cw = Pm[3]; cDw = LOG2E / Pm[4]; k1 = Pm[5]; k2 = Pm[6];
        // beyond this distance a wall's force on me is below |A| e^-wall_efolds (22: 5.6e-7 N, crowdstep.hip; its contact terms are exact zeros there):
        // a polygon that every agent of the wavefront is that far from is skipped in the substep loop
        wall_cut = r + safety + a.wall_efolds * fmaxf(Pm[2], obs_type == 1 ? Pm[4] : 0.0f);
        ko = Pm[7]; kd = Pm[8]; alpha = Pm[9]; klam = Pm[10];
        dt_m = a.dt / m;                        // (F / m) * dt             (:277,:282)
        inv_alpha = 1.0f / alpha;
        inertia = 0.5f * m * r * r;             // :265
        dt_inertia = a.dt / inertia;            // (torque / I) * dt        (:279)
        if constexpr (!N3L || PP) sp = make_socp(sraw);
        g0x = gl[0]; g0y = gl[1];
        // goal lists of <= 2 entries (every Gym scenario) rotate in registers; longer ones go through memory
        gk = a.G;
        for (int g = a.G - 1; g >= 2; --g)
            if (isnan(gi[2 * g]) || isnan(gi[2 * g + 1])) gk = g;
        if (a.G >= 2 && (isnan(gl[2]) || isnan(gl[3]))) gk = 1;
        if (isnan(gl[0]) || isnan(gl[1])) gk = 0;
        if (gk == 2) { g1x = gl[2]; g1y = gl[3]; }
    }
    const bool respawn_here = valid && (!has_wflags || (wflag_raw & 1));

    auto robot_step = [&]() { // robot_agent.py:114-136
        if (a.flags & CS_ROBOT_UNICYCLE) {
            float c, s;
            sincos_fast(th + ay, s, c);
            px += c * ax * dt; py += s * ax * dt;
            th = fmodf(th + ay, 6.283185307179586f);
            if (th < 0) th += 6.283185307179586f;
            sincos_fast(th, s, c);
            vx = c * ax; vy = s * ax;
        } else {
            px += ax * dt; py += ay * dt; vx = ax; vy = ay;
        }
    };

    float cs = 1.0f, sn = 0.0f; // cos / sin of my theta, carried from one substep to the next

    // imitation learning (LEAN = 4): the robot lane integrates the robot's own motion model; the humans' lanes each hand it their
    // term of its social force.  LDS (aliases of regions this build does not use): per world [state (x, y, vx, vy)] [radius + margin],
    // and one float2 slot per lane for the terms.
    float4* lds_rob = reinterpret_cast<float4*>(lds_vr);
    float2* lds_rf = reinterpret_cast<float2*>(lds_g0x);

    // ---- cs_gym_step: the head of the Gym step, on the rows as they came in (social_nav_gym.py:229-233) -- the swept robot-human
    //      distances by the humans' lanes, then the lane of row 0 walks its world's in index order and does the episode bookkeeping
    //      (gymhead.h: the very code of k_collision_reward_wave).  One scalar branch in a plain cs_step.
    if constexpr (MAXT == 64) {
        if (a.gym.out != nullptr) {
            float* clo = lds_g0x;                              // [T] (the block-respawn scratch: unused by a block of one wavefront)
            const float rpx = hrb[0], rpy = hrb[1], rr = hrb[2], rgx = hrb[3], rgy = hrb[4], gax = hact[0], gay = hact[1];   // (loaded at the top)
            clo[tid] = human ? gym_swept_closest(px, py, vx, vy, r, rpx, rpy, rr, gax, gay, a.gym.T) : INFINITY;
            LDS_ORDER_FENCE();
            if (valid && row == 0) {
                const bool take = gym_head_world(a.gym, w, n, clo + base, rpx, rpy, rr, rgx, rgy, gax, gay, hpre);
                if constexpr (FOLD) { if (a.gym.fold.on) gym_fold_decide(a.gym, w, take, take && !((a.gym.bk.mode == 2 && hpre.prev) || hpre.pending), hpre); }
            }
            LDS_ORDER_FENCE();
        }
    }

    // ---- prologue: publish substep-0 rows ----------------------------------------------
    if (is_robot && robot_moves) robot_step();
    const float my_rs = r + safety;
    // Helbing / Guo exponents with MY radius + safety space folded into the offset: (r_ij - d) cB + lA = (rs_j - d) cB + (lA + rs_i cB),
    // one add less per pair; the contact test rd > 0 becomes max(rs_j - d) > -rs_i
    float lAi = 0.0f, lCi = 0.0f;
    for (int i = tid; i < 2 * TP; i += T) { // padding stays finite
        lds_p[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        lds_v[i] = make_float2(0.0f, 0.0f);
    }
    if constexpr (PP)
        for (int i = tid; i < TP; i += T) lds_pp[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    // bounding circle of every staged polygon (centre = mean of its segments' midpoints, radius = farthest endpoint)
    for (int q = tid; q < (NO_WALLS || a.Smax <= 0 ? 0 : a.seg_tab / a.Smax); q += T) {
        float cx = 0.0f, cy = 0.0f, cnt = 0.0f;
        for (int sg = 0; sg < a.Smax; ++sg) {
            const float4 e = lds_seg[q * a.Smax + sg];
            if (lds_sinv[q * a.Smax + sg] > 0.0f) { cx += e.x + 0.5f * e.z; cy += e.y + 0.5f * e.w; cnt += 1.0f; }
        }
        const float ic = cnt > 0.0f ? 1.0f / cnt : 0.0f;
        cx *= ic; cy *= ic;
        float r2 = 0.0f;
        for (int sg = 0; sg < a.Smax; ++sg) {
            const float4 e = lds_seg[q * a.Smax + sg];
            if (lds_sinv[q * a.Smax + sg] > 0.0f) {
                const float ax0 = e.x - cx, ay0 = e.y - cy, bx0 = ax0 + e.z, by0 = ay0 + e.w;
                r2 = fmaxf(r2, fmaxf(fmaf(ax0, ax0, ay0 * ay0), fmaf(bx0, bx0, by0 * by0)));
            }
        }
        lds_poly[q] = make_float4(cx, cy, cnt > 0.0f ? sqrtf(r2) : -1.0e30f, 0.0f);   // an empty polygon is never near
    }
    // radius + safety space never changes during a launch: it is stored once in both buffers, a substep only rewrites
    // (x, y) -- two 8-byte LDS stores instead of two 16-byte ones
    auto publish = [&](int buf) {      // my position, both copies
        const float2 me = make_float2(px, py);
        *reinterpret_cast<float2*>(&lds_p[buf * TP + pbase + row]) = me;
        *reinterpret_cast<float2*>(&lds_p[buf * TP + pbase + rows + row]) = me;
    };
    // Helbing / Guo pair-once builds read partner velocities only in the (rare) contact pass, which publishes them itself
    constexpr bool VEL_ON_DEMAND = N3L && SOC != 2 && PEQ;
    auto publish_v = [&](int buf) {    // stored linear velocity; second copy only where the rotated loop reads it
        if constexpr (!VEL_ON_DEMAND) {
            lds_v[buf * TP + pbase + row] = make_float2(vx, vy);
            if constexpr (N3L && SOC == 2) lds_v[buf * TP + pbase + rows + row] = make_float2(vx, vy);
        }
    };
    if (valid) {
        // (PP: the sign of my C / A goes along in the free fourth slot: the partner that evaluates my side needs it)
        const float4 me = make_float4(px, py, my_rs, PP ? sp.sAC : 0.0f);
        for (int buf = 0; buf < 2; ++buf) {
            lds_p[buf * TP + pbase + row] = me;
            lds_p[buf * TP + pbase + rows + row] = me;
        }
        if constexpr (PP) {
            // my side of a pair as my partner will evaluate it: exponent (rs_partner - dist) cB + (lA + cB rs_me); a row without
            // parameters (the robot: a source only, its own sum is never used) publishes finite zeros
            const float4 mine = make_float4(fmaf(sp.cB, my_rs, sp.lA), sp.cB, fmaf(sp.cD, my_rs, sp.lC), sp.cD);
            lds_pp[pbase + row] = mine;
            lds_pp[pbase + rows + row] = mine;
        }
        publish_v(0);
        float rvx = vx, rvy = vy;
        if (HEADED > 0 && human) {
            sincos_fast(th, sn, cs);
            rvx = cs * bvx + (-sn) * bvy;
            rvy = sn * bvx + cs * bvy;
        }
        lds_vr[tid] = make_float2(rvx, rvy);
    }
    __syncthreads();

#ifdef CS_STAMPS
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    int cur = 0;
    lAi = fmaf(sp.cB, my_rs, sp.lA); lCi = fmaf(sp.cD, my_rs, sp.lC);
    // compile-time row counts: radius + safety space of my partners (ring distance k + 1) never changes during a launch -- held in
    // registers, so that a substep reads 8-byte (x, y) partner rows from the LDS instead of 12-byte ones (the LDS pipe is a co-bottleneck)
    constexpr int RSN = (ROWS_CT > 0 && N3L) ? (ROWS_CT - 1) / 2 + 1 : 1;
    float rsj[RSN];
#pragma unroll
    for (int k = 0; k < RSN; ++k) rsj[k] = 0.0f;
    // ... and the reaction slot of partner k: (row + 1 + k) mod rows.  With the modulo every slot of an accumulator row is written by
    // exactly one lane in every group, so the FIRST group stores instead of accumulating (no zeroing pass, no read), and a receiver reads
    // one slot per row instead of a low and a high copy -- a quarter of the loop's LDS bytes at 25 rows
    int aoff[RSN];
#pragma unroll
    for (int k = 0; k < RSN; ++k) aoff[k] = 0;
    // ... and, with per-agent parameters (PP), its parameter row and the sign of its C / A: nothing but (x, y) is read per pair
    constexpr int PPN = (ROWS_CT > 0 && PP) ? RSN : 1;
    float4 ppj[PPN];
    float sacj[PPN];
#pragma unroll
    for (int k = 0; k < PPN; ++k) { ppj[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); sacj[k] = 0.0f; }
    if constexpr (ROWS_CT > 0 && N3L) {
        if (valid) {
#pragma unroll
            for (int k = 0; k < RSN; ++k) {
                const float4 pk = lds_p[pbase + row + 1 + k];
                rsj[k] = pk.z;
                if constexpr (PP) { sacj[k] = pk.w; ppj[k] = lds_pp[pbase + row + 1 + k]; }
                const int t = row + 1 + k;
                aoff[k] = t >= ROWS_CT ? t - ROWS_CT : t;
            }
        }
    }
    // Two wavefronts share a SIMD at the benchmark's 4096 worlds, and of two ready wavefronts of equal priority the arbiter issues the OLDER:
    // the first half of the grid finished its 20 substeps 16 % ahead of the second (s_memtime stamps: two classes of exactly 1024
    // wavefronts), which then ran its last three substeps alone, at a lone wavefront's latency-bound pace.  The younger wavefront of a
    // SIMD (a.young_from, set by the launcher for a grid of exactly two per SIMD) raises its priority in every other substep: the two
    // take turns (priority 2 / 1) and finish within 3 % of each other (31.1 -> 29.4 us with the short rare paths, -> 28.3 us with this).
    // ---- wall pairs (LEAN = 2: the 50-row crowd with polygon walls, cfg5's per-GPU shard) --------------------------------------------
    // Every lane used to walk every polygon some agent of its wavefront is near: with 50 agents somebody is near each of the three
    // polygons, so the wave vote never skips -- 3 x (vote + 5 segments x 13 + force) = 252 of a substep's 710 vector instructions --
    // while half to two thirds of the 150 (agent, polygon) pairs are within the reach of the force in that window, none before it
    // (tools/wall_pairs_stats.py).
    // Round 4 compacted the pairs inside every substep (votes, prefix counts, rows handed over through LDS): 90 instructions of
    // bookkeeping per substep and no gain.  Here the pairs are numbered ONCE per launch: an agent that starts farther than
    // reach + n_substeps dt v_desired from a polygon cannot get within its reach during the launch (speeds are clamped to v_desired and
    // the respawn rule -- the only jump -- keeps the builds with CS_RESPAWN on the old pass), so the numbering holds for all substeps.
    // A substep then costs ONE closest-point + force per lane and pass (two passes for 65 .. 128 pairs; more: the old pass for this
    // launch) and three LDS reads per agent; a lane whose agent touches its polygon adds the k1 / k2 terms from the agent's refreshed
    // velocity (published by every agent at the head of the substep).  Same operations on the same operands per pair; a polygon beyond
    // an agent's reach adds an exact zero instead of a force below the reach's bound.  8192 x 50 + 3 polygons, Gym steps 20-70:
    // 211 -> 197 us at the old reach of 36 e-folding lengths, 205 -> 191 us at 22 (profiles/r5e_wall_pairs_ab.txt).
    constexpr int WP_O = 4;                    // polygons per world the pair numbers of an agent cover (one byte each)
    bool wp_on = false;
    int wp_count = 0;
    unsigned wp_mine = 0x80808080u;            // my pairs' numbers, one byte per polygon; WP_MAX = none (lds_wres[WP_MAX] stays zero)
    if constexpr (LEAN == 2 && MAXT == 64) {
        if (a.wall_pairs != 0) {                                              // (kernel argument: wave-uniform)
            const float travel = (float)a.nsub * a.dt * vd + 1.0e-3f;
            int total = 0;
            for (int o = 0; o < a.O; ++o) {
                const float4 pc = lds_poly[(sbase / a.Smax) + o];
                const float dxc = px - pc.x, dyc = py - pc.y, lim = pc.z + wall_cut + travel;
                const bool reach = human && fmaf(dxc, dxc, dyc * dyc) < lim * lim;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(reach);
                const int rank = total + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                if (reach && rank < WP_MAX) {
                    lds_wrec[rank] = (pbase + row) | ((sbase + o * a.Smax) << 8) | (tid << 20);
                    lds_wrs[rank] = make_float2(r, safety);
                    wp_mine = (wp_mine & ~(0xFFu << (8 * o))) | ((unsigned)rank << (8 * o));
                }
                total += __builtin_popcountll(mk);
            }
            if (tid == 0) { lds_wres[WP_MAX] = make_float2(0.0f, 0.0f); *lds_wlaw = make_float4(Aw, cBw, k1, k2); }   // (lane 0 is a human of the block's first world)
            wp_count = total;
            wp_on = total <= WP_MAX;             // (more: this launch walks the polygons as before)
            LDS_ORDER_FENCE();
        }
    }
    const bool prio_young = MAXT == 64 && (int)blockIdx.x >= a.young_from;
    // base priority 1: above the generator's wavefronts of a refill pass on the side stream (priority 0, and OLDER than any of mine, so a tie
    // would go to them): the Gym step in NEXT_STEP mode 43.3 -> 41.5 us, a plain launch unchanged
    if constexpr (MAXT == 64) __builtin_amdgcn_s_setprio(1);
    for (int sub = 0; sub < a.nsub; ++sub) {
        if constexpr (MAXT == 64) { if (prio_young) { if (sub & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); } }
        const int nxt = cur ^ 1;
        STAMP(7);
        if (a.snap != nullptr || a.trace != nullptr) {   // (one scalar branch for both recorders: neither is on in a plain cs_step)
            // imitation block: what the robot's own integrator sees of the crowd at this substep (it runs before update_humans)
            if (a.snap != nullptr && human) a.snap[((long)sub * a.W + w) * n + row] = make_float4(px, py, vx, vy);
            // cs_step_trace: my row as the previous substep left it (the last one is written behind the loop)
            if (a.trace != nullptr && (human || is_robot) && sub > 0)
                write_trace(a.trace + (((long)(sub - 1) * a.W + w) * rows + row) * 12, px, py, th, vx, vy, bvx, bvy, om, gx, gy, g0x, g0y);
        }
        const int Hf = (rows - 1) >> 1;
        const float4* rp = lds_p + cur * TP + pbase + row + 1;   // rp[k]: partner at ring distance k + 1
        const float2* rv = lds_v + cur * TP + pbase + row + 1;
        const float4* rpp = lds_pp + pbase + row + 1;            // (PP) ... and its parameter row
        float4 qa[UA], pa[PP ? UA : 1];
        float2 va[UA];
        auto fetch_pp = [&](float4 (&pq)[PP ? UA : 1], int kk) {
            if constexpr (PP) {
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    if constexpr (ROWS_CT > 0) pq[u] = ppj[kk + u];   // (kk is a compile-time constant at every call of these builds)
                    else pq[u] = rpp[kk + u];
                }
            }
        };
        auto fetch = [&](float4 (&q)[UA], float2 (&vq)[UA], int kk) {
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                if constexpr (ROWS_CT > 0 && N3L) {   // (kk is a compile-time constant at every call of these builds)
                    const float2 xy = *reinterpret_cast<const float2*>(&rp[kk + u]);
                    q[u] = make_float4(xy.x, xy.y, rsj[kk + u], PP ? sacj[kk + u] : 0.0f);
                } else {
                    q[u] = rp[kk + u];
                }
                if constexpr (N3L && SOC == 2) vq[u] = rv[kk + u]; else vq[u] = make_float2(0.0f, 0.0f);
            }
        };
        if constexpr (IMIT) {
            // -- update_robot(t, dt) (motion_model_manager.py:615-629) BEFORE update_humans: the robot's model sees the humans as they
            //    stand, the crowd then sees the moved robot (its row is re-published into THIS substep's buffer).  Same operations,
            //    same order as k_robot_model_step (robot_model.h): the fused launch equals the alternating launches bit for bit.
            const int rsoc = a.rm_type % 3;
            const bool rheaded = a.rm_type >= 3;
            float rsn = 0.0f, rcs = 1.0f;
            rmodel::RState rs;
            if (is_robot) {
                rs.px = px; rs.py = py; rs.yaw = th; rs.vx = vx; rs.vy = vy; rs.bvx = bvx; rs.bvy = bvy; rs.om = om;
                rs.radius = r; rs.mass = m; rs.gx = gx; rs.gy = gy; rs.vd = vd; rs.fdx = rm_fdx; rs.fdy = rm_fdy;
                rmodel::refresh_velocity(rs, rheaded, rsn, rcs);
                lds_rob[2 * lw] = make_float4(rs.px, rs.py, rs.vx, rs.vy);
                lds_rob[2 * lw + 1] = make_float4(r + a.rm_margin, 0.0f, 0.0f, 0.0f);   // (the region is the prologue's scratch: rewritten here)
            }
            LDS_ORDER_FENCE();
            if (human) {
                const float4 rq = lds_rob[2 * lw];
                const float rme = lds_rob[2 * lw + 1].x;
                float tx, ty;
                rmodel::pair_term(rsoc, a.rm_P, rq.x, rq.y, rq.z, rq.w, px, py, vx, vy, rme + r + rm_hm, tx, ty);
                lds_rf[tid] = make_float2(tx, ty);
            }
            LDS_ORDER_FENCE();
            if (is_robot) {
                float sx = 0.0f, sy = 0.0f;
                const float2* tf = lds_rf + base;
#pragma unroll 8
                for (int j = 0; j < n; ++j) { const float2 t = tf[j]; sx += t.x; sy += t.y; }   // (eight terms requested per trip; summed in index order)
                rmodel::integrate(rs, a.rm_type, a.rm_P, sx, sy, 0.0f, 0.0f, rsn, rcs, dt, 0);
                px = rs.px; py = rs.py; th = rs.yaw; vx = rs.vx; vy = rs.vy; bvx = rs.bvx; bvy = rs.bvy; om = rs.om;
                rm_fdx = rs.fdx; rm_fdy = rs.fdy;
                publish(cur);
                publish_v(cur);
            }
            LDS_ORDER_FENCE();
        }
        // lean build without walls: request the first group's partner rows first thing; the goal test and part A below run
        // while they are in flight.  With walls the rows would be held in registers across the segment loops (17 spilled VGPRs
        // in the 50-row build at the three-wave budget), so they are fetched at the head of the group loop instead (fetching
        // between the segment loops and the heading / torque arithmetic spills as well); worth 1 % on the 8192 x 50 shard
        if constexpr (LEAN && NO_WALLS) { if (valid && Hf >= UA) fetch(qa, va, 0); }
        // even row counts: the antipodal partner (evaluated by both ends, no hand-over) is requested up front as well
        float4 qz = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float2 vz = make_float2(0.0f, 0.0f);
        if constexpr (LEAN && NO_WALLS) {
            if (valid && (rows & 1) == 0) {
                if constexpr (ROWS_CT > 0 && N3L) { const float2 xy = *reinterpret_cast<const float2*>(&rp[(ROWS_CT - 1) / 2]); qz = make_float4(xy.x, xy.y, rsj[(ROWS_CT - 1) / 2], 0.0f); }
                else qz = rp[Hf];
                if constexpr (SOC == 2) vz = rv[Hf];
            }
        }
        if constexpr (LEAN) {
            // -- goal switch, forces_parallel.py:226-234, predicated: lists of <= 2 goals rotate in registers
            const float gdx = g0x - px, gdy = g0y - py;
            const bool hit = human && fmaf(gdx, gdx, gdy * gdy) <= r * r;
            const bool sw = hit && gk == 2;
            const float t0 = g0x, t1 = g0y;
            g0x = sw ? g1x : g0x; g0y = sw ? g1y : g0y;
            g1x = sw ? t0 : g1x; g1y = sw ? t1 : g1y;
            gdirty = gdirty || sw;
            gx = hit ? g0x : gx; gy = hit ? g0y : gy;
        } else if (human) {
            // -- goal switch, forces_parallel.py:226-234 (on the incoming position)
            const float gdx = g0x - px, gdy = g0y - py;
            if (fmaf(gdx, gdx, gdy * gdy) <= r * r) { // |goals[i][0] - p| <= r ; rare, divergent
                const int k = gk;
                if (k <= 2) {                       // rotation of a list of 0, 1 or 2 goals: registers only
                    if (k == 2) {
                        const float t0 = g0x, t1 = g0y;
                        g0x = g1x; g0y = g1y; g1x = t0; g1y = t1;
                        gdirty = true;
                    }
                } else if (kmode & M_COMMIT_GOALS) {
                    const float r0 = gi[0], r1 = gi[1];
                    for (int g = 0; g + 1 < k; ++g) { gi[2 * g] = gi[2 * g + 2]; gi[2 * g + 1] = gi[2 * g + 3]; }
                    if (k > 0) { gi[2 * (k - 1)] = r0; gi[2 * (k - 1) + 1] = r1; }
                    g0x = gi[0]; g0y = gi[1];
                } else if (k > 1) {
                    g0x = gi[2]; g0y = gi[3];
                }
                gx = g0x; gy = g0y;
            }
        }
        STAMP(0);
        // -- social force, pair-once form (:87-133).  Lane i evaluates its partners at ring distance 1 .. (rows-1)/2
        //    inside its world (rows even: plus the antipodal one, evaluated by both ends).  The partner's share -f
        //    is added to accumulator slot [k mod UA][i + distance] of UA LDS rows: plain read-modify-write, no
        //    atomics -- one wavefront executes its LDS operations in order, so each of the UA chains is race-free.
        //    Index i + distance runs past the world's rows without a modulo: receiver j sums slots j and j + rows.
        //    Order of a substep: the first group's partner rows are requested from LDS, then everything that does not
        //    depend on this substep's social force is computed while they are in flight (rotation, desired and wall
        //    forces, the Farina torque, the new heading and its sine / cosine), then the partner groups, and only the
        //    short force-dependent tail (body-frame projection, Euler step, publish) follows the reaction sum.
        float fsx = 0.0f, fsy = 0.0f;
        // (the reaction accumulators need no zeroing: the first partner group of a substep stores into them)
        STAMP(8);
        // -- wall pairs: one (agent, polygon) pair per lane and pass, numbered in the prologue (two passes for more than 64 pairs)
        const bool wp_step = wp_on;
        if constexpr (LEAN == 2 && MAXT == 64) {
            if (wp_on && wp_count > 0) {
                {   // every agent's refreshed linear velocity, for the contact terms of a pair whose agent touches its polygon
                    float wvx = vx, wvy = vy;
                    if constexpr (HEADED > 0) { wvx = cs * bvx + (-sn) * bvy; wvy = sn * bvx + cs * bvy; }   // (the statements of part A below)
                    lds_wcv[tid] = make_float2(wvx, wvy);
                }
                LDS_ORDER_FENCE();
                const float4 law = *lds_wlaw;
                for (int k0 = 0; k0 < wp_count; k0 += 64) {                                             // (wave-uniform trip count)
                    const int k = k0 + tid;
                    if (k < wp_count) {
                        const int wr = lds_wrec[k];
                        const float2 wq = lds_wrs[k];
                        const float2 ap = *reinterpret_cast<const float2*>(&lds_p[cur * TP + (wr & 0xFF)]);   // the agent's incoming position
                        const float4* sgp = lds_seg + ((wr >> 8) & 0xFFF);
                        const float* sip = lds_sinv + ((wr >> 8) & 0xFFF);
                        float best = INFINITY, bdx = 0.0f, bdy = 0.0f;
                        for (int s0 = 0; s0 < a.Smax; ++s0) {                                            // first argmin over the polygon's segments
                            const float4 e = sgp[s0];
                            const float iv = sip[s0];
                            const float qx = ap.x - e.x, qy = ap.y - e.y;
                            const float t = fmaf(qx, e.z, qy * e.w) * iv;
                            const float ts = fminf(fmaxf(t, 0.0f), 1.0f);
                            const float ddx = fmaf(-ts, e.z, qx), ddy = fmaf(-ts, e.w, qy);
                            const float d = fmaf(ddx, ddx, ddy * ddy);
                            const bool better = d < best;
                            best = better ? d : best; bdx = better ? ddx : bdx; bdy = better ? ddy : bdy;
                        }
                        const float bcl = fmaxf(best, 1e-30f);
                        const float inv = rsq_fast(bcl);
                        const float dist = best * inv;
                        const float nx = bdx * inv, ny = bdy * inv;
                        const float rd = wq.x - dist + wq.y;
                        float2 f;
                        if (rd > -1.0e-3f) {   // (rare, lane-divergent) body contact: the k1 / k2 terms on the refined distance and the tangential velocity
                            const float2 cv = lds_wcv[wr >> 20];
                            const float dv = -(cv.y * nx - cv.x * ny);
                            const float m0 = fmaxf(0.0f, wq.x - dist_refined(bcl, inv) + wq.y);
                            const float fn = fmaf(law.x, exp2_fast(rd * law.y), law.z * m0);
                            const float ft = -(law.w * m0) * dv;
                            f = make_float2(fn * nx - ft * ny, fn * ny + ft * nx);
                        } else {
#pragma clang fp contract(off)
                            const float fn = law.x * exp2_fast(rd * law.y);
                            f = make_float2(fn * nx, fn * ny);
                        }
                        lds_wres[k] = f;
                    }
                }
                LDS_ORDER_FENCE();
            }
        }
        // -- part A of the per-agent update: everything that does not need this substep's social force
        const float c = cs, s = sn;          // rotation matrix of the incoming heading, :254-256
        float cvx = vx, cvy = vy;            // refreshed linear velocity
        float fdx = 0.0f, fdy = 0.0f, fox = 0.0f, foy = 0.0f;
        float th_n = th, sn_n = sn, cs_n = cs, torque_a = 0.0f;
        if (human) {
            if constexpr (HEADED > 0) {
                cvx = c * bvx + (-s) * bvy;
                cvy = s * bvx + c * bvy;
            }
            // -- desired force, :23-40
            {
                const float dx = gx - px, dy = gy - py;
                const float d2 = fmaf(dx, dx, dy * dy);
                const float inv = rsq_fast(fmaxf(d2, 1e-30f));
                const float wx = m_tau * (dx * inv * vd - cvx), wy = m_tau * (dy * inv * vd - cvy);
                const bool far_ = d2 * inv > r;   // selects, not a branch: a data-dependent branch costs ~55 cycles of latency
                fdx = far_ ? wx : 0.0f;
                fdy = far_ ? wy : 0.0f;
            }
            // -- obstacle force: closest point per polygon :236-252, then :136-162
            if (obst != nullptr && wp_step) {
                // (wall pairs) my pairs' forces in polygon order -- the reference's order of summation; a polygon beyond my reach left its zero
#pragma clang fp contract(off)
                for (int o = 0; o < a.O; ++o) {
                    const float2 f = lds_wres[(wp_mine >> (8 * o)) & 0xFFu];
                    fox = fox + f.x;
                    foy = foy + f.y;
                }
                fox *= inv_O; foy *= inv_O;
            } else if (obst != nullptr) {
                for (int o = 0; o < a.O; ++o) {
                    // first argmin over the polygon's segments, on squared distances (same order)
                    float best = INFINITY, bdx = 0.0f, bdy = 0.0f;   // the first slot always beats +inf: a first argmin
                    if (a.seg_tab > 0) {
                        {   // nobody of this wavefront within reach of the polygon: its force is < 5e-13 N on everyone, skip it
                            const float4 pc = lds_poly[(sbase / a.Smax) + o];
                            const float dxc = px - pc.x, dyc = py - pc.y, lim = pc.z + wall_cut;
                            if (__builtin_amdgcn_ballot_w64(fmaf(dxc, dxc, dyc * dyc) < lim * lim) == 0) continue;
                        }
                        // branch-free, four slots per trip (then two, then one): the LDS reads of a trip (uniform addresses,
                        // broadcasts) are issued before its arithmetic; a NaN slot is a far-away degenerate segment in the table
                        const float4* sgp = lds_seg + sbase + o * a.Smax;
                        const float* sip = lds_sinv + sbase + o * a.Smax;
                        auto trip = [&](int s0, auto width) {
                            constexpr int NW = decltype(width)::value;
                            float4 e[NW];
                            float iv[NW];
#pragma unroll
                            for (int j = 0; j < NW; ++j) { e[j] = sgp[s0 + j]; iv[j] = sip[s0 + j]; }
#pragma unroll
                            for (int j = 0; j < NW; ++j) {
                                const float qx = px - e[j].x, qy = py - e[j].y;
                                const float t = fmaf(qx, e[j].z, qy * e[j].w) * iv[j];
                                const float ts = fminf(fmaxf(t, 0.0f), 1.0f);
                                const float ddx = fmaf(-ts, e[j].z, qx), ddy = fmaf(-ts, e[j].w, qy);   // p - (a + ts e)
                                const float d = fmaf(ddx, ddx, ddy * ddy);
                                const bool better = d < best;
                                best = better ? d : best;
                                bdx = better ? ddx : bdx;
                                bdy = better ? ddy : bdy;
                            }
                        };
                        int s0 = 0;
                        // polygons of up to six slots (triangles to hexagons) in ONE trip: all their LDS reads are in flight together
                        if (a.Smax == 5) { trip(0, std::integral_constant<int, 5>{}); s0 = 5; }
                        else if (a.Smax == 6) { trip(0, std::integral_constant<int, 6>{}); s0 = 6; }
                        else if (a.Smax == 3) { trip(0, std::integral_constant<int, 3>{}); s0 = 3; }
                        for (; s0 + 4 <= a.Smax; s0 += 4) trip(s0, std::integral_constant<int, 4>{});
                        if (s0 + 2 <= a.Smax) { trip(s0, std::integral_constant<int, 2>{}); s0 += 2; }
                        if (s0 < a.Smax) trip(s0, std::integral_constant<int, 1>{});
                    } else {
                        for (int sg = 0; sg < a.Smax; ++sg) {
                            const float4 seg = *reinterpret_cast<const float4*>(obst + ((long)o * a.Smax + sg) * 4);
                            const float x1 = seg.x, y1 = seg.y, ex = seg.z - seg.x, ey = seg.w - seg.y;
                            float d, ddx = 0.0f, ddy = 0.0f;
                            if (isnan(seg.x)) {
                                d = 3.0e38f; // NaN slot: the reference stores iinfo(int64).max as the distance (:247)
                            } else {
                                const float t = ((px - x1) * ex + (py - y1) * ey) * rcp_fast(fmaf(ex, ex, ey * ey));
                                const float ts = fminf(fmaxf(t, 0.0f), 1.0f);
                                ddx = px - fmaf(ts, ex, x1); ddy = py - fmaf(ts, ey, y1);
                                d = fmaf(ddx, ddx, ddy * ddy);
                            }
                            if (d < best) { best = d; bdx = ddx; bdy = ddy; }
                        }
                    }
                    const float bcl = fmaxf(best, 1e-30f);
                    const float inv = rsq_fast(bcl);
                    const float dist = best * inv;
                    const float nx = bdx * inv, ny = bdy * inv;
                    const float rd = r - dist + safety;
                    // Helbing walls: the body-contact terms (k1, k2; the refined distance, the tangential velocity) are exact zeros
                    // unless somebody overlaps the polygon -- a wave vote, with a millimetre of margin for the unrefined distance; the
                    // short branch leaves the same bits as the long one with m0 = 0 (fn = A e + k1 0, ft = -(k2 0) dv = -+0: the
                    // products rounded once, then the sums).  8192 x 50 + 3 polygons: 221.5 -> 215.5 us
                    if (obs_type != 0 || __builtin_amdgcn_ballot_w64(rd > -1.0e-3f) != 0) {
                        const float dv = -(cvy * nx - cvx * ny);                 // -(v . t), t = (-ny, nx)
                        const float m0 = fmaxf(0.0f, r - dist_refined(bcl, inv) + safety);
                        const float fn = fmaf(Aw, exp2_fast(rd * cBw), k1 * m0);
                        float ft;                                                 // coefficient of t
                        if (obs_type == 0) ft = -(k2 * m0) * dv;
                        else ft = (-Cw * exp2_fast(rd * cDw) - k2 * m0) * dv;
                        fox += fn * nx - ft * ny;
                        foy += fn * ny + ft * nx;
                    } else {
#pragma clang fp contract(off)
                        const float fn = Aw * exp2_fast(rd * cBw);
                        const float tx = fn * nx, ty = fn * ny;   // (not contracted into the sums: the long branch rounds the products too)
                        fox = fox + tx;
                        foy = foy + ty;
                    }
                }
                fox *= inv_O; foy *= inv_O;
            }
            if constexpr (HEADED > 0) {
                th_n = wrap_angle(fmaf(om, dt, th));   // :278; the new heading needs omega of the incoming row only
                sincos_fast(th_n, sn_n, cs_n);
            }
            if constexpr (HEADED == 1) {               // Farina: the torque follows the desired force alone, :165-182
                const float kf = klam * norm2(fdx, fdy);
                const float k_theta = inertia * kf;
                const float k_omega = inertia * (1.0f + alpha) * sqrt_fast(kf * inv_alpha);
                const float delta = atan2_fast(s * fdx - c * fdy, c * fdx + s * fdy);
                torque_a = -k_theta * delta - k_omega * om;
            }
        }
        STAMP(1);
        if constexpr (N3L) {
            if (valid) {
                float2* acc = lds_acc + pbase + row + 1;                 // acc[u * 2T + k]: that partner's slot in row u
                float2* accw = lds_acc + pbase;                          // compile-time row builds: slot aoff[k] of the world
                float ex = 0.0f, ey = 0.0f, rdmax = -1.0e30f;   // max over my partners of rs_j - dist
                if constexpr (!(LEAN && NO_WALLS)) { if (Hf >= UA) { fetch(qa, va, 0); fetch_pp(pa, 0); } }
                // (fx, fy): the force on me, in units of sign(A_me); (gx, gy): what goes to the partner's reaction slot -- the same force
                // when all parameters are equal (exact antisymmetry), the partner's OWN law on the shared geometry otherwise (PP)
                auto pair_once = [&](const float4 q, const float2 vq, const float4 pj, float& fx, float& fy, float& gx, float& gy) {
                    const float dx = px - q.x, dy = py - q.y;
                    if constexpr (SOC == 2) {
                        pair_force_moussaid_once(sp, dx, dy, vx - vq.x, vy - vq.y, my_rs + q.z, fx, fy);
                        gx = fx; gy = fy;
                    } else {
                        // [A e^{rd/B}] n + [C e^{rd/D}] t, in units of sign(A); the k1 / k2 contact parts are exact
                        // zeros unless rd > 0 and are added by the contact pass below
                        const float d2 = fmaf(dx, dx, dy * dy);
                        const float inv = rsq_fast(d2);
                        const float rd = fmaf(-d2, inv, q.z);                     // rs_j - dist  (= rij - dist - rs_i)
                        const float ga = exp2_fast(fmaf(rd, sp.cB, lAi)) * inv;   // |A| e^{(rij - dist)/B} / dist
                        fx = ga * dx; fy = ga * dy;
                        if constexpr (SOC == 1) {
                            const float gc = exp2_fast(fmaf(rd, sp.cD, lCi)) * (inv * sp.sAC); // +-|C| e^{(rij - dist)/D} / dist
                            fx = fmaf(-gc, dy, fx); fy = fmaf(gc, dx, fy);                       // along t = (-ny, nx)
                        }
                        if constexpr (PP) {
                            const float rdi = fmaf(-d2, inv, my_rs);                  // rs_me - dist: the partner's exponent offset holds its own rs
                            const float gb = exp2_fast(fmaf(rdi, pj.y, pj.x)) * inv;
                            gx = gb * dx; gy = gb * dy;
                            if constexpr (SOC == 1) {
                                const float gd = exp2_fast(fmaf(rdi, pj.w, pj.z)) * (inv * q.w);   // q.w: sign(C_j) sign(A_j)
                                gx = fmaf(-gd, dy, gx); gy = fmaf(gd, dx, gy);
                            }
                        } else {
                            gx = fx; gy = fy;
                        }
                        rdmax = fmaxf(rdmax, rd);
                    }
                };
                // groups of UA partners; the partner rows of the NEXT group are fetched while the current group is
                // evaluated (two register sets, a compiler memory barrier pins the prefetch), and the accumulator
                // slots are read before the evaluation and written after it
                // accumulator row pitch: 2 KiB, out of reach of the ds_read2_b64 / ds_write2_b64 offset fields on purpose:
                // the paired forms take 8 / 13 LDS cycles, two single b64 accesses 4 / 12 (MI355X_MICROARCH.md, LDS table)
                constexpr int AR = ACC_PITCH;
                auto group = [&](const float4 (&q)[UA], const float2 (&vq)[UA], const float4 (&pq)[PP ? UA : 1], float2 (&ac)[UA], const int (&so)[UA]) {
                    if constexpr (SOC == 2) {
#pragma unroll
                        for (int u = 0; u < UA; ++u) {
                            float fx, fy, gx, gy;
                            pair_once(q[u], vq[u], make_float4(0.0f, 0.0f, 0.0f, 0.0f), fx, fy, gx, gy);
                            ex += fx; ey += fy;
                            ac[u].x += gx; ac[u].y += gy;
                        }
                    } else {
                        // Helbing / Guo: the UA partners advance stage by stage (scheduling barriers between the
                        // stages), so the UA v_rsq_f32 and the UA v_exp_f32 issue back to back and each result is
                        // consumed ~UA instructions later instead of right behind its transcendental
                        float dx[UA], dy[UA], d2[UA], inv[UA], rd[UA], ea[UA], ec[UA], eb[UA], ed[UA];
#pragma unroll
                        for (int u = 0; u < UA; ++u) {
                            dx[u] = px - q[u].x; dy[u] = py - q[u].y;
                            d2[u] = fmaf(dx[u], dx[u], dy[u] * dy[u]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < UA; ++u) inv[u] = rsq_fast(d2[u]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < UA; ++u) {
                            rd[u] = fmaf(-d2[u], inv[u], q[u].z);
                            ea[u] = fmaf(rd[u], sp.cB, lAi);
                            if constexpr (SOC == 1) ec[u] = fmaf(rd[u], sp.cD, lCi);
                            if constexpr (PP) {   // the partner's side of the same pair, from its parameter row
                                const float rdi = fmaf(-d2[u], inv[u], my_rs);
                                eb[u] = fmaf(rdi, pq[u].y, pq[u].x);
                                if constexpr (SOC == 1) ed[u] = fmaf(rdi, pq[u].w, pq[u].z);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < UA; ++u) {
                            ea[u] = exp2_fast(ea[u]);
                            if constexpr (SOC == 1) ec[u] = exp2_fast(ec[u]);
                            if constexpr (PP) {
                                eb[u] = exp2_fast(eb[u]);
                                if constexpr (SOC == 1) ed[u] = exp2_fast(ed[u]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < UA; ++u) {
                            const float ga = ea[u] * inv[u];
                            if constexpr (PP) {
                                const float gb = eb[u] * inv[u];
                                if constexpr (SOC == 1) {
                                    const float gc = ec[u] * (inv[u] * sp.sAC), gd = ed[u] * (inv[u] * q[u].w);
                                    ex += fmaf(-gc, dy[u], ga * dx[u]); ey += fmaf(gc, dx[u], ga * dy[u]);
                                    ac[u].x += fmaf(-gd, dy[u], gb * dx[u]); ac[u].y += fmaf(gd, dx[u], gb * dy[u]);
                                } else {
                                    ex = fmaf(ga, dx[u], ex); ey = fmaf(ga, dy[u], ey);
                                    ac[u].x = fmaf(gb, dx[u], ac[u].x); ac[u].y = fmaf(gb, dy[u], ac[u].y);
                                }
                            } else if constexpr (SOC == 1) {
                                const float gc = ec[u] * (inv[u] * sp.sAC);
                                const float fx = fmaf(-gc, dy[u], ga * dx[u]), fy = fmaf(gc, dx[u], ga * dy[u]);
                                ex += fx; ey += fy;
                                ac[u].x += fx; ac[u].y += fy;
                            } else {
                                // Helbing: f = ga d goes into my sum and into the partner's slot by one FMA each (4 instead of
                                // 2 multiplies + 4 adds; the two sums see the same product, rounded once per sum)
                                ex = fmaf(ga, dx[u], ex); ey = fmaf(ga, dy[u], ey);
                                ac[u].x = fmaf(ga, dx[u], ac[u].x); ac[u].y = fmaf(ga, dy[u], ac[u].y);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < UA; ++u) rdmax = fmaxf(rdmax, rd[u]);   // (pairs of them fuse into v_max3_f32)
                    }
#pragma unroll
                    for (int u = 0; u < UA; ++u) accw[u * AR + so[u]] = ac[u];
                    LDS_ORDER_FENCE(); // the next group's slots were written by other lanes in this group
                };
                // reaction slot of the partner at ring distance k + 1: its row index inside the world, (row + 1 + k) mod rows
                auto slot_rt = [&](int k) { const int t = row + 1 + k; return t >= rows ? t - rows : t; };
                int k0 = 0;
                if constexpr (ROWS_CT > 0) {
                    // rows known at compile time: the groups are laid out one after the other, no loop, no scalar branches
                    constexpr int NG = ((ROWS_CT - 1) / 2) / UA;
                    float4 qb[UA], pb[PP ? UA : 1];
                    float2 vb[UA], ac[UA];
                    for_each_index([&](auto gtag) {
                        constexpr int g = decltype(gtag)::value;
                        int so[UA];
#pragma unroll
                        for (int u = 0; u < UA; ++u) {
                            so[u] = aoff[g * UA + u];
                            ac[u] = g == 0 ? make_float2(0.0f, 0.0f) : accw[u * AR + so[u]];   // the first group stores: no zeroing pass
                        }
                        if constexpr (g + 1 < NG) {
                            if constexpr (g & 1) { fetch(qa, va, (g + 1) * UA); fetch_pp(pa, (g + 1) * UA); } else { fetch(qb, vb, (g + 1) * UA); fetch_pp(pb, (g + 1) * UA); }
                        }
                        asm volatile("" ::: "memory");
                        if constexpr (g & 1) group(qb, vb, pb, ac, so); else group(qa, va, pa, ac, so);
                    }, std::make_integer_sequence<int, NG>{});
                    k0 = NG * UA;
                } else if (Hf >= UA) {
                    float4 qb[UA], pb[PP ? UA : 1];
                    float2 vb[UA], ac[UA];
                    int so[UA];
                    bool first = true;
                    for (;;) {
#pragma unroll
                        for (int u = 0; u < UA; ++u) {
                            so[u] = slot_rt(k0 + u);
                            ac[u] = first ? make_float2(0.0f, 0.0f) : accw[u * AR + so[u]];   // (uniform: the first group stores)
                        }
                        first = false;
                        const bool more_b = k0 + 2 * UA <= Hf;
                        if (more_b) { fetch(qb, vb, k0 + UA); fetch_pp(pb, k0 + UA); }
                        asm volatile("" ::: "memory");
                        group(qa, va, pa, ac, so);
                        k0 += UA;
                        if (!more_b) break;
#pragma unroll
                        for (int u = 0; u < UA; ++u) { so[u] = slot_rt(k0 + u); ac[u] = accw[u * AR + so[u]]; }
                        const bool more_a = k0 + 2 * UA <= Hf;
                        if (more_a) { fetch(qa, va, k0 + UA); fetch_pp(pa, k0 + UA); }
                        asm volatile("" ::: "memory");
                        group(qb, vb, pb, ac, so);
                        k0 += UA;
                        if (!more_a) break;
                    }
                }
                if constexpr (ROWS_CT > 0) {
                    constexpr int HC = (ROWS_CT - 1) / 2;
                    if constexpr (HC % UA != 0) {   // (UA = 2: at most one partner left over)
                        static_assert(HC % UA == 1 || UA > 2, "remainder of the compile-time group layout");
#pragma unroll
                        for (int k = (HC / UA) * UA; k < HC; ++k) {
                            float fx, fy, gx, gy;
                            float2 vq = make_float2(0.0f, 0.0f);
                            if constexpr (SOC == 2) vq = rv[k];
                            float4 qk;
                            if constexpr (PP) qk = rp[k];
                            else { const float2 xy = *reinterpret_cast<const float2*>(&rp[k]); qk = make_float4(xy.x, xy.y, rsj[k], 0.0f); }
                            pair_once(qk, vq, PP ? rpp[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f), fx, fy, gx, gy);
                            ex += fx; ey += fy;
                            float2 ac = (k == 0) ? make_float2(0.0f, 0.0f) : accw[aoff[k]];   // (partner 0 without a full group: first writer of row 0)
                            ac.x += gx; ac.y += gy;
                            accw[aoff[k]] = ac;
                            LDS_ORDER_FENCE();
                        }
                    }
                } else
                for (int k = k0; k < Hf; ++k) {   // (fewer than UA partners left; k - k0 is the accumulator row they go to when no group ran)
                    float fx, fy, gx, gy;
                    float2 vq = make_float2(0.0f, 0.0f);
                    if constexpr (SOC == 2) vq = rv[k];
                    pair_once(rp[k], vq, PP ? rpp[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f), fx, fy, gx, gy);
                    ex += fx; ey += fy;
                    const int so = slot_rt(k);
                    float2 ac = (k == 0) ? make_float2(0.0f, 0.0f) : accw[so];   // partner 0 of a world without a full group: first writer of row 0
                    ac.x += gx; ac.y += gy;
                    accw[so] = ac;
                    LDS_ORDER_FENCE();
                }
                if ((rows & 1) == 0) { // antipodal partner: each end evaluates it for itself (lean build: row fetched at the top)
                    float fx, fy;
                    if constexpr (!(LEAN && NO_WALLS)) {
                        if constexpr (ROWS_CT > 0 && N3L) { const float2 xy = *reinterpret_cast<const float2*>(&rp[(ROWS_CT - 1) / 2]); qz = make_float4(xy.x, xy.y, rsj[(ROWS_CT - 1) / 2], 0.0f); }
                        else qz = rp[Hf];
                        if constexpr (SOC == 2) vz = rv[Hf];
                    }
                    float gx, gy;   // (each end evaluates its own side: nothing handed over)
                    pair_once(qz, vz, make_float4(0.0f, 0.0f, 0.0f, 0.0f), fx, fy, gx, gy);
                    ex += fx; ey += fy;
                }
                STAMP(9);
                LDS_ORDER_FENCE(); // the reaction slots of this lane were written by its partners
                const float2* rr = lds_acc + pbase + row;
                // one slot per accumulator row; a row no group wrote in this substep (fewer partners than rows: worlds of <= 4 rows) holds
                // nothing.  (Left-over partners of a world without any full group all go to row 0 -- one of them at most with UA = 2.)
                const int nr = Hf >= UA ? UA : (Hf > 0 ? 1 : 0);
                float rx = 0.0f, ry = 0.0f;
                if (nr > 0) { const float2 lo = rr[0]; rx = lo.x; ry = lo.y; }
#pragma unroll
                for (int u = 1; u < UA; ++u)
                    if (u < nr) { const float2 lo = rr[u * AR]; rx += lo.x; ry += lo.y; }
                fsx = ex - rx; fsy = ey - ry;
                STAMP(10);
                if constexpr (SOC != 2) {
                    fsx *= sp.sA; fsy *= sp.sA;
                    if (__builtin_amdgcn_ballot_w64(rdmax > -my_rs) != 0) { // contact somewhere in this wavefront
#ifdef CS_STAMPS
                        st_acc[11] += 1000;   // (diagnostic: how often the contact pass runs, per mille of the substeps)
#endif
                        if constexpr (!PP) {
                            // Equal parameters: every term of the all-partners sum below is an exact zero except those of the rows that
                            // touch somebody, so only those rows are gone through -- as SOURCES, broadcast from their lane's registers (the
                            // published position and, here, velocity of a row ARE its lane's registers), every lane of the world adding its
                            // own term from the source in the order of the loop it replaces (increasing row): the same bits, ~25 instructions
                            // per touching row instead of 16 x rows.  The launch waits for the wavefront that takes this path.
                            auto bcast = [](float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); };
                            // S: the lower ends of the overlapping pairs (they saw it in the pair loop); T: S and whoever overlaps one of S
                            unsigned long long T2 = __builtin_amdgcn_ballot_w64(rdmax > -my_rs);
                            for (unsigned long long m2 = T2; m2 != 0; m2 &= m2 - 1) {
                                const int sa = __builtin_ctzll(m2);
                                const float ax_ = bcast(px, sa), ay_ = bcast(py, sa), ars = bcast(my_rs, sa);
                                const float dx = px - ax_, dy = py - ay_, lim = my_rs + ars + 1.0e-3f;
                                const bool same = sa >= base && sa < base + rows && sa != tid;
                                T2 |= __builtin_amdgcn_ballot_w64(same && fmaf(dx, dx, dy * dy) < lim * lim);
                            }
                            for (unsigned long long m2 = T2; m2 != 0; m2 &= m2 - 1) {
                                const int sa = __builtin_ctzll(m2);
                                const float4 q = make_float4(bcast(px, sa), bcast(py, sa), bcast(my_rs, sa), 0.0f);
                                const float2 vj = make_float2(bcast(vx, sa), bcast(vy, sa));
                                const bool same = sa >= base && sa < base + rows && sa != tid;
                                const float dx = px - q.x, dy = py - q.y;
                                const float d2 = fmaf(dx, dx, dy * dy);
                                const float inv = rsq_fast(d2);
                                const float m0 = fmaxf(0.0f, (my_rs + q.z) - dist_refined(d2, inv));
                                const float nx = dx * inv, ny = dy * inv;
                                const float dv = (vj.y - vy) * nx - (vj.x - vx) * ny;     // (v_j - v_i) . t
                                const float fn = sp.k1 * m0, ft = (sp.k2 * m0) * dv;
                                const float tx = fn * nx - ft * ny, ty = fn * ny + ft * nx;
                                fsx += same ? tx : 0.0f;
                                fsy += same ? ty : 0.0f;
                            }
                        } else {
                        const float4* pp = lds_p + cur * TP + pbase;
                        float2* pvel = lds_v + cur * TP + pbase;
                        if constexpr (VEL_ON_DEMAND) {
                            pvel[row] = make_float2(vx, vy); // every lane of the wavefront is here: publish the velocities now
                            LDS_ORDER_FENCE();
                        }
                        // per-agent parameters: my refreshed velocity, partner j < i refreshed, j > i stored (prange == range order)
                        const float vix = PP ? cvx : vx, viy = PP ? cvy : vy;
                        const float2* vr = lds_vr + cur * T + base;
#pragma unroll 8
                        for (int j = 0; j < rows; ++j) { // rare path, but the launch waits for the wavefront that takes it: eight partner rows requested per trip
                            const float4 q = pp[j];
                            float2 vj = pvel[j];
                            if constexpr (PP && HEADED > 0) { if (j < row) vj = vr[j]; }
                            const float dx = px - q.x, dy = py - q.y;
                            const float d2 = (j == row) ? 1.0e30f : fmaf(dx, dx, dy * dy);
                            const float inv = rsq_fast(d2);
                            const float m0 = fmaxf(0.0f, (my_rs + q.z) - dist_refined(d2, inv));
                            const float nx = dx * inv, ny = dy * inv;
                            const float dv = (vj.y - viy) * nx - (vj.x - vix) * ny;     // (v_j - v_i) . t
                            const float fn = sp.k1 * m0, ft = (sp.k2 * m0) * dv;
                            fsx += fn * nx - ft * ny;
                            fsy += fn * ny + ft * nx;
                        }
                        }
                    }
                }
            }
        }
        STAMP(2);
        if (human) {
            // -- social force, every lane evaluates all its partners: O(N) rows broadcast from LDS, :43-84
            if constexpr (!N3L) {
                // all_params_equal: every row's stored velocity (the reference evaluates all pairs
                // before any in-place refresh); else: my refreshed velocity, partner j<i refreshed,
                // j>i stored (prange == range order; identical from the 2nd fused substep on)
                const float vix = PEQ ? vx : cvx, viy = PEQ ? vy : cvy;
                const float4* pp = lds_p + cur * TP + pbase;
                const float2* pvel = lds_v + cur * TP + pbase;
                const float2* vr = lds_vr + cur * T + base;
                auto partner_vel = [&](int j) {
                    float2 v2 = pvel[j];
                    if constexpr (!PEQ && HEADED > 0) { if (j < row) v2 = vr[j]; }
                    return v2;
                };
                if constexpr (SOC == 2) {
                    for (int j = 0; j < rows; ++j) {
                        const float4 q = pp[j];
                        const float2 vj = partner_vel(j);
                        pair_force_moussaid(sp, px, py, vix, viy, q.x, q.y, vj.x, vj.y, my_rs + q.z, j == row, fsx, fsy);
                    }
                } else {
                    // Helbing / Guo (:117-118).  Per partner the force is
                    //   [A e^{rd/B} + k1 max(0,rd)] n + [C e^{rd/D} + k2 max(0,rd) dv] t      (C = 0: Helbing)
                    // Main loop: the exponential parts for every partner, branch-free (one ds_read_b128,
                    // ~13 VALU + 2 transcendental ops each; A and C folded into the exponent; the next group
                    // of four is fetched from LDS while the current one is evaluated).  The k1 / k2 contact
                    // parts are exact zeros unless rd > 0: they are added by a second pass that a wavefront
                    // runs only when one of its lanes touched a partner in this substep.
                    float eax = 0.0f, eay = 0.0f, ecx = 0.0f, ecy = 0.0f, rdmax = -1.0e30f;   // max over the partners of rs_j - dist
                    constexpr int U = 8;                  // partners in flight per lane (independent rsq -> exp chains)
                    const float4* rp = pp + row + 1;      // partner k of mine = row (i + 1 + k) mod rows, k = 0 .. rows-2
                    const int np = rows - 1;
                    auto group = [&](const float4 (&q)[U], int rem, auto tail_tag) {
                        constexpr bool TAIL = decltype(tail_tag)::value;
                        float rdk[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const float dx = px - q[u].x, dy = py - q[u].y;
                            float d2 = fmaf(dx, dx, dy * dy);
                            if constexpr (TAIL) d2 = (u >= rem) ? 1.0e30f : d2; // padding slot (wave-uniform): dist 1e15 -> 0
                            const float inv = rsq_fast(d2);
                            const float rd = fmaf(-d2, inv, q[u].z);                  // rs_j - dist: my own rs sits in the exponent offsets (lAi, lCi)
                            const float ga = exp2_fast(fmaf(rd, sp.cB, lAi)) * inv;   // |A| e^{(rij - dist)/B} / dist
                            eax = fmaf(ga, dx, eax); eay = fmaf(ga, dy, eay);
                            if constexpr (SOC == 1) {
                                const float gc = exp2_fast(fmaf(rd, sp.cD, lCi)) * inv;   // |C| e^{(rij - dist)/D} / dist
                                ecx = fmaf(-gc, dy, ecx); ecy = fmaf(gc, dx, ecy);       // along t = (-ny, nx)
                            }
                            rdk[u] = rd;
                        }
#pragma unroll
                        for (int u = 0; u < U; u += 2) rdmax = fmaxf(fmaxf(rdmax, rdk[u]), rdk[u + 1]); // v_max3_f32
                    };
                    auto run = [&](const float4 (&q)[U], int rem) {
                        if (rem >= U) group(q, U, std::false_type{});
                        else group(q, rem, std::true_type{});
                    };
                    // two register sets, filled alternately: the next group's LDS reads are in flight
                    // while the current one is evaluated (no copies, no re-load at the point of use)
                    float4 qa[U], qb[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) qa[u] = rp[u];
                    for (int k0 = 0; k0 < np; k0 += 2 * U) {
                        if (k0 + U < np) {
#pragma unroll
                            for (int u = 0; u < U; ++u) qb[u] = rp[k0 + U + u]; // rows past np are finite padding
                        }
                        asm volatile("" ::: "memory");
                        run(qa, np - k0);
                        if (k0 + U >= np) break;
                        if (k0 + 2 * U < np) {
#pragma unroll
                            for (int u = 0; u < U; ++u) qa[u] = rp[k0 + 2 * U + u];
                        }
                        asm volatile("" ::: "memory");
                        run(qb, np - k0 - U);
                    }
                    fsx = sp.sA * eax; fsy = sp.sA * eay;
                    if constexpr (SOC == 1) { fsx = fmaf(sp.sC, ecx, fsx); fsy = fmaf(sp.sC, ecy, fsy); }
                    if (__builtin_amdgcn_ballot_w64(rdmax > -my_rs) != 0) { // contact somewhere in this wavefront
                        for (int j = 0; j < rows; ++j) {
                            const float4 q = pp[j];
                            const float2 vj = partner_vel(j);
                            const float dx = px - q.x, dy = py - q.y;
                            const float d2 = (j == row) ? 1.0e30f : fmaf(dx, dx, dy * dy);
                            const float inv = rsq_fast(d2);
                            const float m0 = fmaxf(0.0f, (my_rs + q.z) - dist_refined(d2, inv));
                            const float nx = dx * inv, ny = dy * inv;
                            const float dv = (vj.y - viy) * nx - (vj.x - vix) * ny;     // (v_j - v_i) . t
                            const float fn = sp.k1 * m0, ft = (sp.k2 * m0) * dv;
                            fsx += fn * nx - ft * ny;
                            fsy += fn * ny + ft * nx;
                        }
                    }
                }
            }
            STAMP(6);
            // -- part B: total force, body frame, torque  :262-271, :165-182
            // (no-walls builds: fo = 0 is a compile-time fact, and x + 0 is not an identity the compiler may drop)
            const float fix = NO_WALLS ? fdx + fsx : fdx + fox + fsx, fiy = NO_WALLS ? fdy + fsy : fdy + foy + fsy;
            const float fpx = NO_WALLS ? fsx : fox + fsx, fpy = NO_WALLS ? fsy : foy + fsy;   // what acts across the heading: fo + fs
            float gfx = fix, gfy = fiy, torque = torque_a;
            if constexpr (HEADED > 0) {
                if constexpr (HEADED == 2) {  // torque on the total force
                    const float kf = klam * norm2(fix, fiy);
                    const float k_theta = inertia * kf;
                    const float k_omega = inertia * (1.0f + alpha) * sqrt_fast(kf * inv_alpha);
                    // bound_angle(theta - atan2(Fy, Fx)) is the signed angle from F to the heading:
                    // atan2(|F| sin(theta - phi), |F| cos(theta - phi)) -- one atan2, no wrap needed
                    const float delta = atan2_fast(s * fix - c * fiy, c * fix + s * fiy);
                    torque = -k_theta * delta - k_omega * om;
                }
                gfx = fix * c + fiy * s;
                gfy = ko * (fpx * (-s) + fpy * c) - kd * bvy;
            }
            // -- explicit Euler, :273-283 (position uses the velocity stored in the incoming row)
            const float in_vx = cvx, in_vy = cvy; // what the reference leaves in agents_state[i,3:5]
            px += vx * dt; py += vy * dt;
            if constexpr (HEADED > 0) {
                th = th_n;
                bvx = fmaf(gfx, dt_m, bvx); bvy = fmaf(gfy, dt_m, bvy);
                const float nb2 = fmaf(bvx, bvx, bvy * bvy);
                const float ninv = rsq_fast(fmaxf(nb2, 1e-30f));
                if (nb2 * ninv > vd) { const float sc = vd * ninv; bvx *= sc; bvy *= sc; }
                om = fmaf(torque, dt_inertia, om);
                sn = sn_n; cs = cs_n;
                vx = cs * bvx + (-sn) * bvy;
                vy = sn * bvx + cs * bvy;
            } else {
                vx = fmaf(gfx, dt_m, vx); vy = fmaf(gfy, dt_m, vy);
                const float nb2 = fmaf(vx, vx, vy * vy);
                const float ninv = rsq_fast(fmaxf(nb2, 1e-30f));
                if (nb2 * ninv > vd) { const float sc = vd * ninv; vx *= sc; vy *= sc; }
            }
            if ((kmode & M_MUTATE_INPUT) && sub == 0) {
                float* si = a.Sin + sidx * a.in_as;
                if (HEADED > 0) { si[3 * a.in_fs] = in_vx; si[4 * a.in_fs] = in_vy; }
                si[10 * a.in_fs] = gx; si[11 * a.in_fs] = gy;
            }
            publish(nxt);
            publish_v(nxt);
            if constexpr (!PEQ && HEADED > 0) lds_vr[nxt * T + tid] = make_float2(vx, vy);
            STAMP(3);
        } else if (is_robot) {
            // the robot's move of the NEXT substep happens before that substep's update_humans
            if (robot_moves && sub + 1 < a.nsub) robot_step();
            publish(nxt);
            publish_v(nxt);
            if constexpr (!PEQ && HEADED > 0) lds_vr[nxt * T + tid] = make_float2(vx, vy);
        }
        // block of one wavefront: its LDS operations execute in order, program order is all the next substep needs
        // (no s_waitcnt lgkmcnt(0) + s_barrier on the published rows)
        if constexpr (MAXT == 64) LDS_ORDER_FENCE(); else __syncthreads();
        STAMP(4);
        // -- parallel-traffic respawn, motion_model_manager.py:407-422 (sequential inside a world)
        if (a.flags & CS_RESPAWN) {
            const float rdx = px - g0x, rdy = py - g0y;
            const int flag = (human && respawn_here && fmaf(rdx, rdx, rdy * rdy) < 9.0f) ? 1 : 0; // |p - g| < 3
            if constexpr (MAXT == 64) {
                // one wavefront holds whole worlds: no barrier, no serial lane.  The reference respawns the flagged
                // humans of a world in index order, each behind everybody else (:411-417): x_0 = max(max_x + 2 max_r,
                // bound), and the c-th flagged one (c lower-indexed flagged rows in its world) lands at
                // x_c = max(x_{c-1} + 2 max_r, bound) because x_{c-1} is then the rightmost human.
                // ONE data-dependent branch in the common (nobody flagged) case: the vote is taken inside the branch, where
                // only the flagged lanes are active -- exactly the lanes it has to count
                if (flag) {
                    const unsigned long long fm = __builtin_amdgcn_ballot_w64(true);
                    const unsigned long long wm = (rows >= 64 ? ~0ull : ((1ull << rows) - 1ull)) << base;
                    const int c = __builtin_popcountll(fm & wm & ((1ull << tid) - 1ull));
                    const float4* pvn = lds_p + nxt * TP + pbase;
                    float mx = pvn[0].x, mr = pvn[0].z;
                    // (the rows requested together -- all of them when the row count is a compile-time constant, eight per trip otherwise:
                    // one at a time was 24 dependent LDS round trips per respawn, and the launch waits for its slowest wavefront: 31.1 -> 29.9 us at cfg3)
                    if constexpr (ROWS_CT > 0 && ROWS_CT <= 32) {
#pragma unroll
                        for (int j = 1; j < ROWS_CT; ++j) {
                            if (j < n) { mx = fmaxf(mx, pvn[j].x); mr = fmaxf(mr, pvn[j].z); }
                        }
                    } else {
#pragma unroll 8
                        for (int j = 1; j < n; ++j) {
                            mx = fmaxf(mx, pvn[j].x);
                            mr = fmaxf(mr, pvn[j].z);
                        }
                    }
                    if (robot_row) { // consider_robot: the robot where it stands in THIS substep
                        const float4 qr = lds_p[cur * TP + pbase + n];
                        mx = fmaxf(mx, qr.x);
                        mr = fmaxf(mr, qr.z);
                    }
                    float x = fmaxf(mx + mr * 2.0f, a.bx);
                    for (int t = 0; t < c; ++t) x = fmaxf(x + mr * 2.0f, a.bx);
                    px = x;
                    py = (py >= 0.0f) ? fminf(py, a.by) : fmaxf(py, -a.by);
                    publish(nxt);
                    g0y = py;               // human.set_goals([[goals[0][0], position[1]]])   :418
                    bvy = g0x; om = g0y;    // states[i,6:8] = goal  (reference writes cols 6:8) :421
                    for (int g = 0; g < a.G; ++g) { gi[2 * g] = g0x; gi[2 * g + 1] = g0y; } //   :422
                    gk = a.G; g1x = g0x; g1y = g0y;
                }
            } else if (__syncthreads_or(flag) != 0) {
                lds_flag[tid] = flag;
                lds_g0x[tid] = g0x;
                __syncthreads();
                if (valid && row == 0) {
                    float4* pvn = lds_p + nxt * TP + pbase;
                    for (int i = 0; i < n; ++i) {
                        if (!lds_flag[base + i]) continue;
                        float mx = pvn[0].x, mr = pvn[0].z;
                        for (int j = 1; j < n; ++j) {
                            mx = fmaxf(mx, pvn[j].x);
                            mr = fmaxf(mr, pvn[j].z);
                        }
                        if (robot_row) { // consider_robot: the robot where it stands in THIS substep
                            const float4 qr = lds_p[cur * TP + pbase + n];
                            mx = fmaxf(mx, qr.x);
                            mr = fmaxf(mr, qr.z);
                        }
                        float4 q = pvn[i];
                        q.x = fmaxf(mx + mr * 2.0f, a.bx);
                        q.y = (q.y >= 0.0f) ? fminf(q.y, a.by) : fmaxf(q.y, -a.by);
                        pvn[i] = q;
                        pvn[rows + i] = q;
                    }
                }
                __syncthreads();
                if (flag) {
                    const float4 q = lds_p[nxt * TP + pbase + row];
                    px = q.x; py = q.y;
                    g0y = py;               // human.set_goals([[goals[0][0], position[1]]])   :418
                    bvy = g0x; om = g0y;    // states[i,6:8] = goal  (reference writes cols 6:8) :421
                    for (int g = 0; g < a.G; ++g) { gi[2 * g] = g0x; gi[2 * g + 1] = g0y; } //   :422
                    gk = a.G; g1x = g0x; g1y = g0y;
                }
            }
        }
        STAMP(5);
        LDS_ORDER_FENCE(); // rows republished by the respawn rule are read by other lanes in the next substep
        cur = nxt;
    }
#ifdef CS_STAMPS
    if (a.stamps != nullptr && (threadIdx.x & 63) == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 12;
        for (int k = 0; k < 12; ++k) o[k] = st_acc[k];
    }
#endif

    if (a.trace != nullptr && (human || is_robot) && a.nsub > 0)
        write_trace(a.trace + (((long)(a.nsub - 1) * a.W + w) * rows + row) * 12, px, py, th, vx, vy, bvx, bvy, om, gx, gy, g0x, g0y);

    // ---- epilogue ---------------------------------------------------------------------------
    if (kmode & M_PEEK) {
        if (human) {
            float* o = a.peek_out + ((long)w * n + row) * 8;
            o[0] = px; o[1] = py; o[2] = th; o[3] = vx; o[4] = vy; o[5] = om;
            o[6] = g0x; o[7] = g0y; // human.goals[0] (the goals array), not the state's goal columns (:297)
        }
        return;
    }
    // cs_gym_step_staged: what the head decided for my world (gymhead.h GymFold).  A world that takes over its staged episode writes
    // nothing of the stepped rows; the wavefront copies the slot over it below.
    int fold_code = 0;
    if constexpr (FOLD) {
        if (a.gym.out != nullptr && a.gym.fold.on && valid)
            fold_code = __hip_atomic_load(a.gym.fold.pending + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool taken_over = fold_code >= 2;
    if (human && gdirty && (kmode & M_COMMIT_GOALS) && !taken_over) { gi[0] = g0x; gi[1] = g0y; gi[2] = g1x; gi[3] = g1y; }
    if (valid && !taken_over) {
        float* o = a.Sout + sidx * a.out_as;
        const long fs = a.out_fs;
        if (human || is_robot) {
            o[0] = px; o[fs] = py; o[2 * fs] = th; o[3 * fs] = vx; o[4 * fs] = vy; o[5 * fs] = bvx;
            o[6 * fs] = bvy; o[7 * fs] = om; o[10 * fs] = gx; o[11 * fs] = gy;
            if (a.Sout != a.Sin || is_robot) { o[8 * fs] = r; o[9 * fs] = m; o[12 * fs] = vd; }
        }
        // the Gym's observation of the stepped crowd (SocialNavGym.compute_humans_observable_state, social_nav_gym.py:100-105) straight
        // from the registers: what cs_gym_observe would read back from the rows just written
        if (a.obs != nullptr && human) {
            float* ob = a.obs + ((long)w * n + row) * a.obs_cols;
            ob[0] = px; ob[1] = py; ob[2] = vx; ob[3] = vy; ob[4] = r;
            if (a.obs_cols == 7) { ob[5] = th; ob[6] = om; }
        }
        if (is_robot && robot_moves && a.robot != nullptr) {
            float* rb = a.robot + (long)w * 13;
            rb[0] = px; rb[1] = py; rb[2] = th; rb[3] = vx; rb[4] = vy;
        }
        if constexpr (IMIT) {
            if (is_robot) {   // what cs_robot_model_step leaves: the robot's dynamic columns and its model's remembered desired force
                float* rb = a.robot + (long)w * 13;
                rb[0] = px; rb[1] = py; rb[2] = th; rb[3] = vx; rb[4] = vy; rb[5] = bvx; rb[6] = bvy; rb[7] = om;
                a.rm_memory[(long)w * 2] = rm_fdx; a.rm_memory[(long)w * 2 + 1] = rm_fdy;
            }
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
    if constexpr (FOLD) {
        if (a.gym.out != nullptr && a.gym.fold.on) {
            const GymFold& f = a.gym.fold;
            for (int l = 0; l < a.wpb; ++l) {                                   // the worlds of this wavefront, one after the other (wave-uniform)
                const int wl = blockIdx.x * a.wpb + l;
                if (wl >= a.W) break;
                const int code = __builtin_amdgcn_readlane(fold_code, l * rows);  // (the lane of the world's row 0 holds its code like every lane of the world)
                if (code == 0) continue;
                if (code >= 2) {
                    const long slot = code - 2;
                    const int status = csimpl::copy_world<true>(f.copy, slot, wl, tid, f.staged_status + slot);   // (a world that could not be generated is not copied)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every load of the slot has returned
                    if (tid == 0) {
                        f.failed[wl] = status != 0 ? 1 : 0;
                        __hip_atomic_store(f.pending + wl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // LAST: from here on the refill may overwrite the slot (it now belongs to episode epoch + depth)
                        __hip_atomic_store(f.epoch + wl, f.epoch[wl] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else if (tid == 0) {
                    f.failed[wl] = 2;                                             // deferred: the slot was not staged yet (pending[wl] stays 1)
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// build selection (host side)
// ------------------------------------------------------------------------------------------
using kfn = void (*)(const KArgs);

// Which instantiation of k_sfm_step a launch runs (crowdstep.hip select_variant decides; cs_step_variant reports it)
struct Variant { int maxt, occ, rows_ct, lean; bool peq; };

template <int MAXT, int OCC, int ROWS_CT, int LEAN>
kfn pick_kernel(int type, bool peq)
{
#define CS_CASE(SOC, HD)                                                                                      \
    if constexpr (LEAN != 0) return (kfn)k_sfm_step<SOC, HD, true, MAXT, OCC, ROWS_CT, LEAN>;                      \
    else return peq ? (kfn)k_sfm_step<SOC, HD, true, MAXT, OCC, ROWS_CT, 0>                               \
                    : (kfn)k_sfm_step<SOC, HD, false, MAXT, OCC, ROWS_CT, 0>;
    switch (type) {
        case 0: CS_CASE(0, 0) case 1: CS_CASE(1, 0) case 2: CS_CASE(2, 0)
        case 3: CS_CASE(0, 1) case 4: CS_CASE(1, 1) case 5: CS_CASE(2, 1)
        case 6: CS_CASE(0, 2) case 7: CS_CASE(1, 2) case 8: CS_CASE(2, 2)
    }
#undef CS_CASE
    return nullptr;
}

// one lookup per translation unit of builds: the kernel of variant `v` for model `type`, or nullptr when this unit does not hold it
#define CS_V(MT, OC, RC, LN) if (v.maxt == MT && v.occ == OC && v.rows_ct == RC && v.lean == LN) return pick_kernel<MT, OC, RC, LN>(type, v.peq);
kfn sfm_builds_generic(const Variant& v, int type);   // sfmstep_generic.hip: one world per block
kfn sfm_builds_leanrt(const Variant& v, int type);   // sfmstep_leanrt.hip: the lean builds with a run-time row count
kfn sfm_builds_lean25(const Variant& v, int type);   // sfmstep_lean25.hip: 25 rows per world, plain crowd batch
kfn sfm_builds_lean30(const Variant& v, int type);   // sfmstep_lean30.hip: 30 rows per world, plain crowd batch
kfn sfm_builds_small(const Variant& v, int type);   // sfmstep_small.hip: 10 and 20 rows per world, plain crowd batch
kfn sfm_builds_lean50(const Variant& v, int type);   // sfmstep_lean50.hip: 50 rows per world without / with walls
kfn sfm_builds_robot26(const Variant& v, int type);   // sfmstep_robot26.hip: 25 humans + a visible robot
kfn sfm_builds_robotx(const Variant& v, int type);   // sfmstep_robotx.hip: 5 / 10 / 50 humans + a visible robot
kfn sfm_builds_imit(const Variant& v, int type);     // sfmstep_imit.hip: a visible robot under its own human motion model (imitation learning)
kfn sfm_builds_peragent(const Variant& v, int type); // sfmstep_peragent.hip: 25 rows per world, per-agent parameters (Helbing / Guo laws)

} // namespace cstep
