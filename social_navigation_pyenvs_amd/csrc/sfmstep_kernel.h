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
constexpr int WG_WAVES_MAX = 4;   // (launch bound of the one-wavefront builds: four independent wavefronts per workgroup; a bound of eight -- which measures the same as
                                  //  four -- makes the compiler allocate the Moussaid builds ten registers fewer and their launches 2 us longer)
template <int SOC, int HEADED, bool PEQ, int MAXT, int OCC, int ROWS_CT, int LEAN>
// One-wavefront builds (MAXT = 64) are launched as workgroups of a.wg_waves INDEPENDENT wavefronts (four by default), each the "block" the
// rest of this file talks about -- its own worlds, its own slice of the dynamic LDS, no barrier with the others.  Measured (tools/ab_wg_waves.sh):
// cfg3 29.3 -> 28.2 us per launch, a launch of one substep 7.3 -> 6.1 us, with 512 workgroups of four instead of 2048 of one; inside an XCD the
// wavefronts then start within 0.04 us instead of 0.24 (wall-clock stamps of the diagnostic build), the rest of the gain is per-workgroup work of
// the dispatcher no wavefront sees.  (The 2 - 5 us between the first and the last wavefront's start that the same stamps show are offsets between
// the eight XCDs, whatever the workgroup size: HISTORY.md.)
__global__ __launch_bounds__(MAXT == 64 ? 64 * WG_WAVES_MAX : MAXT, OCC) void k_sfm_step(const KArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw0[];
    const int wave_in_wg = MAXT == 64 ? (int)(threadIdx.x >> 6) : 0;
    unsigned char* smem_raw = smem_raw0 + (MAXT == 64 ? (size_t)wave_in_wg * a.lds_per_wave : 0);
    const int vblock = MAXT == 64 ? (int)blockIdx.x * a.wg_waves + wave_in_wg : (int)blockIdx.x;   // the one-wavefront block this wavefront is
#ifdef CS_STAMPS
    unsigned long long pst_first, pst_first_rt;   // the wavefront's first instruction (shader clock, and the 100 MHz wall clock all wavefronts share)
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(pst_first), "=s"(pst_first_rt)::"memory");
#endif
    // all_params_equal, whole worlds inside one wavefront: every unordered pair is evaluated ONCE (as the reference
    // does, forces_parallel.py:100-131: F[i,j] = f, F[j,i] = -f) and the reaction handed over through LDS.
    // Per-agent parameters (forces_parallel.py:43-84, :261: F_i = sum_j f(P_i; i, j), no antisymmetry) take the same loop for the Helbing /
    // Guo laws (PP): the lane that visits the pair (i, j) evaluates BOTH directions -- d, 1/d and the overlap are shared, each side's
    // A e^{./B} (+ C e^{./D}) comes from its own parameters, the partner's read from an LDS row beside its position -- 12 VALU + 3
    // transcendentals per unordered pair instead of 2 x (12 + 2) (the peragent bench entry: 443 -> 35x VALU per wavefront-substep).
    constexpr bool N3L = MAXT == 64 && (PEQ || SOC != 2);
    constexpr bool PP = N3L && !PEQ;
    const int T = MAXT == 64 ? 64 : blockDim.x;
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
    constexpr int WP_NW = 4;                                        // segments of a pair's polygon per trip of its closest-point loop
    int* lds_wrec = reinterpret_cast<int*>(lds_poly + (a.Smax > 0 ? a.seg_tab / a.Smax : 0));   // [WP_MAX] pair: agent's row in lds_p | first segment << 8 | agent lane << 20
    float2* lds_wres = reinterpret_cast<float2*>(lds_wrec + WP_MAX);                             // [WP_MAX + 1] the pairs' forces ([WP_MAX]: the zero of "no pair")
    float4* lds_wlaw = reinterpret_cast<float4*>(lds_wres + WP_MAX + 2);                         // [1] the wall law A, log2 e / B, k1, k2 (all_params_equal)
    float2* lds_wrs = lds_vr;                                                                    // [WP_MAX] pair: the agent's radius, safety space
    float2* lds_wcv = reinterpret_cast<float2*>(lds_g0x);                                        // [T] the agents' refreshed linear velocities (contact terms)
    const int tid = MAXT == 64 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    static_assert(!LEAN || (PEQ && MAXT == 64), "the lean build is a pair-once build");
    constexpr bool LEAN_ROBOT = LEAN == 3 || LEAN == 4 || LEAN == 5;   // the robot is the last row (5: with walls)
    constexpr bool IMIT = LEAN == 4;                       // ... and follows its own human motion model
    constexpr bool NO_WALLS = LEAN == 1 || LEAN == 3 || LEAN == 4;     // wall code compiled out
    const int rows = ROWS_CT > 0 ? ROWS_CT : a.rows;
    const int n = LEAN_ROBOT ? rows - 1 : (LEAN ? rows : a.n);
    const int kmode = LEAN_ROBOT ? ((int)M_COMMIT_GOALS | (a.mode & (int)M_ROBOT_FROM_ARRAY)) : (LEAN ? (int)M_COMMIT_GOALS : a.mode);
    const int lw = tid / rows;
    const int row = tid - lw * rows;
    const int w = vblock * a.wpb + lw;
    const bool valid = (lw < a.wpb) && (w < a.W);
    const bool robot_row = LEAN_ROBOT ? true : (LEAN ? false : (a.flags & CS_ROBOT_ROW) != 0);
    const bool human = valid && row < n;
    const bool is_robot = valid && robot_row && row == n;
    const int base = lw * rows;        // first row of my world in the per-row arrays (lds_v, lds_vr, ...)
    const int pbase = lw * a.ws;       // first row of my world in the doubled position buffers
    const float dt = a.dt;
    const int obs_type = (a.type == 1 || a.type == 4 || a.type == 7) ? 1 : 0;

    // the load phase: every global load of the prologue issued before the first use (one memory round trip); the row, the parameter rows, the goal list, the world flag, the action, the Gym head's inputs
#include "sfmstep_load.inc"
    // the prologue: per-launch constants, wall segments and polygon circles staged in LDS, the Gym step's head (cs_gym_step), substep-0 rows published, the (agent, polygon) pairs of the wall pass numbered (LEAN = 2)
#include "sfmstep_prologue.inc"
    const bool prio_young = MAXT == 64 && vblock >= a.young_from;
    // base priority 1: above the generator's wavefronts of a refill pass on the side stream (priority 0, and OLDER than any of mine, so a tie
    // would go to them): the Gym step in NEXT_STEP mode 43.3 -> 41.5 us, a plain launch unchanged
    if constexpr (MAXT == 64) __builtin_amdgcn_s_setprio(1);
    for (int sub = 0; sub < a.nsub; ++sub) {
        // head of a substep: priority turn, recorders (imitation snapshot, cs_step_trace), the robot under its own motion model (LEAN = 4), partner-row fetch helpers, the goal switch
#include "sfmstep_sub_head.inc"
        // part A: what does not depend on this substep's social force -- the wall pairs' pass, refreshed velocity, desired force, the all-lanes wall pass, heading and torque
#include "sfmstep_sub_part_a.inc"
        // the pair-once loop (each unordered pair evaluated once, reaction handed over through LDS), the reaction sum, and the contact pass behind a wave vote
#include "sfmstep_sub_pairloop.inc"
        // all partners per lane (per-agent parameters on Moussaid, worlds of more than one wavefront), part B: total force, body frame, torque, the Euler step, the rows published for the next substep
#include "sfmstep_sub_tail.inc"
        // the parallel-traffic respawn rule, sequential inside a world, by wave ballot
#include "sfmstep_sub_respawn.inc"
        STAMP(5);
        LDS_ORDER_FENCE(); // rows republished by the respawn rule are read by other lanes in the next substep
        cur = nxt;
    }
#ifdef CS_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pst_loop_end)::"memory");
#endif

    if (a.trace != nullptr && (human || is_robot) && a.nsub > 0)
        write_trace(a.trace + (((long)(a.nsub - 1) * a.W + w) * rows + row) * 12, px, py, th, vx, vy, bvx, bvy, om, gx, gy, g0x, g0y);

    // the epilogue: the row written back once per launch, goal lists, observation rows, robot rows, and the take-over of a staged episode (cs_gym_step_staged)
#include "sfmstep_epilogue.inc"
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
