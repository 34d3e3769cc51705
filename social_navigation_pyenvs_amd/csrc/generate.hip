// generate.hip -- device-side scenario generators: reset() of W worlds without W serial host
// rejection-sampling loops.
//
// Restates, one GPU lane per world, what SocialNavGym.reset does per world on the host:
//   np.random.seed(offset[phase] + case)                         /root/reference/social_gym/social_nav_gym.py:135-137
//   hybrid_scenario: np.random.choice([...]) then re-seed         social_nav_gym.py:156-157, 185-186
//   generate_circular_crossing_setting                            /root/reference/social_gym/social_nav_sim.py:200-299
//   generate_parallel_traffic_scenario                            social_nav_sim.py:301-362
//   generate_circular_crossing_with_static_obstacles              social_nav_sim.py:364-431
//   HumanAgent rows built by reset_sim                            social_nav_sim.py:97-198 (mass 75, zero velocities)
// The reference draws from numpy's legacy global stream: MT19937 seeded by init_genrand (integer seed),
// random_sample() = (a >> 5, b >> 6) 53-bit doubles, uniform(lo, hi) = lo + (hi - lo) * random_sample(),
// choice of two = next_uint32 & 1.  All of it is restated here in integer / f64 arithmetic, so the draw ORDER and the
// accept / reject decisions are the reference's; rows are rounded to f32 only when they are stored.
//
// MT19937 state: 624 words per world, kept in a caller-provided scratch [624][W] (word-major: the seeding and twist
// loops of the 64 lanes of a wavefront touch consecutive addresses).
//
// gfx950 only; not a hot path (one launch per reset), written for clarity.

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>

#include "common.h"
#include "crowdstep.h"
#include "gymhead.h"
#include "worldcopy.h"

#pragma clang fp contract(off) // the reference's numpy expressions are not fused

namespace {

using csimpl::fail;

constexpr int GEN_MAXN = 128; // humans per world the per-lane placement arrays hold

struct GArgs {
    cs_generator g;
    int W, rows, G, robot_row;
    int orca;     // the batch is an ORCA crowd: columns 5:7 carry RVO2's preferred velocity (orca_pref below)
    float* S;
    long as, fs;
    float* goals;
    float* robot;
    int* world_flags;
    const uint32_t* seeds;
    const int32_t* mask;
    int32_t* status;
    int32_t* scenario_out;
    uint32_t* mt; // [624][W]
};

struct MT {
    uint32_t* s;
    long stride;
    int pos;
    __device__ uint32_t& at(int k) { return s[(long)k * stride]; }
};

// numpy/random/src/mt19937/mt19937.c: mt19937_seed (init_genrand)
__device__ void mt_seed(MT& m, uint32_t seed)
{
    uint32_t prev = seed;
    m.at(0) = prev;
    for (int i = 1; i < 624; ++i) {
        prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
        m.at(i) = prev;
    }
    m.pos = 624;
}

__device__ void mt_twist(MT& m)
{
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    int kk = 0;
    for (; kk < 624 - 397; ++kk) {
        const uint32_t y = (m.at(kk) & UPPER) | (m.at(kk + 1) & LOWER);
        m.at(kk) = m.at(kk + 397) ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    }
    for (; kk < 623; ++kk) {
        const uint32_t y = (m.at(kk) & UPPER) | (m.at(kk + 1) & LOWER);
        m.at(kk) = m.at(kk - 227) ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    }
    const uint32_t y = (m.at(623) & UPPER) | (m.at(0) & LOWER);
    m.at(623) = m.at(396) ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    m.pos = 0;
}

__device__ uint32_t mt_next(MT& m)
{
    if (m.pos >= 624) mt_twist(m);
    uint32_t y = m.at(m.pos++);
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// legacy random_sample(): 53-bit double in [0, 1)
__device__ double rnd(MT& m)
{
    const uint32_t a = mt_next(m) >> 5, b = mt_next(m) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

__device__ double uniform(MT& m, double lo, double hi) { return lo + (hi - lo) * rnd(m); }

// social_gym/src/utils.py:7-13 (Python's sign-of-divisor modulo == fmod for the operand signs of each branch)
__device__ double bound_angle_d(double a)
{
    const double two_pi = 2.0 * 3.141592653589793, pi = 3.141592653589793;
    if (a >= two_pi) a = fmod(a, two_pi);
    if (a <= -two_pi) a = fmod(a, two_pi);
    if (a > pi) a -= two_pi;
    if (a < -pi) a += two_pi;
    return a;
}

__device__ double norm2d(double x, double y) { return sqrt(x * x + y * y); } // np.linalg.norm of a 2-vector

// An ORCA crowd keeps RVO2's preferred velocity in columns 5:7 of its rows (set_state_orca / update_goals_orca,
// motion_model_manager.py:105-133: (g - p) / |g - p| if |g - p| > desired_speed else (g - p)), set when the simulator is built -- so
// a world generated on the device must carry it from its first substep on.  float32 on the stored (rounded) columns, as the host
// form of the same reset computes it (BatchedSocialNavGym._reset_on_device).
__device__ __forceinline__ void orca_pref(float* s, long fs, float px, float py, float gx, float gy, float vd)
{
    const float dx = gx - px, dy = gy - py;
    const float dn = sqrtf(dx * dx + dy * dy);
    const float dd = dn > 1e-30f ? dn : 1e-30f;
    s[5 * fs] = dn > vd ? dx / dd : dx;
    s[6 * fs] = dn > vd ? dy / dd : dy;
}

__global__ __launch_bounds__(64) void k_generate(const GArgs a)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.W) return;
    if (a.mask != nullptr && a.mask[w] == 0) return;
    const cs_generator& g = a.g;
    const int n = g.n;
    const double pi = 3.141592653589793;
    MT m{a.mt + w, (long)a.W, 624};
    const uint32_t seed = a.seeds[w];
    mt_seed(m, seed);
    int scenario = g.scenario;
    if (scenario == CS_SCN_HYBRID) { // np.random.choice(['circle_crossing', 'parallel_traffic']); np.random.seed(seed)
        scenario = (mt_next(m) & 1u) ? CS_SCN_PARALLEL_TRAFFIC : CS_SCN_CIRCULAR_CROSSING;
        mt_seed(m, seed);
    }
    double px[GEN_MAXN], py[GEN_MAXN], yaw[GEN_MAXN], rad[GEN_MAXN], spd[GEN_MAXN];
    int status = 0;
    const double R = g.circle_radius, L = g.traffic_length, H = g.traffic_height, rr = g.robot_radius;
    double rpx = 0, rpy = 0, ryaw = 0, rgx = 0, rgy = 0;
    const bool insert_robot = g.insert_robot != 0;

    if (scenario == CS_SCN_CIRCULAR_CROSSING || scenario == CS_SCN_PARALLEL_TRAFFIC) {
        for (int i = 0; i < n; ++i) { // one speed draw then one radius draw per human (:217-220)
            if (g.randomize_attributes) { spd[i] = uniform(m, 0.5, 1.5); rad[i] = uniform(m, 0.3, 0.5); }
            else { spd[i] = 1.0; rad[i] = 0.3; }
        }
    }
    if (scenario == CS_SCN_CIRCULAR_CROSSING) {
        rpx = 0.0; rpy = 0.0 - R; ryaw = pi / 2.0; rgx = 0.0; rgy = 0.0 + R;
        if (!g.randomize_positions) { // evenly spaced (:231-251)
            const int slots = n + (insert_robot ? 1 : 0);
            const double step = (2.0 * pi) / slots;
            for (int i = 0; i < n; ++i) {
                const int k = insert_robot ? i + 1 : i;
                const double off = insert_robot ? -(pi / 2.0) : 0.0;
                px[i] = 0.0 + R * cos(off + step * k); py[i] = 0.0 + R * sin(off + step * k);
                yaw[i] = insert_robot ? bound_angle_d((pi / 2.0) + step * k) : bound_angle_d(-pi + step * k);
            }
        } else {
            for (int i = 0; i < n && status == 0; ++i) {
                bool placed = false;
                for (int t = 0; t < g.max_tries; ++t) {
                    const double angle = rnd(m) * pi * 2.0;
                    const double nx = (rnd(m) - 0.5) * spd[i];
                    const double ny = (rnd(m) - 0.5) * spd[i];
                    const double x = 0.0 + R * cos(angle) + nx, y = 0.0 + R * sin(angle) + ny;
                    bool collide = false;
                    for (int j = 0; j < i; ++j) {
                        const double md = rad[i] + rad[j] + 0.2;
                        // goal of a placed human: (-x + 2 cx, -y + 2 cx)  (:280 uses the centre's x twice; cx = cy = 0)
                        if (norm2d(x - px[j], y - py[j]) < md || norm2d(x - (-px[j] + 0.0), y - (-py[j] + 0.0)) < md) {
                            collide = true;
                            break;
                        }
                    }
                    if (insert_robot && (norm2d(x - rpx, y - rpy) < rad[i] + rr + 0.2 || norm2d(x - rgx, y - rgy) < rad[i] + rr + 0.2))
                        collide = true;
                    if (!collide) {
                        px[i] = x; py[i] = y; yaw[i] = bound_angle_d(pi + angle);
                        placed = true;
                        break;
                    }
                }
                if (!placed) status = 1;
            }
        }
    } else if (scenario == CS_SCN_PARALLEL_TRAFFIC) {
        rpx = -(L / 2.0) + 1.0; rpy = 0.0; ryaw = 0.0; rgx = (L / 2.0) - 1.0; rgy = 0.0;
        double area = 0.0;
        for (int i = 0; i < n; ++i) area += pi * (rad[i] * rad[i]);
        if (area > L * H * 0.4) status = 2; // ValueError in the reference (:318-319)
        for (int i = 0; i < n && status == 0; ++i) {
            bool placed = false;
            for (int t = 0; t < g.max_tries; ++t) {
                const double lo = -(L / 2.0) + rad[i], hi = L / 2.0 - rad[i];
                const double x = (hi - lo) * rnd(m) + lo;
                const double y = (rnd(m) - 0.5) * H;
                bool collide = false;
                for (int j = 0; j < i; ++j)
                    if (norm2d(x - px[j], y - py[j]) - rad[i] - rad[j] - 0.1 < 0.0) { collide = true; break; }
                if (insert_robot && norm2d(x - rpx, y - rpy) - rad[i] - rr - 0.1 < 0.0) collide = true;
                if (!collide) {
                    px[i] = x; py[i] = y; yaw[i] = bound_angle_d(-pi);
                    placed = true;
                    break;
                }
            }
            if (!placed) status = 1;
        }
    } else { // circular crossing with static obstacles
        rpx = 0.0; rpy = 0.0 - R; ryaw = pi / 2.0; rgx = 0.0; rgy = 0.0 + R;
        const double inner = R - 3.0;
        for (int i = 0; i < n; ++i) { // the three "obstacles" are immobile humans with a drawn radius (:381-387)
            if (i < 3) { spd[i] = 0.0; rad[i] = 1.0 + (rnd(m) - 1.0) * 0.4; }
            else { spd[i] = 1.0; rad[i] = 0.3; }
        }
        const double sector = pi / (double)(n / 2);
        for (int i = 0; i < n && status == 0; ++i) {
            bool placed = false;
            for (int t = 0; t < g.max_tries; ++t) {
                double angle, nx, ny, ring;
                if (i < 3) {
                    angle = sector * (-0.5 + 2.0 * i + (rnd(m) - 0.5) * 0.5);
                    nx = (rnd(m) - 0.5) * 0.1; ny = (rnd(m) - 0.5) * 0.1;
                    ring = inner;
                } else {
                    angle = sector * (0.5 + 2.0 * i + (rnd(m) - 0.5) * 0.5);
                    nx = (rnd(m) - 0.5) * 0.7; ny = (rnd(m) - 0.5) * 0.7;
                    ring = R;
                }
                const double x = 0.0 + ring * cos(angle) + nx, y = 0.0 + ring * sin(angle) + ny;
                bool collide = false;
                for (int j = 0; j < i; ++j) {
                    const double md = rad[i] + rad[j] + 0.2;
                    const double gxj = (j < 3) ? px[j] : (-px[j] + 0.0), gyj = (j < 3) ? py[j] : (-py[j] + 0.0);
                    if (norm2d(x - px[j], y - py[j]) < md || norm2d(x - gxj, y - gyj) < md) { collide = true; break; }
                }
                if (norm2d(x - rpx, y - rpy) < rad[i] + rr + 0.2 || norm2d(x - rgx, y - rgy) < rad[i] + rr + 0.2) collide = true;
                if (!collide) {
                    px[i] = x; py[i] = y; yaw[i] = bound_angle_d(pi + angle);
                    placed = true;
                    break;
                }
            }
            if (!placed) status = 1;
        }
    }
    if (a.status != nullptr) a.status[w] = status;
    if (a.scenario_out != nullptr) a.scenario_out[w] = scenario;
    if (status != 0) return; // the rows of a world that could not be generated are left untouched

    // ---- rows: [px,py,theta,vx,vy,bvx,bvy,omega,r,m,gx,gy,vd]  (agent.py:256-258), goals [n][G][2] NaN-padded
    const float nanf_ = __builtin_nanf("");
    for (int i = 0; i < n; ++i) {
        double g0x, g0y, g1x, g1y;
        int ng;
        if (scenario == CS_SCN_PARALLEL_TRAFFIC) { g0x = -(L / 2.0) - 3.0; g0y = py[i]; g1x = g1y = 0; ng = 1; }
        else if (scenario == CS_SCN_CIRCULAR_CROSSING_STATIC_OBSTACLES && i < 3) { g0x = px[i]; g0y = py[i]; g1x = px[i]; g1y = py[i]; ng = 2; }
        else { g0x = 0.0 * 2.0 - px[i]; g0y = 0.0 * 2.0 - py[i]; g1x = px[i]; g1y = py[i]; ng = 2; }
        float* s = a.S + ((long)w * a.rows + i) * a.as;
        const long fs = a.fs;
        s[0] = (float)px[i]; s[fs] = (float)py[i]; s[2 * fs] = (float)yaw[i];
        for (int c = 3; c < 8; ++c) s[c * fs] = 0.0f;
        s[8 * fs] = (float)rad[i]; s[9 * fs] = (float)g.human_mass; s[10 * fs] = (float)g0x; s[11 * fs] = (float)g0y;
        s[12 * fs] = (float)spd[i];
        if (a.orca) orca_pref(s, fs, (float)px[i], (float)py[i], (float)g0x, (float)g0y, (float)spd[i]);
        float* gl = a.goals + ((long)w * n + i) * a.G * 2;
        for (int k = 0; k < a.G; ++k) {
            float gx = nanf_, gy = nanf_;
            if (k == 0) { gx = (float)g0x; gy = (float)g0y; }
            else if (k == 1 && ng == 2) { gx = (float)g1x; gy = (float)g1y; }
            gl[2 * k] = gx; gl[2 * k + 1] = gy;
        }
    }
    // robot safe-state row (social_nav_gym.py:211-213: robot.set(px, py, gx, gy, 0, 0, yaw, w=0))
    auto robot_row = [&](float* o, long fs) {
        o[0] = (float)rpx; o[fs] = (float)rpy; o[2 * fs] = (float)ryaw;
        for (int c = 3; c < 8; ++c) o[c * fs] = 0.0f;
        o[8 * fs] = (float)rr; o[9 * fs] = (float)g.robot_mass; o[10 * fs] = (float)rgx; o[11 * fs] = (float)rgy;
        o[12 * fs] = (float)g.robot_desired_speed;
    };
    if (a.robot != nullptr) robot_row(a.robot + (long)w * 13, 1);
    if (a.robot_row) robot_row(a.S + ((long)w * a.rows + n) * a.as, a.fs);
    if (a.world_flags != nullptr) a.world_flags[w] = (scenario == CS_SCN_PARALLEL_TRAFFIC) ? 1 : 0; // respawn rule on (:359-360)
}

// ------------------------------------------------------------------------------------------
// One WAVEFRONT per world (n <= 64): the latency form, used for the masked auto-reset inside a device-resident loop.
// MT19937 state in LDS (the current block and its successor, twisted cooperatively); attribute draws are read as wave-wide
// broadcasts.  The rejection loops of the circular crossing and of the parallel traffic are SPECULATIVE: eight attempts of
// the human being placed are evaluated at once, each from its own words of the stream by eight lanes that share the test
// against the placed humans (kept in LDS); the first accepted attempt wins and the stream position moves to just behind
// it, so a human mostly costs one round whatever its rejection count -- the slowest world of a batch no longer sets the
// latency of a masked reset.
// Same draws, same f64 expressions, same accept / reject decisions as k_generate above.
// ------------------------------------------------------------------------------------------
struct WMT {
    uint32_t* cur; // [624] in LDS: the block the stream is in
    uint32_t* nxt; // [624] in LDS: twist(cur), kept ready so that speculative attempts can read across the block boundary
    int pos;       // wave-uniform index into cur; 624 <= pos < 1248 means "in nxt" until the next advance
};

// nw = twist(old) (genrand's block update, out of place), cooperatively: chunks of 64 in increasing order -- element kk reads
// old[kk], old[kk + 1] and either old[kk + 397] (kk < 227) or nw[kk - 227], written at least three chunks earlier
__device__ void wmt_twist_from(const uint32_t* old, uint32_t* nw, int lane)
{
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    for (int k0 = 0; k0 < 623; k0 += 64) {
        const int kk = k0 + lane;
        if (kk < 623) {
            const uint32_t y = (old[kk] & UPPER) | (old[kk + 1] & LOWER);
            nw[kk] = (kk < 227 ? old[kk + 397] : nw[kk - 227]) ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        __syncthreads();
    }
    if (lane == 0) {
        const uint32_t y = (old[623] & UPPER) | (nw[0] & LOWER);
        nw[623] = nw[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    }
    __syncthreads();
}

__device__ void wmt_seed(WMT& m, uint32_t seed, int lane)
{
    __syncthreads();
    if (lane == 0) { // the init_genrand recurrence is serial
        uint32_t prev = seed;
        m.cur[0] = prev;
        for (int i = 1; i < 624; ++i) {
            prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
            m.cur[i] = prev;
        }
    }
    __syncthreads();
    wmt_twist_from(m.cur, m.nxt, lane);
    m.pos = 624;   // numpy leaves the seeded block exhausted: the first draw comes from its twist
}

__device__ void wmt_advance(WMT& m, int lane)   // the stream has left `cur`: make `nxt` current and prepare the block after it
{
    uint32_t* t = m.cur; m.cur = m.nxt; m.nxt = t;
    m.pos -= 624;
    __syncthreads();
    wmt_twist_from(m.cur, m.nxt, lane);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__device__ uint32_t wmt_next(WMT& m, int lane)
{
    if (m.pos >= 624) wmt_advance(m, lane);
    return mt_temper(m.cur[m.pos++]);
}

__device__ double wrnd(WMT& m, int lane)
{
    const uint32_t a = wmt_next(m, lane) >> 5, b = wmt_next(m, lane) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

// random_sample() number q of the stream counted from word index k0 (k0 < 624, read-ahead below 1248), without consuming it
__device__ __forceinline__ double wmt_peek_double(const WMT& m, int k0, int q)
{
    const int k = k0 + 2 * q;
    const uint32_t wa = (k < 624) ? m.cur[k] : m.nxt[k - 624];
    const uint32_t wb = (k + 1 < 624) ? m.cur[k + 1] : m.nxt[k + 1 - 624];
    const uint32_t a = mt_temper(wa) >> 5, b = mt_temper(wb) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

// np.linalg.norm((dx, dy)) < md, decided on the squares; the square root only inside the rounding band of the comparison
__device__ __forceinline__ bool closer_than(double dx, double dy, double md)
{
    const double d2 = dx * dx + dy * dy, md2 = md * md;
    bool c = d2 < md2;
    if (fabs(d2 - md2) <= 1e-14 * md2) c = sqrt(d2) < md;
    return c;
}

// where a generated world goes (the batch being reset, a staging batch, or the live batch of cs_consume_staged_worlds)
struct GenOut {
    float* S; long as, fs; float* goals; float* robot; int* world_flags; int rows, G, robot_row, orca;
};

__device__ __forceinline__ GenOut gen_out(const GArgs& a)
{
    return GenOut{a.S, a.as, a.fs, a.goals, a.robot, a.world_flags, a.rows, a.G, a.robot_row, a.orca};
}

// One world by one wavefront (the block): the reference's generator for `seed`, rows written to `a` when it succeeds.  Returns the
// status (0 ok, 1 rejection sampling gave up, 2 traffic too dense); `scenario_out` = the scenario drawn (hybrid batches).
__device__ int generate_world_wave(const cs_generator& g, const GenOut& a, int w, int lane, uint32_t seed, int& scenario_out)
{
    __shared__ uint32_t s_mt[2][624];
    __shared__ double s_rad[64], s_spd[64], s_px[64], s_py[64], s_yaw[64];
    const int n = g.n;
    const double pi = 3.141592653589793;
    WMT m;
    m.cur = s_mt[0];
    m.nxt = s_mt[1];
    m.pos = 624;
    wmt_seed(m, seed, lane);
    int scenario = g.scenario;
    if (scenario == CS_SCN_HYBRID) {
        scenario = (wmt_next(m, lane) & 1u) ? CS_SCN_PARALLEL_TRAFFIC : CS_SCN_CIRCULAR_CROSSING;
        // np.random.seed(seed) again (:156-157): the stream restarts at its first word, which is word 0 of the block the
        // choice was just drawn from -- rewinding the position is the whole re-seeding
        m.pos = 0;
    }
    const double R = g.circle_radius, L = g.traffic_length, H = g.traffic_height, rr = g.robot_radius;
    const bool insert_robot = g.insert_robot != 0;
    double rpx = 0, rpy = 0, ryaw = 0, rgx = 0, rgy = 0;
    double mx = 0, my = 0, myaw = 0; // placed human `lane`
    int status = 0;

    // attributes: every lane sees every draw; slot i keeps its own in LDS for the uniform reads below
    if (scenario == CS_SCN_CIRCULAR_CROSSING || scenario == CS_SCN_PARALLEL_TRAFFIC) {
        for (int i = 0; i < n; ++i) {
            double sp = 1.0, ra = 0.3;
            if (g.randomize_attributes) { sp = 0.5 + (1.5 - 0.5) * wrnd(m, lane); ra = 0.3 + (0.5 - 0.3) * wrnd(m, lane); }
            if (lane == i) { s_spd[i] = sp; s_rad[i] = ra; }
        }
    } else {
        for (int i = 0; i < n; ++i) {
            double sp = 1.0, ra = 0.3;
            if (i < 3) { sp = 0.0; ra = 1.0 + (wrnd(m, lane) - 1.0) * 0.4; }
            if (lane == i) { s_spd[i] = sp; s_rad[i] = ra; }
        }
    }
    __syncthreads();
    const double myrad = lane < n ? s_rad[lane] : 0.0;

    if (scenario == CS_SCN_CIRCULAR_CROSSING) {
        rpx = 0.0; rpy = 0.0 - R; ryaw = pi / 2.0; rgx = 0.0; rgy = 0.0 + R;
        if (!g.randomize_positions) {
            const int slots = n + (insert_robot ? 1 : 0);
            const double step = (2.0 * pi) / slots;
            const int k = insert_robot ? lane + 1 : lane;
            const double off = insert_robot ? -(pi / 2.0) : 0.0;
            mx = 0.0 + R * cos(off + step * k); my = 0.0 + R * sin(off + step * k);
            myaw = insert_robot ? bound_angle_d((pi / 2.0) + step * k) : bound_angle_d(-pi + step * k);
        } else {
            // speculative rejection sampling: the wavefront evaluates 8 attempts of human i at once, attempt a = lane / 8 from
            // its own six words of the stream (read ahead without consuming them), its 8 lanes sharing the test against the
            // placed humans (lane % 8 takes every eighth one).  The first attempt none of whose lanes found a collision wins
            // and the stream moves to just behind it: the same draws and decisions as the sequential loop.
            const int att = lane >> 3, sub = lane & 7;
            for (int i = 0; i < n && status == 0; ++i) {
                const double ri = s_rad[i], si = s_spd[i];
                bool placed = false;
                for (int t0 = 0; t0 < g.max_tries && !placed; t0 += 8) {
                    if (m.pos >= 624) wmt_advance(m, lane);
                    const int base = m.pos, k0 = base + 6 * att;
                    const double angle = wmt_peek_double(m, k0, 0) * pi * 2.0;
                    const double nx = (wmt_peek_double(m, k0, 1) - 0.5) * si;
                    const double ny = (wmt_peek_double(m, k0, 2) - 0.5) * si;
                    double sa, ca;
                    sincos(angle, &sa, &ca); // one range reduction for both (same values as cos() and sin(): G6 parity)
                    const double x = 0.0 + R * ca + nx, y = 0.0 + R * sa + ny;
                    bool collide = t0 + att >= g.max_tries;
                    for (int j = sub; j < i; j += 8) {
                        const double qx = s_px[j], qy = s_py[j], md = ri + s_rad[j] + 0.2;
                        collide |= closer_than(x - qx, y - qy, md) || closer_than(x - (-qx + 0.0), y - (-qy + 0.0), md);
                    }
                    if (insert_robot && sub == 0)
                        collide |= closer_than(x - rpx, y - rpy, ri + rr + 0.2) || closer_than(x - rgx, y - rgy, ri + rr + 0.2);
                    const unsigned long long bad = __builtin_amdgcn_ballot_w64(collide);
                    int win = -1;
                    for (int q = 7; q >= 0; --q) if (((bad >> (8 * q)) & 0xFFull) == 0) win = q;
                    if (win >= 0) {
                        if (lane == 8 * win) { s_px[i] = x; s_py[i] = y; s_yaw[i] = bound_angle_d(pi + angle); }
                        m.pos = base + 6 * (win + 1);
                        placed = true;
                    } else {
                        m.pos = base + 6 * 8;
                    }
                    __syncthreads();
                }
                if (!placed) status = 1;
            }
            if (lane < n) { mx = s_px[lane]; my = s_py[lane]; myaw = s_yaw[lane]; }
        }
    } else if (scenario == CS_SCN_PARALLEL_TRAFFIC) {
        rpx = -(L / 2.0) + 1.0; rpy = 0.0; ryaw = 0.0; rgx = (L / 2.0) - 1.0; rgy = 0.0;
        double area = 0.0;
        for (int i = 0; i < n; ++i) area += pi * (s_rad[i] * s_rad[i]);
        if (area > L * H * 0.4) status = 2;
        const int att = lane >> 3, sub = lane & 7;
        for (int i = 0; i < n && status == 0; ++i) {   // speculative, four words per attempt (see the circular crossing)
            const double ri = s_rad[i];
            const double lo = -(L / 2.0) + ri, hi = L / 2.0 - ri;
            bool placed = false;
            for (int t0 = 0; t0 < g.max_tries && !placed; t0 += 8) {
                if (m.pos >= 624) wmt_advance(m, lane);
                const int base = m.pos, k0 = base + 4 * att;
                const double x = (hi - lo) * wmt_peek_double(m, k0, 0) + lo;
                const double y = (wmt_peek_double(m, k0, 1) - 0.5) * H;
                bool collide = t0 + att >= g.max_tries;
                for (int j = sub; j < i; j += 8) {
                    // norm(...) - ri - radius_j - 0.1 < 0, decided on the squares outside the rounding band of that expression
                    const double dx = x - s_px[j], dy = y - s_py[j], rj = s_rad[j];
                    const double d2 = dx * dx + dy * dy, T = ri + rj + 0.1, T2 = T * T;
                    bool c = d2 < T2;
                    if (fabs(d2 - T2) <= 1e-12 * T2) c = norm2d(dx, dy) - ri - rj - 0.1 < 0.0;
                    collide |= c;
                }
                if (insert_robot && sub == 0) {
                    const double dx = x - rpx, dy = y - rpy;
                    const double d2 = dx * dx + dy * dy, T = ri + rr + 0.1, T2 = T * T;
                    bool c = d2 < T2;
                    if (fabs(d2 - T2) <= 1e-12 * T2) c = norm2d(dx, dy) - ri - rr - 0.1 < 0.0;
                    collide |= c;
                }
                const unsigned long long bad = __builtin_amdgcn_ballot_w64(collide);
                int win = -1;
                for (int q = 7; q >= 0; --q) if (((bad >> (8 * q)) & 0xFFull) == 0) win = q;
                if (win >= 0) {
                    if (lane == 8 * win) { s_px[i] = x; s_py[i] = y; s_yaw[i] = bound_angle_d(-pi); }
                    m.pos = base + 4 * (win + 1);
                    placed = true;
                } else {
                    m.pos = base + 4 * 8;
                }
                __syncthreads();
            }
            if (!placed) status = 1;
        }
        if (lane < n) { mx = s_px[lane]; my = s_py[lane]; myaw = s_yaw[lane]; }
    } else {
        rpx = 0.0; rpy = 0.0 - R; ryaw = pi / 2.0; rgx = 0.0; rgy = 0.0 + R;
        const double inner = R - 3.0;
        const double sector = pi / (double)(n / 2);
        for (int i = 0; i < n && status == 0; ++i) {
            const double ri = s_rad[i];
            bool placed = false;
            for (int t = 0; t < g.max_tries; ++t) {
                double angle, nx, ny, ring;
                if (i < 3) {
                    angle = sector * (-0.5 + 2.0 * i + (wrnd(m, lane) - 0.5) * 0.5);
                    nx = (wrnd(m, lane) - 0.5) * 0.1; ny = (wrnd(m, lane) - 0.5) * 0.1;
                    ring = inner;
                } else {
                    angle = sector * (0.5 + 2.0 * i + (wrnd(m, lane) - 0.5) * 0.5);
                    nx = (wrnd(m, lane) - 0.5) * 0.7; ny = (wrnd(m, lane) - 0.5) * 0.7;
                    ring = R;
                }
                double sa, ca;
                sincos(angle, &sa, &ca);
                const double x = 0.0 + ring * ca + nx, y = 0.0 + ring * sa + ny;
                const double md = ri + myrad + 0.2;
                const double gxj = (lane < 3) ? mx : (-mx + 0.0), gyj = (lane < 3) ? my : (-my + 0.0);
                const bool hit = lane < i && (norm2d(x - mx, y - my) < md || norm2d(x - gxj, y - gyj) < md);
                bool collide = __builtin_amdgcn_ballot_w64(hit) != 0;
                if (norm2d(x - rpx, y - rpy) < ri + rr + 0.2 || norm2d(x - rgx, y - rgy) < ri + rr + 0.2) collide = true;
                if (!collide) {
                    if (lane == i) { mx = x; my = y; myaw = bound_angle_d(pi + angle); }
                    placed = true;
                    break;
                }
            }
            if (!placed) status = 1;
        }
    }
    scenario_out = scenario;
    if (status != 0) return status;

    const float nanf_ = __builtin_nanf("");
    if (lane < n) {
        const int i = lane;
        double g0x, g0y, g1x, g1y;
        int ng;
        if (scenario == CS_SCN_PARALLEL_TRAFFIC) { g0x = -(L / 2.0) - 3.0; g0y = my; g1x = g1y = 0; ng = 1; }
        else if (scenario == CS_SCN_CIRCULAR_CROSSING_STATIC_OBSTACLES && i < 3) { g0x = mx; g0y = my; g1x = mx; g1y = my; ng = 2; }
        else { g0x = 0.0 * 2.0 - mx; g0y = 0.0 * 2.0 - my; g1x = mx; g1y = my; ng = 2; }
        float* s = a.S + ((long)w * a.rows + i) * a.as;
        const long fs = a.fs;
        s[0] = (float)mx; s[fs] = (float)my; s[2 * fs] = (float)myaw;
        for (int c = 3; c < 8; ++c) s[c * fs] = 0.0f;
        s[8 * fs] = (float)myrad; s[9 * fs] = (float)g.human_mass; s[10 * fs] = (float)g0x; s[11 * fs] = (float)g0y;
        s[12 * fs] = (float)s_spd[i];
        if (a.orca) orca_pref(s, fs, (float)mx, (float)my, (float)g0x, (float)g0y, (float)s_spd[i]);
        float* gl = a.goals + ((long)w * n + i) * a.G * 2;
        for (int k = 0; k < a.G; ++k) {
            float gx = nanf_, gy = nanf_;
            if (k == 0) { gx = (float)g0x; gy = (float)g0y; }
            else if (k == 1 && ng == 2) { gx = (float)g1x; gy = (float)g1y; }
            gl[2 * k] = gx; gl[2 * k + 1] = gy;
        }
    }
    if (lane == 0) {
        auto robot_row = [&](float* o, long fs) {
            o[0] = (float)rpx; o[fs] = (float)rpy; o[2 * fs] = (float)ryaw;
            for (int c = 3; c < 8; ++c) o[c * fs] = 0.0f;
            o[8 * fs] = (float)rr; o[9 * fs] = (float)g.robot_mass; o[10 * fs] = (float)rgx; o[11 * fs] = (float)rgy;
            o[12 * fs] = (float)g.robot_desired_speed;
        };
        if (a.robot != nullptr) robot_row(a.robot + (long)w * 13, 1);
        if (a.robot_row) robot_row(a.S + ((long)w * a.rows + n) * a.as, a.fs);
        if (a.world_flags != nullptr) a.world_flags[w] = (scenario == CS_SCN_PARALLEL_TRAFFIC) ? 1 : 0;
    }
    return 0;
}

__global__ __launch_bounds__(64) void k_generate_wave(const GArgs a)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    if (a.mask != nullptr && a.mask[w] == 0) return; // block-uniform
    int scenario = 0;
    const int status = generate_world_wave(a.g, gen_out(a), w, lane, a.seeds[w], scenario);
    if (lane == 0) {
        if (a.status != nullptr) a.status[w] = status;
        if (a.scenario_out != nullptr) a.scenario_out[w] = scenario;
    }
}

// ---- pre-staged episodes (include/crowdstep.h cs_stage_book) -----------------------------------------------------------------
struct StageArgs {
    cs_generator g;
    GenOut staging, live;
    csimpl::CopyArgs copy;          // staging -> live (mask / status unused here)
    const uint32_t* seeds; const uint32_t* base_seed; uint32_t* epoch; uint32_t* staged_seed; int32_t* staged_status; int32_t* failed;
    const int32_t* mask;
    uint32_t stride;
    int W, depth;
};

__device__ __forceinline__ uint32_t load_relaxed(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// ---- the staging protocol: two kernels on two streams, no event, no acquire ----------------------------------------------------------
// Writer: k_refill_staged (side stream).  Readers: k_consume_staged (step stream) and the epilogue of k_sfm_step under cs_gym_step_staged
// (gymhead.h GymFold).  Shared words: the slot's rows / goal lists / robot row / flag / status, its TAG staged_seed[slot], and epoch[w].
// What makes it correct -- each line is an obligation of the code, checked by the tests named at the end:
//   W1  the writer regenerates a slot only when its tag differs from the seed of the episode that belongs in it NOW (a function of
//       epoch[w], base_seed[w], the slot index): a slot a reader may be copying -- tag == the world's next seed, epoch not yet moved --
//       is never written.
//   W2  the writer stores the tag LAST, with a device-scope RELEASE (store_release): a reader that sees the new tag sees every word
//       of the slot the writer stored before it, PROVIDED the reader's loads of those words cannot be served from a line cached before
//       the release -- R2.
//   R1  a reader loads the tag first (device-scope relaxed atomic load) and touches the slot only if tag == the world's next seed:
//       a CONTROL dependency.  No load of a slot word may be issued before the tag has returned and compared equal: the loads sit in the
//       taken branch and are `__hip_atomic_load`s, which the compiler may not hoist above the branch or merge with earlier loads.
//   R2  the slot's words are read with device-scope relaxed atomic loads (copy_world<COHERENT = true>: sc1 loads that miss this XCD's
//       L2 instead of hitting a line from before the refill).  A plain load here would be a silent bug.  Not an acquire on purpose: an
//       acquire fence at device scope invalidates the whole L2 of the XCD beside the running step kernels, each time a world ends.
//   R3  the reader stores epoch[w] (device-scope relaxed) only after EVERY load of the slot has returned: behind a barrier of the
//       block (k_consume_staged) / an s_waitcnt vmcnt(0) of the one wavefront (the step kernel's epilogue).  From that store on W1 lets
//       the writer take the slot.  It is the linearisation point of the take-over.
//   R4  only readers write epoch[w], and one world is read by one block per launch; launches of a reader are ordered on the step stream,
//       launches of the writer on the side stream.
// This is outside the HIP memory model's guarantees for coarse-grained memory between concurrent kernels (no happens-before without
// an acquire); it relies on gfx950's device-scope atomics going to the memory side of the L2s.  The net under it: every regenerated
// episode is compared bit for bit with the in-place generation of the same seed, under three refill cadences, at 96 and at 4096 worlds
// (tests/test_gpu_generators.py::test_step_device_is_the_same_whatever_the_refill_cadence, ::test_one_launch_gym_step_equals_the_two_launches,
// tools/device_loop_soak.py: 191k episodes).  A half-written world would differ from its seed's.
//
// side stream.  Slot s = j * W + w holds the episode of world w that belongs in ring position j now: the one e in (epoch, epoch + depth]
// with e = j mod depth; it is regenerated when its tag is not that episode's seed.  The grid is SMALL (<= 1024 blocks of one wavefront,
// every lane checking one slot per trip, two relaxed device-scope loads): a block per slot -- 65536 workgroups at depth 16 x 4096 worlds,
// nearly all of which leave at once -- kept the workgroup dispatcher busy for ~45 us per pass and the step stream's next launch waited
// behind it.  The slots a wavefront finds stale are generated one after the other by the whole wavefront; only those pay for a release
// (a cache-maintenance operation per checked slot would disturb the step kernel).
__global__ __launch_bounds__(64) void k_refill_staged(const StageArgs a)
{
    const int lane = threadIdx.x;
    const long total = (long)a.W * a.depth;
    for (long s0 = (long)blockIdx.x * 64; s0 < total; s0 += (long)gridDim.x * 64) {
        const long s = s0 + lane;
        uint32_t want = 0;
        bool stale = false;
        if (s < total) {
            const int w = (int)(s % a.W), j = (int)(s / a.W);
            const uint32_t epoch = load_relaxed(a.epoch + w);
            const uint32_t ahead = ((uint32_t)j - epoch - 1u) & (uint32_t)(a.depth - 1);      // 0 .. depth-1
            want = a.base_seed[w] + (epoch + 1u + ahead) * a.stride;
            stale = load_relaxed(a.staged_seed + s) != want;
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(stale);                  // wave-uniform from here on
        while (todo != 0) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const long slot = s0 + l;
            const uint32_t seed = (uint32_t)__builtin_amdgcn_readlane((int)want, l);
            int scenario = 0;
            const int status = generate_world_wave(a.g, a.staging, (int)slot, lane, seed, scenario);
            if (lane == 0) a.staged_status[slot] = status;
            __syncthreads();
            // the tag goes LAST: a consumer that reads `seed` in it (acquire) sees the rows, goal lists, robot row, flag and status above
            if (lane == 0) store_release(a.staged_seed + slot, seed);
            __syncthreads();                                                           // (the generator's LDS is reused by the next slot)
        }
    }
}

// the step's stream: a finished world takes over its staged episode -- or, if the refill has not got to it yet, is generated in place.
// One block per world (4096 workgroups, most of which read one integer and leave: 2.9 us when nobody ends; tools/consume_probe.py): the
// refill's small grid was tried here too -- 64 masks per wavefront, its finished worlds copied one after the other -- and cost 13 us more
// per Gym step (a wavefront with three ends copies serially).
// No acquire / release pair here: an acquire at device scope invalidates this XCD's L2 and a release writes all of it back, each time a
// world ends, beside the step kernels.  The slot is read with device-scope relaxed loads instead (they miss a stale L2 line by themselves),
// after the tag -- a control dependency --, and the epoch is stored after the loads have returned (the barrier waits for them): all the
// refill needs to know before it overwrites the slot is that nobody reads it any more.
__global__ __launch_bounds__(64) void k_consume_staged(const StageArgs a)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    if (!a.mask[w]) return;                                                    // block-uniform (seed and epoch requested with the mask: +2 us for the 4000 blocks that leave here)
    const uint32_t want = a.seeds[w];                                          // the seed the bookkeeping moved this world to
    const uint32_t e = a.epoch[w] + 1u;                                        // (only this kernel writes the epochs)
    const long slot = (long)(e & (uint32_t)(a.depth - 1)) * a.W + w;
    int status;
    if (load_relaxed(a.staged_seed + slot) == want) {                          // written LAST by the refill, behind a release of the slot's words
        status = csimpl::copy_world<true>(a.copy, slot, w, lane, a.staged_status + slot);   // (the status word comes with the first batch)
    } else {
        int scenario = 0;
        status = generate_world_wave(a.g, a.live, w, lane, want, scenario);
        if (status == 0 && a.copy.obs != nullptr) {
            __syncthreads();                                                   // the rows this block just wrote
            for (int k = lane; k < a.copy.n * a.copy.C; k += 64) {
                const int i = k / a.copy.C, c = k - i * a.copy.C;
                a.copy.obs[((long)w * a.copy.n + i) * a.copy.C + c] = a.live.S[((long)w * a.live.rows + i) * a.live.as + csimpl::obs_state_column(c) * a.live.fs];
            }
        }
    }
    if (lane == 0) a.failed[w] = status != 0 ? 1 : 0;
    __syncthreads();                                                           // every lane's loads of the slot have returned
    // LAST: from here on the refill may overwrite the slot (it now belongs to episode e + depth)
    if (lane == 0) __hip_atomic_store(a.epoch + w, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int fill_gen_out(const cs_worlds* w, GenOut& o)
{
    o.robot_row = (w->flags & CS_ROBOT_ROW) ? 1 : 0;
    o.orca = w->type == CS_ORCA ? 1 : 0;
    o.rows = w->n + o.robot_row;
    o.G = w->G;
    o.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { o.as = 13; o.fs = 1; }
    else if (w->layout == CS_LAYOUT_SOA) { o.as = 1; o.fs = (long)w->W * o.rows; }
    else return fail(CS_ERR_ARG, "bad layout");
    o.goals = w->d_goals;
    o.robot = w->d_robot;
    o.world_flags = const_cast<int*>(w->d_world_flags);
    return CS_OK;
}

int check_generator(const cs_generator* gen, const cs_worlds* w)
{
    if (!w->d_state || !w->d_goals) return fail(CS_ERR_ARG, "null device buffer in cs_worlds");
    if (w->W <= 0 || w->n <= 0 || w->G <= 0) return fail(CS_ERR_ARG, "W, n, G must be positive");
    if (gen->n != w->n) return fail(CS_ERR_ARG, "cs_generator.n differs from cs_worlds.n");
    if (gen->n > GEN_MAXN) return fail(CS_ERR_ARG, "the device generators place at most 128 humans per world");
    if (gen->scenario < CS_SCN_CIRCULAR_CROSSING || gen->scenario > CS_SCN_HYBRID) return fail(CS_ERR_ARG, "unknown scenario");
    if (gen->scenario != CS_SCN_PARALLEL_TRAFFIC && w->G < 2) return fail(CS_ERR_ARG, "circular scenarios need G >= 2 goal slots");
    if (gen->scenario == CS_SCN_CIRCULAR_CROSSING_STATIC_OBSTACLES) {
        if (!(gen->circle_radius > 5.0)) return fail(CS_ERR_ARG, "Radius must be greater than 5 for this scenario"); // :375
        if (gen->n < 2) return fail(CS_ERR_ARG, "static-obstacle scenario needs at least 2 humans (sector = pi / (n // 2))");
    }
    if (gen->max_tries <= 0) return fail(CS_ERR_ARG, "max_tries must be positive (the reference's loops are unbounded)");
    return CS_OK;
}

int stage_args(const cs_generator* gen, const cs_worlds* staging, const cs_worlds* live, const cs_stage_book* book, StageArgs& a)
{
    if (!gen || !staging || !book) return fail(CS_ERR_ARG, "null argument");
    if (!book->d_staged_seed || !book->d_epoch || !book->d_base_seed || !book->d_staged_status) return fail(CS_ERR_ARG, "null buffer in cs_stage_book");
    const int K = book->depth;
    if (K <= 0 || (K & (K - 1)) != 0) return fail(CS_ERR_ARG, "cs_stage_book.depth must be a power of two");
    if (staging->W % K != 0) return fail(CS_ERR_ARG, "the staging batch must hold depth * W worlds");
    int rc = check_generator(gen, staging);
    if (rc) return rc;
    if (gen->n > 64) return fail(CS_ERR_ARG, "pre-staged episodes are built for worlds of up to 64 humans (one wavefront per world)");
    std::memset(&a, 0, sizeof(a));
    a.g = *gen;
    if ((rc = fill_gen_out(staging, a.staging))) return rc;
    a.W = staging->W / K; a.depth = K;
    a.seeds = book->d_seeds; a.base_seed = book->d_base_seed; a.epoch = book->d_epoch; a.staged_seed = book->d_staged_seed;
    a.staged_status = book->d_staged_status; a.failed = book->d_failed;
    a.stride = book->seed_stride ? book->seed_stride : (uint32_t)a.W;
    if (live) {
        if (a.W != live->W || staging->n != live->n || staging->G != live->G || staging->layout != live->layout ||
            ((staging->flags ^ live->flags) & CS_ROBOT_ROW) || (!staging->d_robot) != (!live->d_robot) || (!staging->d_world_flags) != (!live->d_world_flags))
            return fail(CS_ERR_ARG, "staging and live worlds differ in shape");
        if ((rc = check_generator(gen, live))) return rc;
        if ((rc = fill_gen_out(live, a.live))) return rc;
        csimpl::CopyArgs& c = a.copy;
        c.W = live->W; c.n = live->n; c.rows = a.live.rows; c.G = live->G;
        c.Ss = staging->d_state; c.Sd = live->d_state; c.as = a.live.as; c.fs = a.live.fs; c.sas = a.staging.as; c.sfs = a.staging.fs;
        c.gs = staging->d_goals; c.gd = live->d_goals; c.rs = staging->d_robot; c.rd = live->d_robot;
        c.fsrc = staging->d_world_flags; c.fdst = const_cast<int*>(live->d_world_flags);
    }
    return CS_OK;
}

} // namespace

int csimpl::stage_fold(const cs_generator* gen, const cs_worlds* staging, const cs_worlds* live, const cs_stage_book* book, int obs_cols, float* d_obs,
                       cstep::GymFold& out)
{
    StageArgs a;
    const int rc = stage_args(gen, staging, live, book, a);
    if (rc) return rc;
    if (!live || !book->d_pending || !book->d_failed) return fail(CS_ERR_ARG, "null argument");
    out.on = 1; out.depth = a.depth; out.W = a.W;
    out.staged_seed = a.staged_seed; out.staged_status = a.staged_status; out.epoch = a.epoch; out.failed = a.failed; out.pending = book->d_pending;
    out.copy = a.copy;
    out.copy.obs = d_obs; out.copy.C = obs_cols;
    return CS_OK;
}

extern "C" {

size_t cs_generate_scratch_bytes(int W) { return W > 0 ? (size_t)624 * (size_t)W * sizeof(uint32_t) : 0; }

int cs_generate_worlds(const cs_generator* gen, const cs_worlds* w, const uint32_t* d_seeds, const int32_t* d_mask,
                       int32_t* d_status, int32_t* d_scenario, void* d_scratch, void* stream)
{
    if (!gen || !w || !d_seeds || !d_scratch) return fail(CS_ERR_ARG, "null argument");
    {
        const int rc = check_generator(gen, w);
        if (rc) return rc;
    }
    GArgs a;
    a.g = *gen;
    a.W = w->W;
    a.robot_row = (w->flags & CS_ROBOT_ROW) ? 1 : 0;
    a.orca = w->type == CS_ORCA ? 1 : 0;
    a.rows = w->n + a.robot_row;
    a.G = w->G;
    a.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; }
    else if (w->layout == CS_LAYOUT_SOA) { a.as = 1; a.fs = (long)w->W * a.rows; }
    else return fail(CS_ERR_ARG, "bad layout");
    a.goals = w->d_goals;
    a.robot = w->d_robot;
    a.world_flags = const_cast<int*>(w->d_world_flags);
    a.seeds = d_seeds;
    a.mask = d_mask;
    a.status = d_status;
    a.scenario_out = d_scenario;
    a.mt = (uint32_t*)d_scratch;
    if (gen->n <= 64) { // one wavefront per world: ~10x shorter latency per world, and faster for full batches too
        hipLaunchKernelGGL(k_generate_wave, dim3(w->W), dim3(64), 0, (hipStream_t)stream, a);
    } else {            // one lane per world, MT19937 state in the scratch buffer
        const int block = 64, grid = (w->W + block - 1) / block;
        hipLaunchKernelGGL(k_generate, dim3(grid), dim3(block), 0, (hipStream_t)stream, a);
    }
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_refill_staged_worlds(const cs_generator* gen, const cs_worlds* staging, const cs_stage_book* book, void* stream)
{
    StageArgs a;
    const int rc = stage_args(gen, staging, nullptr, book, a);
    if (rc) return rc;
    const long slots = (long)a.W * a.depth;
    const int grid = (int)((slots + 63) / 64 < 1024 ? (slots + 63) / 64 : 1024);
    hipLaunchKernelGGL(k_refill_staged, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_consume_staged_worlds(const cs_generator* gen, const cs_worlds* staging, const cs_worlds* live, const int32_t* d_mask,
                             const cs_stage_book* book, int theta_and_omega_visible, float* d_obs, void* stream)
{
    if (!live || !d_mask || !book || !book->d_seeds || !book->d_failed) return fail(CS_ERR_ARG, "null argument");
    StageArgs a;
    const int rc = stage_args(gen, staging, live, book, a);
    if (rc) return rc;
    a.mask = d_mask;
    a.copy.obs = d_obs; a.copy.C = theta_and_omega_visible ? 7 : 5;
    hipLaunchKernelGGL(k_consume_staged, dim3(live->W), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

} // extern "C"
