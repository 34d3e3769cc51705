// stepcommon.h -- what the SFM / HSFM step kernels share (crowdstep.hip: the LDS pair-once / all-partners kernel;
// rowstep.hip: the DPP-row kernel for small worlds): kernel arguments, launch modes, single-instruction math, the pair
// force parameters.  Device code for gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "common.h"
#include "gymhead.h"
#include "respawnx.h"

namespace cstep {

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
    float* obs;            // cs_step_observe: [W][n][obs_cols] observable states of the stepped humans (px, py, vx, vy, radius [, theta, omega]) or null
    int obs_cols;
    int seg_tab;           // wall segments staged in LDS per block (0: read them from global memory in every substep)
    int ws;                // LDS rows between the doubled row blocks of consecutive worlds of a block (>= 2 * rows)
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
    float4* snap;          // optional [nsub][W][n] (x, y, vx, vy) of every human at the START of every substep (imitation block)
    float* trace;          // optional [nsub][W][rows][12] px, py, theta, vx, vy, bvx, bvy, omega, gx, gy, goals[0].x, goals[0].y of every
                           // row AFTER every substep (respawn included; the robot row as the NEXT substep will see it): cs_step_trace
    // the robot under a human motion model inside the crowd's launch (k_sfm_step<..., LEAN = 4>, cs_imitation_block with a visible robot)
    int rm_type;           // the robot's model 0..8
    float rm_margin;       // robot.safety_space as its model adds it to the radius
    const float* rm_hmargin; // [W][rows] the humans' safety space as the robot's model sees it
    float* rm_memory;      // [W][2] robot.desired_force between substeps
    float rm_P[20];        // the robot's parameters
    int wg_waves;          // one-wavefront builds (MAXT = 64): independent wavefronts per workgroup (each a virtual block of its own); 1 elsewhere
    int lds_per_wave;      // ... and the bytes of dynamic LDS each of them owns
    int young_from;        // blocks from this index on are the YOUNGER wavefront of their SIMD (a grid of exactly two wavefronts per SIMD), INT_MAX: no such split
    float wall_efolds;     // a polygon farther than (this many e-folding lengths of the wall force) from every agent of a wavefront is skipped
    int wall_pairs;        // LEAN = 2 builds: the (agent, polygon) pairs within reach over the whole launch are numbered ONCE in the prologue and a
                           // substep evaluates one pair per lane (sfmstep_kernel.h, "wall pairs"); 0: every lane walks every polygon its wavefront is near
    unsigned long long* stamps; // diagnostic build only
    GymHead gym;           // cs_gym_step: reward / termination of the incoming state + episode bookkeeping in the prologue (gym.out == nullptr: none)
};

// ---- math: gfx950 single-instruction transcendentals (1 ulp each) ---------------------------
// The parity bar is 1e-5 absolute on positions/velocities against the f64 reference; v_rsq_f32 /
// v_rcp_f32 / v_sqrt_f32 / v_exp_f32 (<= 1 ulp) stay two orders of magnitude inside it, while the
// IEEE-exact divide / sqrt / libm exp sequences cost 10-15 VALU slots each in an O(N^2) loop.
// Diagnostic build only (-DCS_STAMPS): per-section cycle shares via s_memtime, written to a debug
// buffer that nothing else reads (cdna_hip_programming.md §7 "In-kernel stamps").
#if defined(CS_STAMPS) && !defined(STAMP)
#define STAMP(k)                                                                             \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        unsigned long long t__;                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        st_acc[k] += t__ - st_last;                                                          \
        st_last = t__;                                                                       \
    } while (0)
#elif !defined(STAMP)
#define STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ float rsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float LOG2E = 1.4426950408889634f;
constexpr int PADR = 8;  // partner rows are read in groups of 8: readable (finite) padding behind each buffer

__device__ __forceinline__ float norm2(float x, float y) { return sqrt_fast(fmaf(x, x, y * y)); }

// |d| from d2 and inv = rsq(d2), refined by one Newton step to (almost) a correctly rounded square root.  The body-contact overlap
// max(0, r_ij - dist) is a small difference of two O(1) numbers (4 mm of 0.6 m): ONE ulp of dist is 1.4e-5 of the k1 / k2 contact
// force (120 kN/m), and d2 * v_rsq(d2) alone is ~2.5 ulp off -- measured 1.6e-5 per substep on a Moussaid contact against the
// float32 oracle's 1e-6 (its sqrt is correctly rounded).  The exponential terms keep the unrefined distance (A / B = 25 kN/m,
// and they dominate the instruction count).
__device__ __forceinline__ float dist_refined(float d2, float inv)
{
    const float d0 = d2 * inv;
    return fmaf(fmaf(-d0, d0, d2), 0.5f * inv, d0);
}

// one human's record of cs_step_trace (three 16-byte stores; the record stride is 48 bytes)
__device__ __forceinline__ void write_trace(float* o, float px, float py, float th, float vx, float vy, float bvx, float bvy, float om,
                                            float gx, float gy, float g0x, float g0y)
{
    float4* q = reinterpret_cast<float4*>(o);
    q[0] = make_float4(px, py, th, vx);
    q[1] = make_float4(vy, bvx, bvy, om);
    q[2] = make_float4(gx, gy, g0x, g0y);
}

// sin and cos together: the hardware v_sin_f32 / v_cos_f32 (input in revolutions), 1 multiply + 2 quarter-rate instructions instead
// of the 20-instruction Cody-Waite + minimax form the first rounds used.  Measured on MI355X over 2^26 points of [-pi, pi]
// (tools/sincos_probe.hip): max |error| 2.7e-7 for both, the conversion to revolutions included -- 6e-7 on a velocity R(theta) bv
// with |bv| <= 1.5, inside the 1e-5 parity bar with the rest of a substep's float32 rounding (worst measured per substep: see
// profiles/*parity_report.json).  theta is kept in [-pi, pi] by wrap_angle (the instructions take |revolutions| <= 256).
__device__ __forceinline__ void sincos_fast(float x, float& s, float& c)
{
    const float rev = x * 0.15915494309189535f;
    s = __builtin_amdgcn_sinf(rev);
    c = __builtin_amdgcn_cosf(rev);
}

// python's x % (2 pi) of a float32 heading (robot_agent.py:131: yaw = (yaw + r) % (2 * np.pi), in [0, 2 pi)): x - floor(x / 2 pi) * 2 pi with a
// two-term 2 pi (more exact than fmodf by fl32(2 pi), which is 1.7e-7 off 2 pi per wrap), a dozen instructions instead of ocml's fmodf loop.
__device__ __forceinline__ float mod_two_pi(float x)
{
    const float k = floorf(x * 0.15915494309189535f);
    float t = fmaf(-k, 6.2831855f, x);            // 6.2831855f = fl32(2 pi) = 2 pi + 1.7484555e-7
    t = fmaf(k, 1.7484555e-7f, t);
    if (t < 0.0f) t += 6.2831855f;
    if (t >= 6.2831855f) t -= 6.2831855f;
    return t;
}

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

// bound_angle for the heading update, branch-free: a - 2 pi rint(a / 2 pi) with a two-term (Cody-Waite) 2 pi, which
// is what the reference's branches give for any |a| (fmod by 2 pi, then one fold into [-pi, pi]); ties at exactly
// +-pi stay, as in the reference.  A data-dependent branch costs ~55 cycles of wave latency on this SIMD
// (tools/valu_microbench.hip), the four instructions below ~12.
__device__ __forceinline__ float wrap_angle(float a)
{
    const float k = rintf(a * 0.15915494309189535f);
    a = fmaf(k, -6.2831854820251465f, a);      // float(2 pi)
    a = fmaf(k, 1.7484556000744883e-07f, a);   // float(2 pi) - 2 pi
    // second pass: the identity (k2 = 0, bit for bit) whenever the first one landed in [-pi, pi]; where |omega dt| is so large that
    // float32 cannot hold the heading at all (the reference's own explicit Euler on omega has diverged: 1e11 rad per substep in
    // crushed hsfm_new* crowds) the first pass leaves hundreds of radians, and the heading that is STORED must still be one whose
    // sine / cosine the next substep gets to full accuracy (v_sin_f32 / v_cos_f32 lose 6e-8 per revolution of their argument)
    const float k2 = rintf(a * 0.15915494309189535f);
    return fmaf(k2, -6.2831854820251465f, a);
}

// atan2 with |error| < 2e-7 rad (degree-7 minimax in a^2 on [0,1], a = min/max); atan2(0, 0) = 0
__device__ __forceinline__ float atan2_fast(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1e-30f), mn = fminf(ax, ay);
    const float a = mn * rcp_fast(mx);
    const float s = a * a;
    float r = -0.004054529592394829f;
    r = fmaf(r, s, 0.021862812340259552f);
    r = fmaf(r, s, -0.05591210350394249f);
    r = fmaf(r, s, 0.09642179310321808f);
    r = fmaf(r, s, -0.13908621668815613f);
    r = fmaf(r, s, 0.19946563243865967f);
    r = fmaf(r, s, -0.33329859375953674f);
    r = fmaf(r, s, 0.9999993443489075f);
    r *= a;
    if (ay > ax) r = 1.5707963267948966f - r;
    if (x < 0.0f) r = 3.141592653589793f - r;
    return copysignf(r, y);
}

// atan2(y, x) of a UNIT vector (x, y) -- Moussaid's theta_ij is the angle between two normalised vectors, (cross, dot) of them: the smaller of
// |x|, |y| is then the sine of an angle in [0, pi / 4], and asin(s) = s P(s^2) on [0, 0.7072] (degree-6 minimax fit, 2e-7 rad) needs neither
// the reciprocal nor the max of atan2_fast.  The norm of the argument is 1 to a few float32 ulps (v_rsq): 4e-7 rad at the octant's edge.
__device__ __forceinline__ float atan2_unit(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float sn = fminf(ax, ay);
    const float t = sn * sn;
    float r = 0.1019788458943367f;
    r = fmaf(r, t, -0.05660269409418106f);
    r = fmaf(r, t, 0.06181516870856285f);
    r = fmaf(r, t, 0.03858460113406181f);
    r = fmaf(r, t, 0.07553976029157639f);
    r = fmaf(r, t, 0.16664893925189972f);
    r = fmaf(r, t, 1.0000001192092896f);
    r *= sn;
    if (ay > ax) r = 1.5707963267948966f - r;
    if (x < 0.0f) r = 3.141592653589793f - r;
    return copysignf(r, y);
}

// Social-force parameters used inside the pair loop (subset of the 20-vector, agent.py:268-388),
// with the exponent scales pre-multiplied: exp(x / B) = exp2(x * (log2(e) / B))
struct SocP {
    float Ai, cB, Ci, cD, Ei, k1, k2, lam, gam, ns, ns1;
    float lA, sA, lC, sC; // A * exp(x/B) = sA * exp2(x * cB + lA),  lA = log2|A|  (A = 0 -> exp2(-inf) = 0)
    float sAC;            // sA * sC
    float cg, c1, c2;     // Moussaid: log2(e) / gamma, -ns1^2 log2(e), -ns^2 log2(e)
};

// the social-force entries of a parameter row as loaded (no arithmetic: a caller issues its other loads before it uses them) ...
struct SocRaw { float v[11]; };   // P[1], [3], [5], [7], [9], [10] .. [15]
__device__ __forceinline__ SocRaw load_socraw(const float* P)
{
    SocRaw q;
    q.v[0] = P[1]; q.v[1] = P[3]; q.v[2] = P[5]; q.v[3] = P[7]; q.v[4] = P[9];
#pragma unroll
    for (int k = 0; k < 6; ++k) q.v[5 + k] = P[10 + k];
    return q;
}
// ... and the derived constants of the pair loop
__device__ __forceinline__ SocP make_socp(const SocRaw& q)
{
    SocP s;
    s.Ai = q.v[0]; s.cB = LOG2E / q.v[1]; s.Ci = q.v[2]; s.cD = LOG2E / q.v[3]; s.Ei = q.v[4];
    s.k1 = q.v[5]; s.k2 = q.v[6]; s.lam = q.v[7]; s.gam = q.v[8]; s.ns = q.v[9]; s.ns1 = q.v[10];
    s.lA = log2f(fabsf(s.Ai)); s.sA = copysignf(1.0f, s.Ai);
    s.lC = log2f(fabsf(s.Ci)); s.sC = copysignf(1.0f, s.Ci);
    s.sAC = s.sA * s.sC;
    s.cg = LOG2E / s.gam; s.c1 = -(s.ns1 * s.ns1) * LOG2E; s.c2 = -(s.ns * s.ns) * LOG2E;
    return s;
}
__device__ __forceinline__ SocP load_socp(const float* P) { return make_socp(load_socraw(P)); }

// forces_parallel.py:120-130 / :72-83 -- Moussaid force on (pi, vi) from (pj, vj); `skip` marks the
// lane's own row / padding.  rij = r_i + r_j + safety_i + safety_j.
__device__ __forceinline__ void pair_force_moussaid(const SocP& p, float pix, float piy, float vix, float viy,
                                                    float pjx, float pjy, float vjx, float vjy, float rij,
                                                    bool skip, float& fx, float& fy)
{
    const float dx = pix - pjx, dy = piy - pjy;
    const float d2 = skip ? 1.0f : fmaf(dx, dx, dy * dy);
    const float inv = rsq_fast(d2);
    const float dist = dist_refined(d2, inv);
    const float nx = dx * inv, ny = dy * inv;
    const float rd = rij - dist;
    const float m0 = fmaxf(0.0f, rd);
    const float vdx = vix - vjx, vdy = viy - vjy;
    const float ivx = fmaf(p.lam, vdx, -nx), ivy = fmaf(p.lam, vdy, -ny);
    const float i2 = fmaxf(fmaf(ivx, ivx, ivy * ivy), 1e-30f);
    const float iinv = rsq_fast(i2);
    const float inorm = i2 * iinv;
    const float ix = ivx * iinv, iy = ivy * iinv;
    // theta_ij = wrap(angle(n) - angle(i) + pi) is the signed angle from i to -n (:124): one atan2 of (cross, dot)
    const float th = atan2_unit(iy * nx - ix * ny, -(ix * nx + iy * ny));
    const float k = (th > 0.0f) ? 1.0f : ((th < 0.0f) ? -1.0f : 0.0f);
    const float hx = -iy, hy = ix;
    const float F = p.gam * inorm;
    const float dv = -(vdx * hx + vdy * hy);
    const float w0 = -dist * rcp_fast(F) * LOG2E;                               // (two exponentials of summed exponents: see pair_force_moussaid_once)
    const float a1 = p.ns1 * F * th, a2 = p.ns * F * th;
    const float e1 = p.Ei * exp2_fast(fmaf(-(a1 * a1), LOG2E, w0)), e2 = p.Ei * exp2_fast(fmaf(-(a2 * a2), LOG2E, w0));
    const float sel = skip ? 0.0f : 1.0f;
    const float kk = p.k1 * m0, kt = p.k2 * m0 * dv;
    fx -= sel * ((e1 + kk) * ix + (k * e2 + kt) * hx);
    fy -= sel * ((e1 + kk) * iy + (k * e2 + kt) * hy);
}

// Same force, returned instead of accumulated (no own-row / padding slot): used by the pair-once loop, which
// also hands -f to the partner (the reference's all_params_equal path does exactly that, :100-104).
// CONTACT = false (round 6, the one-wavefront pair-once loop): without the k1 / k2 body-contact terms -- exact zeros unless the two bodies
// overlap -- which a contact pass behind a wave vote adds for the rows that touch somebody (moussaid_contact_term below; the Helbing / Guo
// loops have worked that way since round 1); rd_out = rij - dist, what the vote is taken on.
template <bool CONTACT = true>
__device__ __forceinline__ void pair_force_moussaid_once(const SocP& p, float dx, float dy, float vdx, float vdy,
                                                         float rij, float& fx, float& fy, float* rd_out = nullptr)
{
    const float d2 = fmaf(dx, dx, dy * dy);
    const float inv = rsq_fast(d2);
    const float nx = dx * inv, ny = dy * inv;
    const float ivx = fmaf(p.lam, vdx, -nx), ivy = fmaf(p.lam, vdy, -ny);
    const float i2 = fmaxf(fmaf(ivx, ivx, ivy * ivy), 1e-30f);
    const float iinv = rsq_fast(i2);            // 1 / |w|: also 1 / F up to gamma (F = gamma |w|), no reciprocal of its own
    const float inorm = i2 * iinv;
    const float ix = ivx * iinv, iy = ivy * iinv;
    const float th = atan2_unit(iy * nx - ix * ny, -(ix * nx + iy * ny));
    const float hx = -iy, hy = ix;
    // Ei e^{-dist / F} e^{-(ns1 F theta)^2} and Ei e^{-dist / F} e^{-(ns F theta)^2} as TWO exponentials of summed exponents (round 6: three, and
    // two more products, until then; the sums differ from the products by a rounding of the exponent: < 1e-6 relative)
    const float u = p.gam * inorm * th, u2 = u * u;                            // (F theta)^2 shared by the two Gaussians
    if constexpr (CONTACT) {
        const float dist = dist_refined(d2, inv);
        const float m0 = fmaxf(0.0f, rij - dist);
        const float dv = -(vdx * hx + vdy * hy);
        const float w0 = -dist * iinv * p.cg;                                   // -dist / F in base-2 exponent units
        const float e1 = p.Ei * exp2_fast(fmaf(u2, p.c1, w0)), e2 = p.Ei * exp2_fast(fmaf(u2, p.c2, w0));
        const float e2s = (th == 0.0f) ? 0.0f : copysignf(e2, th);              // sign(theta) e2
        const float kk = p.k1 * m0, kt = p.k2 * m0 * dv;
        fx = -((e1 + kk) * ix + (e2s + kt) * hx);
        fy = -((e1 + kk) * iy + (e2s + kt) * hy);
        if (rd_out) *rd_out = rij - dist;
    } else {
        const float dist = d2 * inv;                                           // (the exponent does not need the refined root: 2.5 ulp of dist / F)
        const float w0 = -dist * iinv * p.cg;
        const float e1 = p.Ei * exp2_fast(fmaf(u2, p.c1, w0)), e2 = p.Ei * exp2_fast(fmaf(u2, p.c2, w0));
        const float e2s = (th == 0.0f) ? 0.0f : copysignf(e2, th);
        fx = -(e1 * ix + e2s * hx);
        fy = -(e1 * iy + e2s * hy);
        *rd_out = rij - dist;
    }
}

// The body-contact part of the Moussaid force on (me) from a partner it may overlap (forces_parallel.py:130: k1 max(0, rd) i_ij + k2 max(0, rd) dv h_ij,
// negated like the whole force): what pair_force_moussaid_once<false> leaves out.  (dx, dy) = p_me - p_j, (vdx, vdy) = v_me - v_j.
__device__ __forceinline__ void moussaid_contact_term(const SocP& p, float dx, float dy, float vdx, float vdy, float rij, float& tx, float& ty)
{
    const float d2 = fmaf(dx, dx, dy * dy);
    const float inv = rsq_fast(d2);
    const float m0 = fmaxf(0.0f, rij - dist_refined(d2, inv));
    const float nx = dx * inv, ny = dy * inv;
    const float ivx = fmaf(p.lam, vdx, -nx), ivy = fmaf(p.lam, vdy, -ny);
    const float i2 = fmaxf(fmaf(ivx, ivx, ivy * ivy), 1e-30f);
    const float iinv = rsq_fast(i2);
    const float ix = ivx * iinv, iy = ivy * iinv;
    const float hx = -iy, hy = ix;
    const float dv = -(vdx * hx + vdy * hy);
    const float kk = p.k1 * m0, kt = p.k2 * m0 * dv;
    tx = -(kk * ix + kt * hx);
    ty = -(kk * iy + kt * hy);
}


} // namespace cstep
