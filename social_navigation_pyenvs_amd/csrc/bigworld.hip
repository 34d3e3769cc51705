// bigworld.hip -- the SFM / HSFM step for worlds beyond one block (more than 1024 rows; SURVEY.md §8 row f3, second half).
//
// Same reference path as crowdstep.hip (update_humans_parallel, /root/reference/social_gym/src/forces_parallel.py:185-284; the
// per-agent social force :43-84, all pairs :87-133), re-laid for thousands of agents in ONE world: a world no longer fits a
// block's LDS, so the rows stay in HBM (double-buffered: every agent reads the incoming rows, as the reference evaluates all
// forces from the incoming state), lane = agent across as many blocks as the world needs, and the partners of an agent are found
// through a uniform grid instead of a walk over all rows: cells whose edge is the force's reach (the distance beyond which a
// partner's force is below |A| e^-36 ~ 5e-13 N: radii + 36 e-folding lengths, from the parameters and radii of the world
// itself), hashed into buckets, counting-sorted per substep; an agent walks the 3 x 3 cells around its own.  In a world small
// against the reach (Moussaid's exp(-d / F) reaches tens of metres) the walk degenerates to all rows, still exact.
// Every ordered pair is evaluated by its own lane (no hand-over between blocks); the contact terms are part of the pair formula.
//
// Supported: all nine types, all_params_equal or per-agent parameters, polygon walls, goal lists of any length, the robot as the
// last row of the state array (a source, never updated), cs_step / cs_update_humans_parallel (in or out of place).
// Round 3: the parallel-traffic respawn rule (k_bw_respawn), a robot handed over and moved through cs_worlds.d_robot (k_bw_robot),
// cs_peek (nothing committed), and the cell lists by an in-tree counting sort (no library kernel on the path).
//
// gfx950 only: no portability macros, no CPU fallback.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "common.h"
#include "crowdstep.h"
#include "stepcommon.h"

namespace {

using namespace cstep;
using csimpl::fail;

struct GArgs {
    int W, n, rows, G, O, Smax, NB, type, flags, mutate;
    float dt;
    const float* Sin; float* Sout; float* Smut; long as, fs;
    float* goals; const float* params; const float* safety; const float* obstacles;
    const int2* cellxy; const int* start; const int* sorted;   // the grid (csimpl::grid_build)
    float* inv_cell;   // [W] device scalars: 1 / cell edge of every world
    float2* in_v;      // [W][rows] refreshed linear velocity of the incoming rows (out-of-place update: written back afterwards)
    int peek;          // cs_peek: goal lists are not rotated in memory, the head each human WOULD have goes to peek_goal
    float2* peek_goal; // [W][n]
    float* peek_out;   // [W][n][8]
    float bx, by;      // respawn bounds
    const int* world_flags;
    float* robot;      // [W][13] cs_worlds.d_robot (robot rows handed over through the array, motion_model_manager.py:359)
    const float* action;
};

// reach of the pair force in every world: max over rows of (r + safety) twice, plus 36 e-folding lengths of the slowest-decaying
// exponential any row's parameters describe (Helbing B; Guo also D; Moussaid F <= gamma (lambda (vd_i + vd_j) + 1))
__global__ __launch_bounds__(256) void k_bw_reach(const GArgs a)
{
    __shared__ float red[3][256];
    const int w = blockIdx.x, t = threadIdx.x;
    float rs = 0.0f, len = 0.0f, vd = 0.0f;
    for (int i = t; i < a.rows; i += 256) {
        const float* s = a.Sin + ((long)w * a.rows + i) * a.as;
        rs = fmaxf(rs, s[8 * a.fs] + a.safety[(long)w * a.rows + i]);
        vd = fmaxf(vd, fmaxf(s[12 * a.fs], sqrtf(s[3 * a.fs] * s[3 * a.fs] + s[4 * a.fs] * s[4 * a.fs])));
    }
    const int soc = a.type % 3;
    for (int i = t; i < a.n; i += 256) {
        const float* P = a.params + ((a.flags & CS_PARAMS_SHARED) ? 0 : (long)w * a.n * 20) + (long)i * 20;
        float l = fabsf(P[3]);
        if (soc == 1) l = fmaxf(l, fabsf(P[7]));
        if (soc == 2) l = fabsf(P[13]);                    // gamma; completed below with the speeds
        len = fmaxf(len, l);
    }
    red[0][t] = rs; red[1][t] = len; red[2][t] = vd;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) for (int k = 0; k < 3; ++k) red[k][t] = fmaxf(red[k][t], red[k][t + off]);
        __syncthreads();
    }
    if (t == 0) {
        float l = red[1][0];
        if (soc == 2) {
            float lam = 0.0f;
            for (int i = 0; i < a.n; ++i) lam = fmaxf(lam, fabsf(a.params[((a.flags & CS_PARAMS_SHARED) ? 0 : (long)w * a.n * 20) + (long)i * 20 + 12]));
            l = l * (lam * 2.0f * red[2][0] + 1.0f);
        }
        const float reach = 2.0f * red[0][0] + 36.0f * l;
        a.inv_cell[w] = 1.0f / fmaxf(reach, 0.5f);
    }
}

// ---- the uniform grid (shared with the ORCA grid path, orca.hip) -------------------------------------------------------
// key of row i of world w = w * NB + bucket(cell of i).  The cell lists are built by a hand-written COUNTING SORT whose result does
// not depend on the order in which the hardware executes anything: (1) histogram of the keys (integer atomics: a count is
// order-independent), (2) exclusive scan -> start[], (3) scatter of the rows into their bucket's segment in whatever order the
// atomics hand out, (4) every row counts the rows of its segment with a SMALLER index: that count is its final place.  Every
// bucket's rows therefore stand in index order, so the walk over a cell -- and with it the floating-point order of the force sums
// -- is the same in every run (fused substeps and repeated launches must agree bit for bit).  Buckets hold ~0.5 rows on average
// (NB >= 2 rows), so step (4) reads a handful of entries per row.
__global__ void k_grid_keys(csimpl::GridView g, const float* S, long as, long fs, const float* d_inv_cell, float inv_cell)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (i >= g.rows) return;
    const float* s = S + ((long)w * g.rows + i) * as;
    const float ic = d_inv_cell ? d_inv_cell[w] : inv_cell;
    const int cx = (int)floorf(s[0] * ic), cy = (int)floorf(s[fs] * ic);
    const long k = (long)w * g.rows + i;
    const unsigned key = (unsigned)(w * g.NB + csimpl::cell_bucket(cx, cy, g.NB));
    g.cellxy[k] = make_int2(cx, cy);
    g.keys[k] = key;
    atomicAdd(&g.fill[key], 1);
}

// exclusive scan of `in` [M] into `out` [M] in three launches: 1024 entries per block (a wavefront-level scan of the four entries of
// every lane, the four wavefront totals through LDS), the block totals scanned by one block, the offsets added back
__global__ __launch_bounds__(256) void k_scan_blocks(const int* in, int* out, int* bsum, int M)
{
    __shared__ int wsum[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long base = (long)blockIdx.x * 1024 + t * 4;
    int v[4], run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < M) ? in[base + k] : 0; run += v[k]; }
    int inc = run;                                       // inclusive scan of the lanes' totals inside the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < wv; ++k) wbase += wsum[k];
    int ex = wbase + inc - run;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < M) out[base + k] = ex; ex += v[k]; }
    if (t == 255) bsum[blockIdx.x] = wbase + inc;
}
__global__ __launch_bounds__(256) void k_scan_sums(int* bsum, int nblk)
{
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nblk; c0 += 256) {
        const int v = (c0 + t < nblk) ? bsum[c0 + t] : 0;
        int inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int wbase = carry_s;
        for (int k = 0; k < wv; ++k) wbase += wsum[k];
        if (c0 + t < nblk) bsum[c0 + t] = wbase + inc - v;
        __syncthreads();
        if (t == 255) carry_s = wbase + inc;
        __syncthreads();
    }
}
__global__ void k_scan_add(int* out, const int* bsum, int M, int total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] += bsum[i >> 10];
    else if (i == M) out[i] = total;
}

__global__ void k_grid_scatter(csimpl::GridView g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (i >= g.rows) return;
    const unsigned key = g.keys[(long)w * g.rows + i];
    g.tmp[g.start[key] + atomicAdd(&g.fill[key], 1)] = i;
}

__global__ void k_grid_rank(csimpl::GridView g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (i >= g.rows) return;
    const unsigned key = g.keys[(long)w * g.rows + i];
    const int p0 = g.start[key], p1 = g.start[key + 1];
    int rank = 0;
    for (int p = p0; p < p1; ++p) rank += (g.tmp[p] < i) ? 1 : 0;
    g.sorted[p0 + rank] = i;
}

template <int SOC, int HEADED, bool PEQ>
__global__ __launch_bounds__(256) void k_bw_sfm_step(const GArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y, rows = a.rows, n = a.n;
    if (i >= rows) return;
    const long fs = a.fs;
    const float* Sw = a.Sin + (long)w * rows * a.as;
    const float* s = Sw + (long)i * a.as;
    float px = s[0], py = s[fs], th = s[2 * fs], vx = s[3 * fs], vy = s[4 * fs], bvx = s[5 * fs], bvy = s[6 * fs], om = s[7 * fs];
    const float r = s[8 * fs], m = s[9 * fs], vd = s[12 * fs];
    float gx = s[10 * fs], gy = s[11 * fs];
    float* o = a.Sout + ((long)w * rows + i) * a.as;
    if (i >= n) {   // the robot row: a source of force, never updated (forces_parallel.py:212-213)
        for (int f = 0; f < 13; ++f) o[f * fs] = s[f * fs];
        return;
    }
    const float safety = a.safety[(long)w * rows + i];
    const long pw = (a.flags & CS_PARAMS_SHARED) ? 0 : (long)w * n * 20;
    const float* P = a.params + pw + (long)i * 20;
    const SocP sp = load_socp(PEQ ? a.params + pw : P);
    const float m_tau = m / P[0];
    const float ko = P[16], kd = P[17], alpha = P[18], klam = P[19];
    const float dt = a.dt, inertia = 0.5f * m * r * r, dt_inertia = dt / inertia;
    const int obs_type = (a.type == 1 || a.type == 4 || a.type == 7) ? 1 : 0;

    // -- goal switch, forces_parallel.py:226-234 (on the incoming position; the list rotates in place)
    float* gi = a.goals + ((long)w * n + i) * a.G * 2;
    {
        const float gdx = gi[0] - px, gdy = gi[1] - py;
        if (fmaf(gdx, gdx, gdy * gdy) <= r * r) {
            int k = a.G;
            for (int g = a.G - 1; g >= 0; --g)
                if (isnan(gi[2 * g]) || isnan(gi[2 * g + 1])) k = g;
            if (a.peek) {                     // nothing is committed: the head the rotated list would have
                gx = k > 1 ? gi[2] : gi[0]; gy = k > 1 ? gi[3] : gi[1];
            } else {
                const float r0 = gi[0], r1 = gi[1];
                for (int g = 0; g + 1 < k; ++g) { gi[2 * g] = gi[2 * g + 2]; gi[2 * g + 1] = gi[2 * g + 3]; }
                if (k > 0) { gi[2 * (k - 1)] = r0; gi[2 * (k - 1) + 1] = r1; }
                gx = gi[0]; gy = gi[1];
            }
            if (a.peek) a.peek_goal[(long)w * n + i] = make_float2(gx, gy);
        } else if (a.peek) a.peek_goal[(long)w * n + i] = make_float2(gi[0], gi[1]);   // human.goals[0] (the goals array, :297)
    }
    float cs = 1.0f, sn = 0.0f, cvx = vx, cvy = vy;
    if constexpr (HEADED > 0) {
        sincos_fast(th, sn, cs);
        cvx = cs * bvx + (-sn) * bvy;
        cvy = sn * bvx + cs * bvy;
    }
    // -- social force: my partners through the grid
    const float my_rs = r + safety;
    const float vix = PEQ ? vx : cvx, viy = PEQ ? vy : cvy;   // (all_params_equal: every row's stored velocity, :220; else :256 then :261)
    // The pair and wall forces of THIS path are evaluated and summed in DOUBLE (round 6).  A world of thousands packs bodies into contact:
    // single terms reach 1e5 N (k1 = 1.2e5 N/m on centimetres of overlap, A e^{rd/B} beyond that) while their sum -- a human squeezed between
    // neighbours and walls -- stays near 1e2 N, and the velocity clamp turns the sum's DIRECTION into the result: float32 terms (1e-7 relative:
    // 0.01 N each) and a float32 sum (ulp(1e5 N) = 0.008 N per addition) left 5e-5 m/s on such rows, as the float32 instantiation of the
    // oracle does (profiles/r5final_parity_report.json "grid path per substep inside the fused block").  The state stays float32; the
    // LDS kernels (worlds of one block: every BASELINE.json configuration) keep their float32 pair loop -- they are within 1e-5 on every
    // substep of every build (tests/test_gpu_parity.py) because a Gym crowd never reaches that regime.
    double fsx = 0.0, fsy = 0.0;
    {
        const int2 mc = a.cellxy[(long)w * rows + i];
        const int* st = a.start + (long)w * a.NB;     // positions in the job-wide sorted list
        const int* so = a.sorted;
        const int2* cxy = a.cellxy + (long)w * rows;
        const float* saf = a.safety + (long)w * rows;
        const float* Pq = PEQ ? a.params + pw : P;    // the parameters of the force ON me (forces_parallel.py:43-84: agent i's own row)
        const double Ai = Pq[1], Bi = Pq[3], Ci = Pq[5], Di = Pq[7], k1s = Pq[10], k2s = Pq[11];
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int cx = mc.x + dx, cy = mc.y + dy;
                const int bk = csimpl::cell_bucket(cx, cy, a.NB);
                for (int p = st[bk]; p < st[bk + 1]; ++p) {
                    const int j = so[p];
                    const int2 cb = cxy[j];
                    if (cb.x != cx || cb.y != cy || j == i) continue;
                    const float* q = Sw + (long)j * a.as;
                    const float qx = q[0], qy = q[fs];
                    float vjx = q[3 * fs], vjy = q[4 * fs];
                    if constexpr (!PEQ && HEADED > 0) {
                        if (j < i && j < n) {   // rows before mine have had their linear velocity refreshed in place (:256, range order)
                            float sj, cj;
                            sincos_fast(q[2 * fs], sj, cj);
                            const float bx = q[5 * fs], by = q[6 * fs];
                            vjx = cj * bx - sj * by; vjy = sj * bx + cj * by;
                        }
                    }
                    if constexpr (SOC == 2) {
                        const float rij = my_rs + q[8 * fs] + saf[j];
                        float tx = 0.0f, ty = 0.0f;
                        pair_force_moussaid(sp, px, py, vix, viy, qx, qy, vjx, vjy, rij, false, tx, ty);
                        fsx += (double)tx; fsy += (double)ty;
                    } else {
                        const double rij = ((double)r + (double)safety) + ((double)q[8 * fs] + (double)saf[j]);
                        const double ddx = (double)px - (double)qx, ddy = (double)py - (double)qy;
                        const double d = sqrt(fmax(ddx * ddx + ddy * ddy, 1e-60));
                        const double rd = rij - d;
                        const double m0 = fmax(0.0, rd);
                        const double nx = ddx / d, ny = ddy / d;
                        const double dv = ((double)vjy - (double)viy) * nx - ((double)vjx - (double)vix) * ny;        // (v_j - v_i) . t
                        const double fn = Ai * exp(rd / Bi) + k1s * m0;
                        double ft = (k2s * m0) * dv;
                        if constexpr (SOC == 1) ft += Ci * exp(rd / Di);
                        fsx += fn * nx - ft * ny;
                        fsy += fn * ny + ft * nx;
                    }
                }
            }
    }
    // -- desired force, :23-40
    float fdx, fdy;
    {
        const float dx = gx - px, dy = gy - py;
        const float d2 = fmaf(dx, dx, dy * dy);
        const float inv = rsq_fast(fmaxf(d2, 1e-30f));
        const bool far_ = d2 * inv > r;
        fdx = far_ ? m_tau * (dx * inv * vd - cvx) : 0.0f;
        fdy = far_ ? m_tau * (dy * inv * vd - cvy) : 0.0f;
    }
    // -- obstacle force: closest point per polygon :236-252, then :136-162
    double fox = 0.0, foy = 0.0;
    if (a.O > 0) {
        const float* obst = a.obstacles + ((a.flags & CS_OBSTACLES_SHARED) ? 0 : (long)w * a.O * a.Smax * 4);
        const double Awd = P[2], Bwd = P[4], Cwd = P[6], Dwd = P[8], k1d = P[10], k2d = P[11];
        for (int ob = 0; ob < a.O; ++ob) {
            double best = INFINITY, bdx = 0.0, bdy = 0.0;
            for (int sg = 0; sg < a.Smax; ++sg) {
                const float4 seg = *reinterpret_cast<const float4*>(obst + ((long)ob * a.Smax + sg) * 4);
                if (isnan(seg.x)) continue;          // (NaN padding: never the minimum)
                const double ex = (double)seg.z - (double)seg.x, ey = (double)seg.w - (double)seg.y;
                const double t = (((double)px - (double)seg.x) * ex + ((double)py - (double)seg.y) * ey) / (ex * ex + ey * ey);
                const double ts = fmin(fmax(t, 0.0), 1.0);
                const double ddx = (double)px - ((double)seg.x + ts * ex), ddy = (double)py - ((double)seg.y + ts * ey);
                const double d = ddx * ddx + ddy * ddy;
                if (d < best) { best = d; bdx = ddx; bdy = ddy; }   // first minimum (argmin, :252)
            }
            const double dist = sqrt(fmax(best, 1e-60));
            const double nx = bdx / dist, ny = bdy / dist;
            const double dv = -((double)cvy * nx - (double)cvx * ny);
            const double rd = ((double)r - dist) + (double)safety;
            const double m0 = fmax(0.0, rd);
            const double fn = Awd * exp(rd / Bwd) + k1d * m0;
            const double ft = obs_type == 0 ? -(k2d * m0) * dv : (-Cwd * exp(rd / Dwd) - k2d * m0) * dv;
            fox += fn * nx - ft * ny;
            foy += fn * ny + ft * nx;
        }
        fox /= (double)a.O; foy /= (double)a.O;
    }
    // -- total force, body frame, torque, explicit Euler  :262-283
    const double fixd = (double)fdx + fox + fsx, fiyd = (double)fdy + foy + fsy;
    const float fix = (float)fixd, fiy = (float)fiyd;
    const float in_vx = cvx, in_vy = cvy;
    px += vx * dt; py += vy * dt;            // the velocity stored in the incoming row
    const double dt_md = (double)dt / (double)m;
    if constexpr (HEADED > 0) {
        const float tfx = HEADED == 1 ? fdx : fix, tfy = HEADED == 1 ? fdy : fiy;
        const float kf = klam * norm2(tfx, tfy);
        const float k_theta = inertia * kf;
        const float k_omega = inertia * (1.0f + alpha) * sqrt_fast(kf / alpha);
        const float delta = atan2_fast(sn * tfx - cs * tfy, cs * tfx + sn * tfy);
        const float torque = -k_theta * delta - k_omega * om;
        const double gfx = fixd * (double)cs + fiyd * (double)sn;
        const double gfy = (double)ko * ((fox + fsx) * (double)(-sn) + (foy + fsy) * (double)cs) - (double)kd * (double)bvy;
        th = wrap_angle(fmaf(om, dt, th));
        double bxd = (double)bvx + gfx * dt_md, byd = (double)bvy + gfy * dt_md;
        const double nb = sqrt(bxd * bxd + byd * byd);
        if (nb > (double)vd) { const double sc = (double)vd / nb; bxd *= sc; byd *= sc; }
        bvx = (float)bxd; bvy = (float)byd;
        om = fmaf(torque, dt_inertia, om);
        float s2, c2;
        sincos_fast(th, s2, c2);
        vx = (float)((double)c2 * bxd - (double)s2 * byd);
        vy = (float)((double)s2 * bxd + (double)c2 * byd);
    } else {
        double vxd = (double)vx + fixd * dt_md, vyd = (double)vy + fiyd * dt_md;
        const double nb = sqrt(vxd * vxd + vyd * vyd);
        if (nb > (double)vd) { const double sc = (double)vd / nb; vxd *= sc; vyd *= sc; }
        vx = (float)vxd; vy = (float)vyd;
    }
    o[0] = px; o[fs] = py; o[2 * fs] = th; o[3 * fs] = vx; o[4 * fs] = vy; o[5 * fs] = bvx; o[6 * fs] = bvy; o[7 * fs] = om;
    o[8 * fs] = r; o[9 * fs] = m; o[10 * fs] = gx; o[11 * fs] = gy; o[12 * fs] = vd;
    if (a.mutate) a.in_v[(long)w * rows + i] = make_float2(in_vx, in_vy);
}

// the reference's in-place writes on the INPUT rows of an out-of-place update (refreshed linear velocity of headed models :256,
// goal columns :231-234), applied after the step: other lanes were still reading those rows during it
__global__ void k_bw_mutate(const GArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (i >= a.n) return;
    float* si = a.Smut + ((long)w * a.rows + i) * a.as;
    const float* so = a.Sout + ((long)w * a.rows + i) * a.as;
    if (a.type >= 3) { const float2 v = a.in_v[(long)w * a.rows + i]; si[3 * a.fs] = v.x; si[4 * a.fs] = v.y; }
    si[10 * a.fs] = so[10 * a.fs]; si[11 * a.fs] = so[11 * a.fs];
}

// cs_peek: [n][8] rows x, y, yaw, Vx, Vy, Omega, Gx, Gy of the stepped rows (motion_model_manager.py:691-709, :294-298)
__global__ void k_bw_peek_rows(const GArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (i >= a.n) return;
    const float* s = a.Sout + ((long)w * a.rows + i) * a.as;
    float* o = a.peek_out + ((long)w * a.n + i) * 8;
    const float2 g = a.peek_goal[(long)w * a.n + i];
    o[0] = s[0]; o[1] = s[a.fs]; o[2] = s[2 * a.fs]; o[3] = s[3 * a.fs]; o[4] = s[4 * a.fs]; o[5] = s[7 * a.fs]; o[6] = g.x; o[7] = g.y;
}

// cs_step_trace on the grid path: rows [first, end) of every world as they stand in S, in write_trace's record layout (stepcommon.h),
// into record `rec` = d_trace + sub * W * rows * 12.  A human's record carries the head of its goal list; the robot's its goal columns.
__global__ void k_bw_trace(const GArgs a, const float* S, float* rec, int first, int end)
{
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (i >= end) return;
    const float* s = S + ((long)w * a.rows + i) * a.as;
    const long fs = a.fs;
    float g0x = s[10 * fs], g0y = s[11 * fs];
    if (i < a.n) { const float* gi = a.goals + ((long)w * a.n + i) * a.G * 2; g0x = gi[0]; g0y = gi[1]; }
    write_trace(rec + ((long)w * a.rows + i) * 12, s[0], s[fs], s[2 * fs], s[3 * fs], s[4 * fs], s[5 * fs], s[6 * fs], s[7 * fs], s[10 * fs], s[11 * fs], g0x, g0y);
}

// The robot handed over through cs_worlds.d_robot: robot.step(action, dt) (robot_agent.py:114-136) and states[-1] = robot row
// (motion_model_manager.py:359) before a substep; one lane per world.  S: the rows the substep is about to read.
__global__ void k_bw_robot(const GArgs a, float* S)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.W) return;
    float* rb = a.robot + (long)w * 13;
    if (a.action != nullptr) {
        const float ax = a.action[(long)w * 2], ay = a.action[(long)w * 2 + 1];
        if (a.flags & CS_ROBOT_UNICYCLE) {
            float c, sn;
            sincos_fast(rb[2] + ay, sn, c);
            rb[0] += c * ax * a.dt; rb[1] += sn * ax * a.dt;
            const float th = mod_two_pi(rb[2] + ay);
            rb[2] = th;
            sincos_fast(th, sn, c);
            rb[3] = c * ax; rb[4] = sn * ax;
        } else {
            rb[0] += ax * a.dt; rb[1] += ay * a.dt; rb[3] = ax; rb[4] = ay;
        }
    }
    float* s = S + ((long)w * a.rows + a.n) * a.as;
    for (int f = 0; f < 13; ++f) s[f * a.fs] = rb[f];
}

// The parallel-traffic respawn rule (motion_model_manager.py:407-422) on the stepped rows of worlds beyond one block: one block per
// world.  The reference respawns the flagged humans (|p - goals[0]| < 3) of a world in index order, each behind everybody else:
// x_0 = max(max_x + 2 max_r, bound) and the c-th flagged one (c lower-indexed flagged humans in its world) lands at
// x_c = max(x_{c-1} + 2 max_r, bound) because x_{c-1} is then the rightmost human (the rule of the crowd kernels, sfmstep_kernel.h /
// orca.hip).  orca = 0: SFM / HSFM rows (max_r over radius + safety space; the reference also writes the goal into state columns
// 6:8, :421); orca = 1: RVO2 agents (max_r over the plain radii -- safety_space is 0 for ORCA humans, :154-158 --, the goal goes to
// the state's goal columns, the preferred velocity of the step stays); orca = 2: the reference's NON-parallel path (agent objects, what
// its RK45 integration runs on: radius + safety space, position and goal list only -- the column 6:8 write is `if self.parallel`).
struct RespawnArgs { int W, n, rows, G, orca; float* S; long as, fs; float* goals; const float* extra; float bx, by; const int* world_flags; };

__global__ __launch_bounds__(256) void k_bw_respawn(const RespawnArgs a)
{
    __shared__ float red[2][256];
    __shared__ int wcnt[4];
    __shared__ int carry_s;
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6, n = a.n, rows = a.rows;
    if (a.world_flags != nullptr && !(a.world_flags[w] & 1)) return;
    float* Sw = a.S + (long)w * rows * a.as;
    const long fs = a.fs;
    float mx = -INFINITY, mr = 0.0f;
    for (int i = t; i < rows; i += 256) {          // consider_robot: the robot row takes part in both maxima
        const float* s = Sw + (long)i * a.as;
        mx = fmaxf(mx, s[0]);
        mr = fmaxf(mr, a.orca == 1 ? s[8 * fs] : s[8 * fs] + a.extra[(long)w * rows + i]);
    }
    red[0][t] = mx; red[1][t] = mr;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) { red[0][t] = fmaxf(red[0][t], red[0][t + off]); red[1][t] = fmaxf(red[1][t], red[1][t + off]); }
        __syncthreads();
    }
    mx = red[0][0]; mr = red[1][0];
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + t;
        float* s = Sw + (long)(i < n ? i : 0) * a.as;
        float* gi = a.goals + ((long)w * n + (i < n ? i : 0)) * a.G * 2;
        const float px = s[0], py = s[fs], g0x = gi[0], g0y = gi[1];
        const float rdx = px - g0x, rdy = py - g0y;
        const bool flag = i < n && (a.orca ? sqrtf(rdx * rdx + rdy * rdy) < 3.0f : fmaf(rdx, rdx, rdy * rdy) < 9.0f);
        const unsigned long long fm = __builtin_amdgcn_ballot_w64(flag);
        if (lane == 0) wcnt[wv] = __builtin_popcountll(fm);
        __syncthreads();
        int c = carry_s + __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
        for (int k = 0; k < wv; ++k) c += wcnt[k];
        if (flag) {
            const float x = csimpl::respawn_x(mx, mr, a.bx, c);   // (respawnx.h: the reference's float64 sum, rounded once)
            const float ny = (py >= 0.0f) ? fminf(py, a.by) : fmaxf(py, -a.by);
            s[0] = x; s[fs] = ny;
            if (a.orca) { s[10 * fs] = g0x; s[11 * fs] = ny; }
            else { s[6 * fs] = g0x; s[7 * fs] = ny; }                  // states[i,6:8] = goal (the reference writes columns 6:8, :421)
            for (int g = 0; g < a.G; ++g) { gi[2 * g] = g0x; gi[2 * g + 1] = ny; }   // human.set_goals([[goals[0][0], position[1]]]) :418, :422
        }
        __syncthreads();
        if (t == 0) carry_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
}

using gfn = void (*)(const GArgs);

template <bool PEQ>
gfn pick_big(int type)
{
    switch (type) {
        case 0: return (gfn)k_bw_sfm_step<0, 0, PEQ>; case 1: return (gfn)k_bw_sfm_step<1, 0, PEQ>; case 2: return (gfn)k_bw_sfm_step<2, 0, PEQ>;
        case 3: return (gfn)k_bw_sfm_step<0, 1, PEQ>; case 4: return (gfn)k_bw_sfm_step<1, 1, PEQ>; case 5: return (gfn)k_bw_sfm_step<2, 1, PEQ>;
        case 6: return (gfn)k_bw_sfm_step<0, 2, PEQ>; case 7: return (gfn)k_bw_sfm_step<1, 2, PEQ>; case 8: return (gfn)k_bw_sfm_step<2, 2, PEQ>;
    }
    return nullptr;
}

} // namespace

namespace csimpl {

size_t grid_bytes(int W, int rows, int NB)
{
    const size_t total = (size_t)W * rows, M = (size_t)W * NB;
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return pad(total * sizeof(int2)) + pad(total * sizeof(unsigned)) + 2 * pad(total * sizeof(int)) + pad((M + 2) * sizeof(int)) + pad((M + 1) * sizeof(int)) +
           pad(((M + 1023) / 1024 + 1) * sizeof(int)) + 1024;
}

int grid_build(const float* S, long as, long fs, int W, int rows, int NB, const float* d_inv_cell, float inv_cell, void* mem, GridView& g, hipStream_t stream)
{
    const size_t total = (size_t)W * rows, M = (size_t)W * NB;
    if ((long long)W * NB >= (1ll << 31) || total >= (1ull << 31)) return fail(CS_ERR_ARG, "worlds beyond one block: W x buckets (or W x rows) does not fit the 32-bit grid keys");
    char* p = (char*)mem;
    auto take = [&](size_t bytes) { char* q = p; p += (bytes + 255) & ~(size_t)255; return q; };
    g.W = W; g.rows = rows; g.NB = NB;
    g.cellxy = (int2*)take(total * sizeof(int2));
    g.keys = (unsigned*)take(total * sizeof(unsigned));
    g.tmp = (int*)take(total * sizeof(int));
    g.sorted = (int*)take(total * sizeof(int));
    g.start = (int*)take((M + 2) * sizeof(int));
    g.fill = (int*)take((M + 1) * sizeof(int));
    const int nblk = (int)((M + 1023) / 1024);
    int* bsum = (int*)take((size_t)(nblk + 1) * sizeof(int));
    const dim3 rgrid((rows + 255) / 256, W);
    HIP_TRY(hipMemsetAsync(g.fill, 0, (M + 1) * sizeof(int), stream));
    hipLaunchKernelGGL(k_grid_keys, rgrid, dim3(256), 0, stream, g, S, as, fs, d_inv_cell, inv_cell);
    hipLaunchKernelGGL(k_scan_blocks, dim3(nblk), dim3(256), 0, stream, g.fill, g.start, bsum, (int)M);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, stream, bsum, nblk);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)((M + 1 + 255) / 256)), dim3(256), 0, stream, g.start, bsum, (int)M, (int)total);
    HIP_TRY(hipMemsetAsync(g.fill, 0, (M + 1) * sizeof(int), stream));
    hipLaunchKernelGGL(k_grid_scatter, rgrid, dim3(256), 0, stream, g);
    hipLaunchKernelGGL(k_grid_rank, rgrid, dim3(256), 0, stream, g);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

void big_respawn_launch(float* S, long as, long fs, int W, int n, int rows, float* goals, int G, const float* extra, int orca, float bx, float by,
                        const int* world_flags, hipStream_t stream)
{
    RespawnArgs r{W, n, rows, G, orca, S, as, fs, goals, extra, bx, by, world_flags};
    hipLaunchKernelGGL(k_bw_respawn, dim3(W), dim3(256), 0, stream, r);
}

int big_world_buckets(int rows)
{
    int NB = 1024;
    while (NB < 2 * rows && NB < (1 << 20)) NB <<= 1;
    return NB;
}

size_t sfm_big_scratch_bytes(const cs_worlds* w)
{
    const int W = w->W, rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    const size_t state_bytes = ((size_t)W * rows * 13 * sizeof(float) + 255) & ~(size_t)255;
    const size_t misc = (((size_t)W * sizeof(float) + 255) & ~(size_t)255) + 2 * ((((size_t)W * rows * sizeof(float2)) + 255) & ~(size_t)255) + 256;
    return 2 * state_bytes + misc + grid_bytes(W, rows, big_world_buckets(rows));
}

// n_substeps Euler substeps of worlds beyond one block.  d_out: where the result goes (w->d_state for cs_step / the in-place
// update); mutate_input: reproduce the reference's in-place writes on w->d_state (out-of-place cs_update_humans_parallel).
int sfm_big_launch(const cs_worlds* w, float dt, int n_substeps, float* d_out, int mutate_input, bool robot_from_array, const float* d_action,
                   float* d_peek, hipStream_t stream, float* d_trace)
{
    const int W = w->W, n = w->n, rows = n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    if (robot_from_array && !(w->flags & CS_ROBOT_ROW)) robot_from_array = false;
    const bool robot_moves = d_action != nullptr && w->d_robot != nullptr;
    const int NB = big_world_buckets(rows);
    const size_t state_bytes = ((size_t)W * rows * 13 * sizeof(float) + 255) & ~(size_t)255;
    const size_t pad_w = (((size_t)W * sizeof(float) + 255) & ~(size_t)255), pad_v = (((size_t)W * rows * sizeof(float2) + 255) & ~(size_t)255);
    char* base = nullptr;
    {
        const int rcs = scratch((void**)&base, sfm_big_scratch_bytes(w), SCRATCH_SFM_BIG, stream);
        if (rcs) return rcs;
    }
    GArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = W; a.n = n; a.rows = rows; a.G = w->G; a.O = w->O; a.Smax = w->Smax; a.NB = NB; a.type = w->type; a.flags = w->flags; a.dt = dt;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; } else { a.as = 1; a.fs = (long)W * rows; }
    a.goals = w->d_goals; a.params = w->d_params; a.safety = w->d_safety; a.obstacles = w->d_obstacles;
    a.bx = w->respawn_bound_x; a.by = w->respawn_bound_y; a.world_flags = w->d_world_flags; a.robot = w->d_robot; a.action = d_action;
    float* SA = (float*)base;
    float* SB = (float*)(base + state_bytes);
    a.inv_cell = (float*)(base + 2 * state_bytes);
    a.in_v = (float2*)(base + 2 * state_bytes + pad_w);
    a.peek_goal = (float2*)(base + 2 * state_bytes + pad_w + pad_v);
    a.peek = d_peek ? 1 : 0; a.peek_out = d_peek;
    void* grid_mem = base + 2 * state_bytes + pad_w + 2 * pad_v + 256;
    const bool peq = (w->flags & CS_ALL_PARAMS_EQUAL) != 0;
    const gfn step = peq ? pick_big<true>(w->type) : pick_big<false>(w->type);
    if (!step) return fail(CS_ERR_TYPE, "Type " + std::to_string(w->type) + " does not exist for this implementation");
    if (d_peek) {
        // one Euler step of size dt, nothing committed (motion_model_manager.py:691-709): the rows go to a scratch buffer, the goal
        // lists stay, the robot row is the one handed over (it does not move)
        float* in = w->d_state;
        if (robot_from_array) {   // the robot row of d_robot without touching the caller's rows: step from a copy
            HIP_TRY(hipMemcpyAsync(SB, w->d_state, (size_t)W * rows * 13 * sizeof(float), hipMemcpyDeviceToDevice, stream));
            a.action = nullptr;
            hipLaunchKernelGGL(k_bw_robot, dim3((W + 63) / 64), dim3(64), 0, stream, a, SB);
            in = SB;
        }
        a.Sin = in; a.Sout = SA;
        hipLaunchKernelGGL(k_bw_reach, dim3(W), dim3(256), 0, stream, a);
        GridView g;
        const int rcg = grid_build(in, a.as, a.fs, W, rows, NB, a.inv_cell, 0.0f, grid_mem, g, stream);
        if (rcg) return rcg;
        a.cellxy = g.cellxy; a.start = g.start; a.sorted = g.sorted;
        hipLaunchKernelGGL(step, dim3((rows + 255) / 256, W), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(k_bw_peek_rows, dim3((n + 255) / 256, W), dim3(256), 0, stream, a);
        HIP_TRY(hipGetLastError());
        return CS_OK;
    }
    float* cur = w->d_state;
    for (int sub = 0; sub < n_substeps; ++sub) {
        const bool last = sub + 1 == n_substeps;
        float* nxt = last ? d_out : ((sub & 1) ? SB : SA);
        if (nxt == cur) nxt = (cur == SA) ? SB : SA;      // (in-place single substep: through a scratch buffer, copied back below)
        a.Sin = cur; a.Sout = nxt;
        a.mutate = (mutate_input && sub == 0) ? 1 : 0;
        a.Smut = w->d_state;
        // robot.step(action, dt) ; states[-1] = robot row, before update_humans (social_nav_gym.py:240-243, motion_model_manager.py:359)
        if (robot_from_array || robot_moves) {
            a.action = robot_moves ? d_action : nullptr;
            hipLaunchKernelGGL(k_bw_robot, dim3((W + 63) / 64), dim3(64), 0, stream, a, cur);
        }
        // cs_step_trace: the robot's record k is the robot as substep k + 1 sees it (it moves before the humans)
        if (d_trace && rows > n && sub > 0)
            hipLaunchKernelGGL(k_bw_trace, dim3(1, W), dim3(64), 0, stream, a, cur, d_trace + (size_t)(sub - 1) * W * rows * 12, n, rows);
        if (sub == 0) hipLaunchKernelGGL(k_bw_reach, dim3(W), dim3(256), 0, stream, a);   // radii, parameters and speed limits hold for the launch
        GridView g;
        int rcg = grid_build(cur, a.as, a.fs, W, rows, NB, a.inv_cell, 0.0f, grid_mem, g, stream);
        if (rcg) return rcg;
        a.cellxy = g.cellxy; a.start = g.start; a.sorted = g.sorted;
        hipLaunchKernelGGL(step, dim3((rows + 255) / 256, W), dim3(256), 0, stream, a);
        if (a.mutate) hipLaunchKernelGGL(k_bw_mutate, dim3((n + 255) / 256, W), dim3(256), 0, stream, a);
        if (w->flags & CS_RESPAWN) big_respawn_launch(nxt, a.as, a.fs, W, n, rows, w->d_goals, w->G, w->d_safety, 0, a.bx, a.by, w->d_world_flags, stream);
        if (d_trace)   // every human's row after this substep (respawn rule included): what the LDS kernels record from their registers
            hipLaunchKernelGGL(k_bw_trace, dim3((n + 255) / 256, W), dim3(256), 0, stream, a, nxt, d_trace + (size_t)sub * W * rows * 12, 0, n);
        cur = nxt;
    }
    if (d_trace && rows > n && n_substeps > 0)   // ... and the robot as it stands at the end
        hipLaunchKernelGGL(k_bw_trace, dim3(1, W), dim3(64), 0, stream, a, cur, d_trace + (size_t)(n_substeps - 1) * W * rows * 12, n, rows);
    HIP_TRY(hipGetLastError());
    if (cur != d_out) HIP_TRY(hipMemcpyAsync(d_out, cur, (size_t)W * rows * 13 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return CS_OK;
}

} // namespace csimpl
