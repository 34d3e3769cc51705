// rowstep.hip -- the SFM / HSFM step for SMALL worlds (2 .. 11 rows): one world per 16-lane DPP row, four worlds per
// wavefront, partners exchanged with DPP row shifts instead of LDS.
//
// Same reference path as crowdstep.hip (update_humans_parallel, /root/reference/social_gym/src/forces_parallel.py:185-284,
// pair forces evaluated once :87-133, substep loop social_gym/social_nav_gym.py:240-245, respawn
// social_gym/src/motion_model_manager.py:407-422) for the plain crowd batch: all_params_equal, no walls, no robot row,
// goal lists of <= 2 entries, state committed in place -- BASELINE.json configs[1] (4096 worlds x 10-agent SFM) and the
// reference's default environment (5 humans).
//
// Why a second kernel: with 10 rows the LDS kernel packs 6 worlds into a wavefront -> 683 wavefronts on 1024 SIMDs, every
// one alone on its SIMD, and a substep is a chain of LDS round trips (publish -> partner fetch -> reaction read-modify-write
// -> reaction sum: ~130 cycles each) that nothing hides: 1850 cycles per substep for 160 vector instructions.  Here lane r
// of a row holds agent r and lanes ROWS..15 mirror the positions of agents 0.. (one DPP move each per substep), so the
// partner at ring distance k is simply the lane k to the right (row_shl:k, foldable into the consuming subtract), and the
// reaction -f of a pair evaluated once comes back with two bound-checked row shifts (row_shr:k from the evaluator below,
// row_shl:ROWS-k from the one that wrapped around; lanes that hold no agent contribute exact zeros).  No LDS, no barrier, no
// memory traffic inside the substep loop; 4 worlds per wavefront = 1024 wavefronts at 4096 worlds, one per SIMD.
//
// gfx950 only: no portability macros, no CPU fallback.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cstring>
#include <type_traits>
#include <utility>

#include "common.h"
#include "crowdstep.h"
#include "stepcommon.h"

namespace {

using namespace cstep;
using csimpl::fail;

// DPP row shifts inside the 16-lane row of the calling lane.  shl<N>: lane l reads lane l + N; shr<N>: lane l reads lane
// l - N; a source outside the row gives 0 (BOUND) or leaves `old` in place.
template <int N, bool BOUND = true> __device__ __forceinline__ float shl(float x, float old = 0.0f)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), 0x100 + N, 0xF, 0xF, BOUND));
}
template <int N, bool BOUND = true> __device__ __forceinline__ float shr(float x, float old = 0.0f)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), 0x110 + N, 0xF, 0xF, BOUND));
}
template <int N> __device__ __forceinline__ float ror(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_max(float x)
{
    x = fmaxf(x, ror<8>(x)); x = fmaxf(x, ror<4>(x)); x = fmaxf(x, ror<2>(x));
    return fmaxf(x, ror<1>(x));
}

template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I + 1>{}), ...);   // 1 .. N
}

template <int SOC, int HEADED, int ROWS>
__global__ __launch_bounds__(256, 2) void k_sfm_step_row16(const KArgs a)   // (up to four independent wavefronts per workgroup: row16_launch)
{
    static_assert(ROWS >= 2 && ROWS <= 11, "ROWS + ROWS / 2 - 1 must stay inside the 16-lane row");
    constexpr int HF = (ROWS - 1) / 2;            // ring distances whose pairs are evaluated once (by the lower end)
    constexpr bool EVEN = (ROWS % 2) == 0;        // even worlds: plus the antipodal partner, evaluated by both ends
    constexpr int KMAX = EVEN ? ROWS / 2 : HF;
    const int tid = threadIdx.x, r = tid & 15;
    __builtin_amdgcn_s_setprio(1);   // (base priority 1: above the generator's wavefronts of a refill pass, sfmstep_kernel.h)
    const int w = blockIdx.x * (blockDim.x >> 4) + (tid >> 4);   // four worlds per wavefront, blockDim.x / 64 independent wavefronts per workgroup
    const bool inw = w < a.W;
    const bool human = inw && r < ROWS;
    const bool mirror = r >= ROWS;                // lanes ROWS .. 15 carry copies of agents 0 .. (positions, Moussaid: velocities)
    const float dt = a.dt;
    const long sidx = (long)(inw ? w : 0) * ROWS + (r < ROWS ? r : 0);

    // ---- load phase: every global load of the prologue is issued before the first result is used -- ONE memory round trip instead of
    // six dependent ones (row; pair parameters; my parameter row; the goal list slot by slot; respawn flag; action).  Straight-line on
    // purpose (sfmstep_kernel.h, load phase): lanes without an agent load row 0 of world 0 / of their world, an absent optional
    // array is replaced by a readable dummy address.
    const float* dummy = a.Sin;
    float px, py, th, vx, vy, bvx, bvy, om, rad, m, gx, gy, vd, safety;
    {
        const float* s = a.Sin + sidx * a.in_as;
        const long fs = a.in_fs;
        px = s[0]; py = s[fs]; th = s[2 * fs]; vx = s[3 * fs]; vy = s[4 * fs]; bvx = s[5 * fs];
        bvy = s[6 * fs]; om = s[7 * fs]; rad = s[8 * fs]; m = s[9 * fs]; gx = s[10 * fs]; gy = s[11 * fs];
        vd = s[12 * fs];
        safety = a.safety[sidx];
    }
    // parameters: P[0] of my world for the pair forces (all_params_equal, forces_parallel.py:220), my own row for the rest
    const long pw = (a.flags & CS_PARAMS_SHARED) ? 0 : (long)(inw ? w : 0) * ROWS * 20;
    const SocRaw sraw = load_socraw(a.params + pw);
    const float* P = a.params + pw + (long)(r < ROWS ? r : 0) * 20;
    const float P0 = P[0], P16 = P[16], P17 = P[17], P18 = P[18], P19 = P[19];
    float* gi = a.goals + sidx * a.G * 2;          // my goal list (a lane without an agent: one it never touches)
    float gl[4];
    gl[0] = gi[0]; gl[1] = gi[1];
    {
        const float* g2 = a.G >= 2 ? gi + 2 : gi;
        gl[2] = g2[0]; gl[3] = g2[1];
    }
    const bool has_wflags = a.world_flags != nullptr;
    const int wflag_raw = *(has_wflags ? a.world_flags + (inw ? w : 0) : reinterpret_cast<const int*>(dummy));
    const bool robot_moves = a.action != nullptr;   // only the invisible robot of the epilogue (no robot row in this build)
    float ax, ay;
    {
        const float* ap = robot_moves ? a.action + (long)(inw ? w : 0) * 2 : dummy;
        ax = ap[0]; ay = ap[1];
    }

    // ---- compute phase
    if (!human) { px = 0; py = 0; th = 0; vx = 0; vy = 0; bvx = 0; bvy = 0; om = 0; rad = 0; m = 1; gx = 0; gy = 0; vd = 0; safety = 0; }
    if (!(inw && robot_moves)) { ax = 0; ay = 0; }
    const SocP sp = make_socp(sraw);
    float m_tau = 0, ko = 0, kd = 0, alpha = 1, klam = 0, dt_m = 0, inv_alpha = 1, inertia = 1, dt_inertia = 0;
    float g0x = gx, g0y = gy, g1x = 0, g1y = 0;
    int gk = 0;
    bool gdirty = false;
    if (human) {
        m_tau = m / P0;
        ko = P16; kd = P17; alpha = P18; klam = P19;
        dt_m = dt / m;
        inv_alpha = 1.0f / alpha;
        inertia = 0.5f * m * rad * rad;
        dt_inertia = dt / inertia;
        g0x = gl[0]; g0y = gl[1];
        gk = a.G;
        for (int g = a.G - 1; g >= 2; --g)
            if (isnan(gi[2 * g]) || isnan(gi[2 * g + 1])) gk = g;
        if (a.G >= 2 && (isnan(gl[2]) || isnan(gl[3]))) gk = 1;
        if (isnan(gl[0]) || isnan(gl[1])) gk = 0;
        if (gk == 2) { g1x = gl[2]; g1y = gl[3]; }
    }
    const bool respawn_here = human && (!has_wflags || (wflag_raw & 1));

    // mirrored value: lanes ROWS .. 15 take the value of lane - ROWS
    auto ext = [&](float x) { const float up = shr<ROWS, true>(x); return mirror ? up : x; };
    // radius + safety never changes during a launch: the sums r_i + r_j + safety_i + safety_j per ring distance are kept
    const float my_rs = rad + safety;
    const float rs_e = ext(my_rs);
    float rsum[KMAX + 1];
    static_for([&](auto kt) { constexpr int k = decltype(kt)::value; rsum[k] = my_rs + shl<k>(rs_e); }, std::make_integer_sequence<int, KMAX>{});

    float cs = 1.0f, sn = 0.0f;
    if (HEADED > 0 && human) sincos_fast(th, sn, cs);

    for (int sub = 0; sub < a.nsub; ++sub) {
        // cs_step_trace: my row as the previous substep left it (the last one is written behind the loop)
        if (a.trace != nullptr && human && sub > 0)
            write_trace(a.trace + (((long)(sub - 1) * a.W + w) * ROWS + r) * 12, px, py, th, vx, vy, bvx, bvy, om, gx, gy, g0x, g0y);
        // -- goal switch, forces_parallel.py:226-234, predicated: lists of <= 2 goals rotate in registers
        {
            const float gdx = g0x - px, gdy = g0y - py;
            const bool hit = human && fmaf(gdx, gdx, gdy * gdy) <= rad * rad;
            const bool sw = hit && gk == 2;
            const float t0 = g0x, t1 = g0y;
            g0x = sw ? g1x : g0x; g0y = sw ? g1y : g0y;
            g1x = sw ? t0 : g1x; g1y = sw ? t1 : g1y;
            gdirty = gdirty || sw;
            gx = hit ? g0x : gx; gy = hit ? g0y : gy;
        }
        // -- social force, every unordered pair once (:87-133)
        const float xe = ext(px), ye = ext(py);
        float vxe = 0.0f, vye = 0.0f;
        if constexpr (SOC == 2) { vxe = ext(vx); vye = ext(vy); }
        float ex = 0.0f, ey = 0.0f, rx = 0.0f, ry = 0.0f, rdm = -1.0f;   // rdm: max over my partners of r_ij - d (masked by `human` once, behind the loop)
        static_for([&](auto kt) {
            constexpr int k = decltype(kt)::value;
            const float dx = px - shl<k>(xe), dy = py - shl<k>(ye);
            float fx, fy;
            if constexpr (SOC == 2) {
                pair_force_moussaid_once(sp, dx, dy, vx - shl<k>(vxe), vy - shl<k>(vye), rsum[k], fx, fy);
            } else {
                // [A e^{rd/B}] n + [C e^{rd/D}] t, in units of sign(A); the k1 / k2 contact parts are exact zeros unless
                // rd > 0 and are added by the contact pass below
                const float d2 = fmaf(dx, dx, dy * dy);
                const float inv = rsq_fast(d2);
                const float rd = fmaf(-d2, inv, rsum[k]);
                // lanes without an agent hand exact zeros to the row shifts: their magnitude is the zero (one select; the displacement is finite)
                const float ga = human ? exp2_fast(fmaf(rd, sp.cB, sp.lA)) * inv : 0.0f;
                fx = ga * dx; fy = ga * dy;
                if constexpr (SOC == 1) {
                    const float gc = human ? exp2_fast(fmaf(rd, sp.cD, sp.lC)) * (inv * sp.sAC) : 0.0f;
                    fx = fmaf(-gc, dy, fx); fy = fmaf(gc, dx, fy);
                }
                rdm = fmaxf(rdm, rd);
            }
            if constexpr (SOC == 2) { fx = human ? fx : 0.0f; fy = human ? fy : 0.0f; }   // (Moussaid: the pair force as a whole)
            // (the first partner starts the sums: `0 + x` is an instruction the compiler must keep -- it turns -0 into +0)
            if constexpr (k == 1) { ex = fx; ey = fy; } else { ex += fx; ey += fy; }
            if constexpr (k <= HF) {
                // the partner's share -f: from the evaluator k lanes below, or from the one that wrapped around the ring
                const float sx = shr<k>(fx) + shl<ROWS - k>(fx), sy = shr<k>(fy) + shl<ROWS - k>(fy);
                if constexpr (k == 1) { rx = sx; ry = sy; } else { rx += sx; ry += sy; }
            }
        }, std::make_integer_sequence<int, KMAX>{});
        const float rdmax = human ? rdm : -1.0f;
        float fsx = ex - rx, fsy = ey - ry;
        if constexpr (SOC != 2) {
            fsx *= sp.sA; fsy *= sp.sA;
            if (__builtin_amdgcn_ballot_w64(rdmax > 0.0f) != 0) {   // contact somewhere in this wavefront (rare)
                static_for([&](auto kt) {
                    constexpr int k = decltype(kt)::value;         // partner (r + k) mod ROWS, k = 1 .. ROWS - 1
                    const bool lo = r + k < ROWS;
                    auto ring = [&](float v) { const float u = shl<k>(v), d = shr<ROWS - k>(v); return lo ? u : d; };
                    const float qx = ring(px), qy = ring(py), qvx = ring(vx), qvy = ring(vy), qrs = ring(my_rs);
                    const float dx = px - qx, dy = py - qy;
                    const float d2 = fmaxf(fmaf(dx, dx, dy * dy), 1e-30f);
                    const float inv = rsq_fast(d2);
                    const float m0 = fmaxf(0.0f, (my_rs + qrs) - dist_refined(d2, inv));
                    const float nx = dx * inv, ny = dy * inv;
                    const float dv = (qvy - vy) * nx - (qvx - vx) * ny;     // (v_j - v_i) . t
                    const float fn = sp.k1 * m0, ft = (sp.k2 * m0) * dv;
                    fsx += fn * nx - ft * ny;
                    fsy += fn * ny + ft * nx;
                }, std::make_integer_sequence<int, ROWS - 1>{});
            }
        }
        // -- the per-agent part, :254-283
        const float c = cs, s = sn;
        float cvx = vx, cvy = vy, fdx = 0.0f, fdy = 0.0f, th_n = th, sn_n = sn, cs_n = cs, torque = 0.0f;
        if constexpr (HEADED > 0) {
            cvx = c * bvx + (-s) * bvy;
            cvy = s * bvx + c * bvy;
        }
        {   // desired force, :23-40
            const float dx = gx - px, dy = gy - py;
            const float d2 = fmaf(dx, dx, dy * dy);
            const float inv = rsq_fast(fmaxf(d2, 1e-30f));
            const float wx = m_tau * (dx * inv * vd - cvx), wy = m_tau * (dy * inv * vd - cvy);
            const bool far_ = d2 * inv > rad;
            fdx = far_ ? wx : 0.0f;
            fdy = far_ ? wy : 0.0f;
        }
        const float fix = fdx + fsx, fiy = fdy + fsy;
        float gfx = fix, gfy = fiy;
        if constexpr (HEADED > 0) {
            th_n = wrap_angle(fmaf(om, dt, th));
            sincos_fast(th_n, sn_n, cs_n);
            const float tfx = HEADED == 1 ? fdx : fix, tfy = HEADED == 1 ? fdy : fiy;   // Farina: torque on the desired force alone
            const float kf = klam * norm2(tfx, tfy);
            const float k_theta = inertia * kf;
            const float k_omega = inertia * (1.0f + alpha) * sqrt_fast(kf * inv_alpha);
            const float delta = atan2_fast(s * tfx - c * tfy, c * tfx + s * tfy);
            torque = -k_theta * delta - k_omega * om;
            gfx = fix * c + fiy * s;
            gfy = ko * (fsx * (-s) + fsy * c) - kd * bvy;
        }
        if (human) {
            // explicit Euler, :273-283 (the position uses the velocity stored in the incoming row)
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
        }
        // -- parallel-traffic respawn, motion_model_manager.py:407-422: the flagged humans of a world are respawned in index
        //    order, each behind everybody else -- the c-th flagged one lands at x_c = max(x_{c-1} + 2 max_r, bound)
        if (a.flags & CS_RESPAWN) {
            const float rdx = px - g0x, rdy = py - g0y;
            const bool flag = respawn_here && fmaf(rdx, rdx, rdy * rdy) < 9.0f;     // |p - g| < 3
            const unsigned long long fm = __builtin_amdgcn_ballot_w64(flag);
            if (fm != 0) {                                                         // all lanes enter: the row reductions need them
                const float mx = row_max(human ? px : -INFINITY);
                const float mr = row_max(human ? my_rs : 0.0f);
                const unsigned rb = (unsigned)(fm >> (tid & 48)) & 0xFFFFu;
                const int cnt_below = __builtin_popcount(rb & ((1u << r) - 1u));
                const float x = csimpl::respawn_x<false>(mx, mr, a.bx, cnt_below);   // (respawnx.h: the reference's float64 sum, rounded once)
                if (flag) {
                    px = x;
                    py = (py >= 0.0f) ? fminf(py, a.by) : fmaxf(py, -a.by);
                    g0y = py;               // human.set_goals([[goals[0][0], position[1]]])   :418
                    bvy = g0x; om = g0y;    // states[i,6:8] = goal  (reference writes cols 6:8) :421
                    for (int g = 0; g < a.G; ++g) { gi[2 * g] = g0x; gi[2 * g + 1] = g0y; } //   :422
                    gk = a.G; g1x = g0x; g1y = g0y;
                }
            }
        }
    }

    if (a.trace != nullptr && human && a.nsub > 0)
        write_trace(a.trace + (((long)(a.nsub - 1) * a.W + w) * ROWS + r) * 12, px, py, th, vx, vy, bvx, bvy, om, gx, gy, g0x, g0y);

    // ---- epilogue ---------------------------------------------------------------------------
    if (human && gdirty) { gi[0] = g0x; gi[1] = g0y; gi[2] = g1x; gi[3] = g1y; }
    if (human) {
        float* o = a.Sout + sidx * a.out_as;
        const long fs = a.out_fs;
        o[0] = px; o[fs] = py; o[2 * fs] = th; o[3 * fs] = vx; o[4 * fs] = vy; o[5 * fs] = bvx;
        o[6 * fs] = bvy; o[7 * fs] = om; o[10 * fs] = gx; o[11 * fs] = gy;
        if (a.obs != nullptr) {   // cs_step_observe: the Gym's observation of the stepped humans (sfmstep_kernel.h epilogue)
            float* ob = a.obs + ((long)w * ROWS + r) * a.obs_cols;
            ob[0] = px; ob[1] = py; ob[2] = vx; ob[3] = vy; ob[4] = rad;
            if (a.obs_cols == 7) { ob[5] = th; ob[6] = om; }
        }
    }
    // invisible robot: advanced by the lane of row 0 (it does not interact with the crowd)
    if (inw && r == 0 && robot_moves && a.robot != nullptr) {
        float* rb = a.robot + (long)w * 13;
        float qx = rb[0], qy = rb[1], qt = rb[2], qvx = rb[3], qvy = rb[4];
        for (int sub = 0; sub < a.nsub; ++sub) {
            if (a.flags & CS_ROBOT_UNICYCLE) {
                const float c = cosf(qt + ay), s = sinf(qt + ay);
                qx += c * ax * dt; qy += s * ax * dt;
                qt = mod_two_pi(qt + ay);
                qvx = cosf(qt) * ax; qvy = sinf(qt) * ax;
            } else {
                qx += ax * dt; qy += ay * dt; qvx = ax; qvy = ay;
            }
        }
        rb[0] = qx; rb[1] = qy; rb[2] = qt; rb[3] = qvx; rb[4] = qvy;
    }
}

using kfn = void (*)(const KArgs);

template <int ROWS>
kfn pick_row16(int type)
{
    switch (type) {
        case 0: return (kfn)k_sfm_step_row16<0, 0, ROWS>; case 1: return (kfn)k_sfm_step_row16<1, 0, ROWS>;
        case 2: return (kfn)k_sfm_step_row16<2, 0, ROWS>; case 3: return (kfn)k_sfm_step_row16<0, 1, ROWS>;
        case 4: return (kfn)k_sfm_step_row16<1, 1, ROWS>; case 5: return (kfn)k_sfm_step_row16<2, 1, ROWS>;
        case 6: return (kfn)k_sfm_step_row16<0, 2, ROWS>; case 7: return (kfn)k_sfm_step_row16<1, 2, ROWS>;
        case 8: return (kfn)k_sfm_step_row16<2, 2, ROWS>;
    }
    return nullptr;
}

} // namespace

namespace csimpl {

bool row16_supports(int rows) { return rows == 5 || rows == 10; }

// One world per 16-lane DPP row, four per wavefront; `a` as crowdstep.hip's launch_step fills it (lean conditions checked there).
int row16_launch(const cstep::KArgs& a, hipStream_t stream)
{
    kfn fn = a.rows == 10 ? pick_row16<10>(a.type) : (a.rows == 5 ? pick_row16<5>(a.type) : nullptr);
    if (!fn) return fail(CS_ERR_ARG, "no DPP-row build for this row count / type");
    // wavefronts share nothing (no LDS, no barrier): four to a workgroup -- fewer workgroups start and end sooner (cfg2 11.83 -> 11.28 us;
    // sfmstep_kernel.h; CROWDSTEP_WG_WAVES = 1 / 2 / 4 for A/B)
    static const int wgw = []{ const char* e = std::getenv("CROWDSTEP_WG_WAVES"); const int v = e ? std::atoi(e) : 4; return (v == 1 || v == 2 || v == 4) ? v : 4; }();
    const int wpg = 4 * wgw;                        // worlds per workgroup
    const int grid = (a.W + wpg - 1) / wpg;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * wgw), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

} // namespace csimpl
