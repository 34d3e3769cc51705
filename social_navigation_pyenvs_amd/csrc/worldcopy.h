// worldcopy.h -- one world copied from a staging batch over the live one (state rows, goal lists, robot row, world flag, and the Gym's
// observation rows of the new humans): the body of cs_copy_worlds_masked* (gymstep.hip) and of cs_consume_staged_worlds (generate.hip).
// The reset it stands for: SocialNavGym.reset, /root/reference/social_gym/social_nav_gym.py:120-225.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace csimpl {

struct CopyArgs {
    int W, n, rows, G;
    const float* Ss; float* Sd; long as, fs;   // strides of the destination rows
    long sas, sfs;                              // ... and of the source rows (a staging batch may hold more worlds: other SoA plane size)
    const float* gs; float* gd;
    const float* rs; float* rd;
    const int* fsrc; int* fdst;
    const int* mask;
    const int* status;   // optional: cs_generate_worlds' per-world status; a world that could not be generated (non-zero) is NOT copied
    float* obs; int C;   // optional: the Gym's observation rows [W][n][C] of the copied worlds are rewritten from the new rows
};

// observation column c of SocialNavGym.compute_humans_observable_state: px, py, vx, vy, radius (, theta, omega)
__device__ __forceinline__ int obs_state_column(int c) { return c == 0 ? 0 : c == 1 ? 1 : c == 2 ? 3 : c == 3 ? 4 : c == 4 ? 8 : c == 5 ? 2 : 7; }

// lanes t = 0 .. 63 of the block that owns world w: source world ws -> destination world w
__device__ __forceinline__ void copy_world(const CopyArgs& a, long ws, int w, int t)
{
    for (int k = t; k < a.rows * 13; k += 64) {
        const int row = k / 13, f = k - row * 13;
        a.Sd[((long)w * a.rows + row) * a.as + f * a.fs] = a.Ss[(ws * a.rows + row) * a.sas + f * a.sfs];
    }
    const long gd0 = (long)w * a.n * a.G * 2, gs0 = ws * a.n * a.G * 2;
    for (int k = t; k < a.n * a.G * 2; k += 64) a.gd[gd0 + k] = a.gs[gs0 + k];
    if (a.rs && a.rd && t < 13) a.rd[(long)w * 13 + t] = a.rs[ws * 13 + t];
    if (a.fsrc && a.fdst && t == 0) a.fdst[w] = a.fsrc[ws];
    if (a.obs != nullptr) {   // same columns as k_gym_observe, read from the source rows
        for (int k = t; k < a.n * a.C; k += 64) {
            const int i = k / a.C, c = k - i * a.C;
            a.obs[((long)w * a.n + i) * a.C + c] = a.Ss[(ws * a.rows + i) * a.sas + obs_state_column(c) * a.sfs];
        }
    }
}

} // namespace csimpl
