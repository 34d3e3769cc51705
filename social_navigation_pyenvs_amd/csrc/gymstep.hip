// gymstep.hip -- the two small kernels that let a whole vectorised Gym step (reward -> bookkeeping -> substeps, with the
// regeneration of finished worlds running beside them -> copy-in -> observation) be ONE HIP graph of library launches:
//   cs_gym_observe        the observation list of SocialNavGym (compute_humans_observable_state, social_nav_gym.py:100-105;
//                         Agent.get_observable_state, agent.py:247-249) as an array gathered on the device
//   cs_copy_worlds_masked the worlds an auto-reset regenerated into a staging batch, copied over the finished ones
// gfx950 only.
#include <hip/hip_runtime.h>

#include "common.h"
#include "crowdstep.h"
#include "worldcopy.h"

namespace {

using csimpl::fail;

__global__ void k_gym_observe(int W, int n, int rows, int C, const float* S, long as, long fs, float* obs)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one lane per (world, human, column)
    if (k >= (long)W * n * C) return;
    const int c = (int)(k % C);
    const long wi = k / C;
    const int i = (int)(wi % n);
    const long w = wi / n;
    obs[k] = S[(w * rows + i) * as + csimpl::obs_state_column(c) * fs];   // px, py, vx, vy, radius (, theta, omega)
}

using csimpl::CopyArgs;

__global__ __launch_bounds__(64) void k_copy_worlds_masked(const CopyArgs a)
{
    const int w = blockIdx.x;
    if (!a.mask[w]) return;
    if (a.status != nullptr && a.status[w] != 0) return;   // a half-built world never replaces a live one
    csimpl::copy_world(a, w, w, threadIdx.x);
}

} // namespace

extern "C" {

int cs_gym_observe(const cs_worlds* w, int theta_and_omega_visible, float* d_obs, void* stream)
{
    if (!w || !d_obs || !w->d_state) return fail(CS_ERR_ARG, "null argument");
    if (w->W <= 0 || w->n <= 0) return fail(CS_ERR_ARG, "bad cs_worlds");
    if (w->layout != CS_LAYOUT_AOS && w->layout != CS_LAYOUT_SOA) return fail(CS_ERR_ARG, "bad layout");
    const int rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0), C = theta_and_omega_visible ? 7 : 5;
    const long as = w->layout == CS_LAYOUT_AOS ? 13 : 1, fs = w->layout == CS_LAYOUT_AOS ? 1 : (long)w->W * rows;
    const long total = (long)w->W * w->n * C;
    hipLaunchKernelGGL(k_gym_observe, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w->W, w->n, rows, C,
                       (const float*)w->d_state, as, fs, d_obs);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_copy_worlds_masked(const cs_worlds* src, const cs_worlds* dst, const int32_t* d_mask, void* stream)
{
    return cs_copy_worlds_masked_status(src, dst, d_mask, nullptr, stream);
}

int cs_copy_worlds_masked_status(const cs_worlds* src, const cs_worlds* dst, const int32_t* d_mask, const int32_t* d_status, void* stream)
{
    return cs_copy_worlds_masked_observe(src, dst, d_mask, d_status, 0, nullptr, stream);
}

int cs_copy_worlds_masked_observe(const cs_worlds* src, const cs_worlds* dst, const int32_t* d_mask, const int32_t* d_status,
                                  int theta_and_omega_visible, float* d_obs, void* stream)
{
    if (!src || !dst || !d_mask) return fail(CS_ERR_ARG, "null argument");
    if (src->W != dst->W || src->n != dst->n || src->G != dst->G || src->layout != dst->layout || ((src->flags ^ dst->flags) & CS_ROBOT_ROW))
        return fail(CS_ERR_ARG, "source and destination worlds differ in shape");
    if (!src->d_state || !dst->d_state || !src->d_goals || !dst->d_goals) return fail(CS_ERR_ARG, "null device buffer in cs_worlds");
    CopyArgs a;
    a.W = src->W; a.n = src->n; a.rows = src->n + ((src->flags & CS_ROBOT_ROW) ? 1 : 0); a.G = src->G;
    a.Ss = src->d_state; a.Sd = dst->d_state;
    a.as = src->layout == CS_LAYOUT_AOS ? 13 : 1; a.fs = src->layout == CS_LAYOUT_AOS ? 1 : (long)src->W * a.rows;
    a.sas = a.as; a.sfs = a.fs;
    a.gs = src->d_goals; a.gd = dst->d_goals; a.rs = src->d_robot; a.rd = dst->d_robot;
    a.fsrc = src->d_world_flags; a.fdst = const_cast<int*>(dst->d_world_flags); a.mask = d_mask; a.status = d_status;
    a.obs = d_obs; a.C = theta_and_omega_visible ? 7 : 5;
    hipLaunchKernelGGL(k_copy_worlds_masked, dim3(a.W), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

} // extern "C"
