// Shared host-side helpers of libcrowdstep.so (error reporting across the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "crowdstep.h"

namespace cstep { struct KArgs; struct GymFold; }

namespace csimpl {

extern thread_local std::string g_err;

inline int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#ifdef CS_STAMPS
extern unsigned long long* g_stamp_buf; // diagnostic build only (tools/stamp_probe.py)
#endif

// cs_gym_step_staged (crowdstep.hip): the staged-episode take-over's arguments for the step kernel's epilogue, built and checked where
// the pre-staged episodes live (generate.hip)
int stage_fold(const cs_generator* gen, const cs_worlds* staging, const cs_worlds* live, const cs_stage_book* book, int obs_cols, float* d_obs,
               cstep::GymFold& out);

// ORCA branch (orca.hip)
int orca_launch(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float* d_peek, hipStream_t stream);
// name of the k_orca_step build orca_launch would run for `w` (cs_step_variant)
int orca_variant(const cs_worlds* w, char* buf, size_t buflen);
// the robot's own ORCA model, one doStep of the robot per world (orca.hip; cs_robot_model_step with CS_ORCA)
int orca_robot_launch(const cs_worlds* w, float robot_margin, const float* d_human_margin, float dt, hipStream_t stream, int just_velocities = 0);
// library-owned device scratch, one block per (device, stream, use); never allocated, grown or freed while `stream` is capturing
// (CS_ERR_ARG then -- cs_reserve_scratch first), a block handed out during a capture is never freed before cs_release_scratch
enum { SCRATCH_ORCA_BIG = 0, SCRATCH_SFM_BIG = 1, SCRATCH_IMITATION = 2 };
int scratch(void** out, size_t bytes, int slot, hipStream_t stream);
int scratch_release_all();
inline size_t imitation_scratch_bytes(const cs_worlds* w, int n_substeps) { return (size_t)n_substeps * w->W * w->n * 4 * sizeof(float); }
size_t sfm_big_scratch_bytes(const cs_worlds* w);
size_t orca_big_scratch_bytes(const cs_worlds* w);
bool orca_uses_grid(const cs_worlds* w);   // ORCA worlds that take the grid path (more than 512 rows, or LDS columns beyond a block)
int big_world_min_rows(int dflt);
int device_simds();   // CUs x 4 of the current device
// the uniform grid of worlds beyond one block (bigworld.hip): rows binned into hashed buckets of square cells, every bucket's rows in
// index order (hand-written counting sort, bigworld.hip); start[w * NB + b] .. start[w * NB + b + 1] = bucket b of world w in `sorted` (row indices)
struct GridView { int W, rows, NB; int2* cellxy; unsigned* keys; int* tmp; int* sorted; int* start; int* fill; };
__host__ __device__ inline int cell_bucket(int cx, int cy, int NB) { return (int)(((unsigned)cx * 73856093u) ^ ((unsigned)cy * 19349663u)) & (NB - 1); }
size_t grid_bytes(int W, int rows, int NB);
// the parallel-traffic respawn rule on the stepped rows of worlds beyond one block (bigworld.hip k_bw_respawn); orca: RVO2 agent rows
void big_respawn_launch(float* S, long as, long fs, int W, int n, int rows, float* goals, int G, const float* extra, int orca, float bx, float by,
                        const int* world_flags, hipStream_t stream);
int big_world_buckets(int rows);   // hashed buckets of a world's grid: the power of two >= 2 * rows (1024 .. 2^20)
int grid_build(const float* S, long as, long fs, int W, int rows, int NB, const float* d_inv_cell, float inv_cell, void* mem, GridView& g, hipStream_t stream);
// SFM / HSFM worlds beyond one block (bigworld.hip)
int sfm_big_launch(const cs_worlds* w, float dt, int n_substeps, float* d_out, int mutate_input, bool robot_from_array, const float* d_action,
                   float* d_peek, hipStream_t stream, float* d_trace = nullptr);
// small worlds, one per 16-lane DPP row (rowstep.hip)
bool row16_supports(int rows);
int row16_launch(const cstep::KArgs& a, hipStream_t stream);
// the robot's n substeps against the crowd snapshots of an imitation block (robot_model.hip)
int robot_block_launch(const cs_worlds* w, int robot_type, const float* robot_params, float robot_margin, const float* d_human_margin,
                       float* d_robot_memory, float dt, int n_substeps, const float4* d_snap, hipStream_t stream);
// social-momentum branch (social_momentum.hip)
int social_momentum_launch(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float* d_peek, hipStream_t stream);

} // namespace csimpl

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return ::csimpl::fail(CS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));      \
    } while (0)
