// Shared host-side helpers of libcrowdstep.so (error reporting across the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "crowdstep.h"

namespace cstep { struct KArgs; }

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

// ORCA branch (orca.hip)
int orca_launch(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float* d_peek, hipStream_t stream);
// name of the k_orca_step build orca_launch would run for `w` (cs_step_variant)
int orca_variant(const cs_worlds* w, char* buf, size_t buflen);
// the robot's own ORCA model, one doStep of the robot per world (orca.hip; cs_robot_model_step with CS_ORCA)
int orca_robot_launch(const cs_worlds* w, float robot_margin, const float* d_human_margin, float dt, hipStream_t stream);
// library-owned device scratch (grow-only, one slot per use, per host thread): the double-buffered state and the neighbour grid of
// worlds beyond one block.  Returns nullptr (and sets the error) when hipMalloc fails.
void* scratch(size_t bytes, int slot);
int big_world_min_rows(int dflt);
// small worlds, one per 16-lane DPP row (rowstep.hip)
bool row16_supports(int rows);
int row16_launch(const cstep::KArgs& a, hipStream_t stream);
// social-momentum branch (social_momentum.hip)
int social_momentum_launch(const cs_worlds* w, float dt, int n_substeps, const float* d_action, float* d_peek, hipStream_t stream);

} // namespace csimpl

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return ::csimpl::fail(CS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));      \
    } while (0)
