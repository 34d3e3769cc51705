// lookahead.hip -- SURVEY.md §8 row f1: the robot's one-step look-ahead over its action set, batched.
// Replaces compute_rotated_states_and_reward + transform_state_to_agent_centric
//   /root/reference/crowd_nav/policy/cadrl.py:42-83, :13-39
// (called once per robot decision by CADRL / SARL / LSTM-RL: crowd_nav/policy/multi_human_rl.py:46, cadrl.py:262),
// for W robots (worlds) at once: the [W][A][n][13|15] value-network input stays on the GPU for the learner.
//
// One block per world.  Phase 1: one lane per action runs the swept collision test over the humans (sequential, with
// the reference's early break) and the agent-centric frame of that action into LDS.  Phase 2: one lane per
// (action, human) element writes a 13- or 15-float row.  The output is the HBM cost: 4 * 13 * A * n bytes per world
// (81 actions x 25 humans: 105 KB), written once; inputs are ~1 KB per world.
#include <hip/hip_runtime.h>

#include <cmath>

#include "common.h"

namespace {

using csimpl::fail;

__global__ __launch_bounds__(256) void k_lookahead(int W, int n, int A, int headed, const float* actions, const float* next,
                                                   const float* cur, const float* robot, int rstride, float dt,
                                                   float* rotated, float* rewards)
{
    extern __shared__ float lds[]; // [A][8]: ax, ay, nrx, nry, cos, sin, dg, -
    const int w = blockIdx.x;
    const int nc = headed ? 6 : 4, cc = headed ? 7 : 5, oc = headed ? 15 : 13;
    const float* rb = robot + (long)w * rstride;
    const float rpx = rb[0], rpy = rb[1], rr = rb[4], rgx = rb[5], rgy = rb[6], rvd = rb[7];
    const float* curw = cur + (long)w * n * cc;
    const float* nxtw = next + (long)w * n * nc;
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        const float ax = actions[2 * a], ay = actions[2 * a + 1];
        const float nrx = rpx + ax * dt, nry = rpy + ay * dt;
        float dmin = 9223372036854775807.0f;
        bool collision = false;
        for (int j = 0; j < n; ++j) { // cadrl.py:56-64, utils.py:22-36
            const float* c = curw + (long)j * cc;
            const float x1 = c[0] - rpx, y1 = c[1] - rpy;
            const float x2 = x1 + (c[2] - ax) * dt, y2 = y1 + (c[3] - ay) * dt;
            const float px = x2 - x1, py = y2 - y1;
            float d;
            if (px == 0.0f && py == 0.0f) d = sqrtf(x1 * x1 + y1 * y1);
            else {
                float u = ((0.0f - x1) * px + (0.0f - y1) * py) / (px * px + py * py);
                u = u > 1.0f ? 1.0f : (u < 0.0f ? 0.0f : u);
                const float qx = x1 + u * px, qy = y1 + u * py;
                d = sqrtf(qx * qx + qy * qy);
            }
            const float dist = d - c[4] - rr;
            if (dist < 0.0f) { collision = true; break; }
            else if (dist < dmin) dmin = dist;
        }
        const float gdx = rgx - nrx, gdy = rgy - nry;
        const float dg = sqrtf(gdx * gdx + gdy * gdy);
        float rew = 0.0f;                      // literals of cadrl.py:69-72
        if (collision) rew = -0.25f;
        else if (dg < rr) rew = 1.0f;
        else if (dmin < 0.2f) rew = (dmin - 0.2f) * 0.5f * dt;
        rewards[(long)w * A + a] = rew;
        const float rot = atan2f(gdy, gdx);    // x axis: next robot position -> goal (:22)
        float* s = lds + 8 * a;
        s[0] = ax; s[1] = ay; s[2] = nrx; s[3] = nry; s[4] = cosf(rot); s[5] = sinf(rot); s[6] = dg;
    }
    __syncthreads();
    float* outw = rotated + (long)w * A * n * oc;
    for (int idx = threadIdx.x; idx < A * n; idx += blockDim.x) {
        const int a = idx / n, j = idx - a * n;
        const float* s = lds + 8 * a;
        const float ax = s[0], ay = s[1], nrx = s[2], nry = s[3], cr = s[4], sr = s[5];
        const float* q = nxtw + (long)j * nc;
        const float hx = q[0] - nrx, hy = q[1] - nry;
        const float hvx = headed ? q[3] : q[2], hvy = headed ? q[4] : q[3];
        const float hr = curw[(long)j * cc + 4];
        float* o = outw + (long)idx * oc;
        o[0] = s[6]; o[1] = rvd; o[2] = 0.0f; o[3] = rr;
        o[4] = ax * cr + ay * sr; o[5] = ay * cr - ax * sr;
        o[6] = hx * cr + hy * sr; o[7] = hy * cr - hx * sr;
        o[8] = hvx * cr + hvy * sr; o[9] = hvy * cr - hvx * sr;
        o[10] = hr; o[11] = sqrtf(hx * hx + hy * hy); o[12] = rr + hr;
        if (headed) { o[13] = q[2] - 0.0f; o[14] = q[5]; }
    }
}

} // namespace

extern "C" int cs_lookahead(int W, int n, int A, int theta_and_omega_visible, const float* d_actions, const float* d_next,
                            const float* d_current, const float* d_robot, int robot_stride, float dt, float* d_rotated,
                            float* d_rewards, void* stream)
{
    if (W <= 0 || n <= 0 || A <= 0) return fail(CS_ERR_ARG, "W, n, A must be positive");
    if (!d_actions || !d_next || !d_current || !d_robot || !d_rotated || !d_rewards) return fail(CS_ERR_ARG, "null argument");
    if (robot_stride < 8) return fail(CS_ERR_ARG, "robot rows need at least 8 columns: px,py,vx,vy,r,gx,gy,v_pref");
    const size_t shmem = (size_t)A * 8 * sizeof(float);
    if (shmem > 60 * 1024) return fail(CS_ERR_ARG, "action set too large");
    hipLaunchKernelGGL(k_lookahead, dim3(W), dim3(256), shmem, (hipStream_t)stream, W, n, A, theta_and_omega_visible ? 1 : 0,
                       d_actions, d_next, d_current, d_robot, robot_stride, dt, d_rotated, d_rewards);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

// ---- HIP graph helpers (include/crowdstep.h) --------------------------------------------------------------
extern "C" int cs_graph_begin_capture(void* stream)
{
    if (!stream) return fail(CS_ERR_ARG, "graph capture needs a non-default stream (cs_stream_create)");
    HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return CS_OK;
}

extern "C" int cs_graph_end_capture(void* stream, void** graph_exec)
{
    if (!stream || !graph_exec) return fail(CS_ERR_ARG, "null argument");
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(CS_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    *graph_exec = exec;
    return CS_OK;
}

extern "C" int cs_graph_launch(void* graph_exec, void* stream)
{
    if (!graph_exec) return fail(CS_ERR_ARG, "null graph");
    HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return CS_OK;
}

extern "C" int cs_graph_destroy(void* graph_exec)
{
    if (graph_exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return CS_OK;
}
