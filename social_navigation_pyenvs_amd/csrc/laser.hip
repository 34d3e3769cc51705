// laser.hip -- batched laser range finder.
//
// Restates LaserSensor.get_laser_measurements  /root/reference/social_gym/src/sensors.py:51-66
//   sphere_ray_intersect   sensors.py:24-33   (disc hit; the sensor inside a disc sees nothing: t < 0 -> max_distance)
//   segment_ray_intersect  sensors.py:35-49   (one-sided: denominator <= 0 -> no hit, endpoints ordered as
//                                              Obstacle.segments stores them, obstacle.py:31-32)
// for W robots at once: one lane per (world, ray).  Humans come from the resident state rows (x, y, radius), walls from
// the same [O][Smax][2][2] NaN-padded array the step kernel reads.  Output-bound: 4 B per ray.
// gfx950 only.

#include <hip/hip_runtime.h>

#include <string>

#include "common.h"
#include "crowdstep.h"

namespace {

using csimpl::fail;

struct LArgs {
    int W, n, rows, O, Smax, samples, obstacles_shared;
    const float* S;
    long as, fs;
    const float* pose;      // [W][pose_stride]: x, y, yaw in columns 0, 1, 2
    int pose_stride;
    const float* obstacles;
    float range, max_distance;
    float* out;             // [W][samples]
};

__global__ __launch_bounds__(64) void k_laser_scan(const LArgs a)
{
    const int w = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.samples) return;
    const float* ps = a.pose + (long)w * a.pose_stride;
    const float x3 = ps[0], y3 = ps[1], yaw = ps[2];
    // np.linspace(yaw - range/2, yaw + range/2, samples): start + k * step, the last sample is the stop value itself
    const float start = yaw - a.range * 0.5f, stop = yaw + a.range * 0.5f;
    float ang = start;
    if (a.samples > 1) ang = (k == a.samples - 1) ? stop : start + (float)k * ((stop - start) / (float)(a.samples - 1));
    const float dx = cosf(ang), dy = sinf(ang);
    float m = a.max_distance;
    for (int i = 0; i < a.n; ++i) {
        const float* s = a.S + ((long)w * a.rows + i) * a.as;
        const float sx = x3 - s[0], sy = y3 - s[a.fs], r = s[8 * a.fs];
        const float b = sx * dx + sy * dy;
        const float c = sx * sx + sy * sy - r * r;
        const float h = b * b - c;
        if (h < 0.0f) continue;
        const float t = -b - sqrtf(h);
        if (t < 0.0f) continue;
        m = fminf(m, t);
    }
    if (a.O > 0) {
        const float* ob = a.obstacles + (a.obstacles_shared ? 0 : (long)w * a.O * a.Smax * 4);
        for (int sg = 0; sg < a.O * a.Smax; ++sg) {
            const float4 q = *reinterpret_cast<const float4*>(ob + (long)sg * 4);
            if (isnan(q.x)) continue;
            const float x1 = q.x, y1 = q.y, x2 = q.z, y2 = q.w;
            // (x3 - x4, y3 - y4) = -(dx, dy)
            const float den = (x1 - x2) * (-dy) - (y1 - y2) * (-dx);
            if (den <= 0.0f) continue;
            const float t = ((x1 - x3) * (-dy) - (y1 - y3) * (-dx)) / den;
            const float u = -((x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)) / den;
            if (t > 0.0f && t < 1.0f && u > 0.0f) {
                const float ix = x1 + t * (x2 - x1), iy = y1 + t * (y2 - y1);
                m = fminf(m, sqrtf((x3 - ix) * (x3 - ix) + (y3 - iy) * (y3 - iy)));
            }
        }
    }
    a.out[(long)w * a.samples + k] = m;
}

} // namespace

extern "C" int cs_laser_scan(const cs_worlds* w, const float* d_pose, int pose_stride, float range, int samples,
                             float max_distance, float* d_out, void* stream)
{
    if (!w || !d_out) return fail(CS_ERR_ARG, "null argument");
    if (w->W <= 0 || w->n < 0 || !w->d_state) return fail(CS_ERR_ARG, "bad cs_worlds");
    if (samples <= 0) return fail(CS_ERR_ARG, "samples must be positive");
    if (max_distance > 10.0f) return fail(CS_ERR_ARG, "Maxium distance for laser is 10 meters"); // sensors.py:13
    if (w->O < 0 || (w->O > 0 && (!w->d_obstacles || w->Smax <= 0))) return fail(CS_ERR_ARG, "bad obstacle description");
    LArgs a;
    a.W = w->W; a.n = w->n; a.rows = w->n + ((w->flags & CS_ROBOT_ROW) ? 1 : 0);
    a.O = w->O; a.Smax = w->Smax; a.samples = samples;
    a.obstacles_shared = (w->flags & CS_OBSTACLES_SHARED) ? 1 : 0;
    a.S = w->d_state;
    if (w->layout == CS_LAYOUT_AOS) { a.as = 13; a.fs = 1; }
    else if (w->layout == CS_LAYOUT_SOA) { a.as = 1; a.fs = (long)w->W * a.rows; }
    else return fail(CS_ERR_ARG, "bad layout");
    if (d_pose) { a.pose = d_pose; a.pose_stride = pose_stride > 0 ? pose_stride : 3; }
    else if (w->d_robot) { a.pose = w->d_robot; a.pose_stride = 13; } // robot safe-state rows: px, py, theta first
    else return fail(CS_ERR_ARG, "no sensor pose: d_pose and cs_worlds.d_robot are both null");
    a.obstacles = w->d_obstacles;
    a.range = range; a.max_distance = max_distance;
    a.out = d_out;
    const int block = 64;
    hipLaunchKernelGGL(k_laser_scan, dim3((samples + block - 1) / block, w->W), dim3(block), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}
