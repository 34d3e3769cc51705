// worldcopy.h -- one world copied from a staging batch over the live one (state rows, goal lists, robot row, world flag, and the Gym's
// observation rows of the new humans): the body of cs_copy_worlds_masked* (gymstep.hip) and of cs_consume_staged_worlds (generate.hip).
// The reset it stands for: SocialNavGym.reset, /root/reference/social_gym/social_nav_gym.py:120-225.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

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

// element e of a world's copy list -- state rows, goal lists, robot row, world flag, observation rows, in that order -- as a (source,
// destination) pair of 32-bit words; {nullptr, nullptr} for an element of an array the batch does not have
struct CopyPair { const uint32_t* s; uint32_t* d; };
__device__ __forceinline__ CopyPair copy_element(const CopyArgs& a, long ws, int w, int e)
{
    auto U = [](const void* p) { return reinterpret_cast<const uint32_t*>(p); };
    auto V = [](void* p) { return reinterpret_cast<uint32_t*>(p); };
    int k = e;
    const int nS = a.rows * 13;
    if (k < nS) {
        const int row = k / 13, f = k - row * 13;
        return {U(a.Ss + (ws * a.rows + row) * a.sas + f * a.sfs), V(a.Sd + ((long)w * a.rows + row) * a.as + f * a.fs)};
    }
    k -= nS;
    const int nG = a.n * a.G * 2;
    if (k < nG) return {U(a.gs + ws * nG + k), V(a.gd + (long)w * nG + k)};
    k -= nG;
    if (k < 13) return (a.rs && a.rd) ? CopyPair{U(a.rs + ws * 13 + k), V(a.rd + (long)w * 13 + k)} : CopyPair{nullptr, nullptr};
    k -= 13;
    if (k < 1) return (a.fsrc && a.fdst) ? CopyPair{U(a.fsrc + ws), V(a.fdst + w)} : CopyPair{nullptr, nullptr};
    k -= 1;
    if (a.obs == nullptr) return {nullptr, nullptr};
    const int i = k / a.C, c = k - i * a.C;   // same columns as k_gym_observe, read from the source rows
    return {U(a.Ss + (ws * a.rows + i) * a.sas + obs_state_column(c) * a.sfs), V(a.obs + ((long)w * a.n + i) * a.C + c)};
}
__device__ __forceinline__ int copy_elements(const CopyArgs& a) { return a.rows * 13 + a.n * a.G * 2 + 13 + 1 + (a.obs != nullptr ? a.n * a.C : 0); }

// lanes t = 0 .. 63 of the block that owns world w: source world ws -> destination world w.  A lane requests its next 12 words before it
// stores the first (a 25-human world is 564 words: ONE batch, one memory round trip; until round 4 each array was a loop of
// load-wait-store trips, ten dependent round trips for such a world -- most of cs_consume_staged_worlds' 8 us)
//   COHERENT: the source was written by a kernel that may still be running on another stream (cs_refill_staged_worlds): its words are
//   read with device-scope relaxed atomic loads, which do not hit a stale line of this XCD's L2
//   st: optional generation status of the source world, requested together with the first batch; non-zero: nothing is stored.  Returns it.
template <bool COHERENT = false>
__device__ __forceinline__ int copy_world(const CopyArgs& a, long ws, int w, int t, const int* st = nullptr)
{
    constexpr int B = 12;
    const int E = copy_elements(a);
    int status = 0;
    for (int e0 = t; e0 - t < E; e0 += 64 * B) {   // (wave-uniform trip count)
        uint32_t v[B];
        uint32_t* d[B];
        int sv = 0;
        if (st != nullptr && e0 == t) sv = COHERENT ? __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *st;
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int e = e0 + 64 * u;
            CopyPair p = {nullptr, nullptr};
            if (e < E) p = copy_element(a, ws, w, e);
            d[u] = p.d;
            v[u] = p.s == nullptr ? 0u : (COHERENT ? __hip_atomic_load(p.s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p.s);
        }
        if (e0 == t) { status = sv; if (status != 0) return status; }
#pragma unroll
        for (int u = 0; u < B; ++u)
            if (d[u] != nullptr) *d[u] = v[u];
    }
    return status;
}

} // namespace csimpl
