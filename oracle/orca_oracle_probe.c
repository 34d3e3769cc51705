/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.   *** PARITY UNPINNED (like orca_oracle.c) ***
 *
 * The ORCA restatement of orca_oracle.c instantiated as a PROBE: the same statements in float32, with
 *   - every division, square root and two-term product sum (2 x 2 determinant, dot product, point + t * direction) perturbed by a
 *     bounded random rounding error: relative k * 2^-24 * u for divisions / roots, absolute k * 2^-24 * u * (the larger product)
 *     for the sums, u uniform in [-1, 1] -- what evaluating the operation with v_rcp / v_sqrt / v_rsq or as mul + fma instead of
 *     mul, mul, add changes (k = 0: the unperturbed restatement, bit for bit);
 *   - every DECISION of Agent::computeNeighbors / computeNewVelocity / linearProgram1-3 recorded in order (kind, outcome,
 *     relative margin of the comparison) for one chosen agent.
 * One purpose: classifying the agent-substeps on which the GPU's fast ORCA arithmetic is beyond 1e-5 of the exact restatement
 * (tests/orca_fast_parity.py): if the restatement's own answer moves that far when its operations are perturbed by <= k ulps, the
 * input sits on a decision edge (the first decision that flips is named) or on an ill-conditioned intersection (no decision flips),
 * and float32 does not determine the answer there.  It is not a reference.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t rng;
    float k;               /* noise amplitude in units of 2^-24 */
    int only_agent, active;
    int* kind; int* outcome; float* margin;
    int cap, n;
} probe_state;
static __thread probe_state PS;

static inline float p_unit(void)
{   /* xorshift64*, uniform in [-1, 1] */
    uint64_t x = PS.rng;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    PS.rng = x;
    const uint32_t r = (uint32_t)((x * 0x2545F4914F6CDD1DULL) >> 40);      /* 24 bits */
    return (float)r * (2.0f / 16777215.0f) - 1.0f;
}
static inline float p_rel(float v) { return (PS.active && PS.k != 0.0f) ? v * (1.0f + PS.k * 5.9604645e-8f * p_unit()) : v; }
static inline float p_sum(float t1, float t2, float v)
{
    if (!(PS.active && PS.k != 0.0f)) return v;
    const float m = fmaxf(fabsf(t1), fabsf(t2));
    return v + PS.k * 5.9604645e-8f * p_unit() * m;
}
static inline int p_decide(int kind, float a, float b, int outcome)
{
    if (PS.active && PS.kind && PS.n < PS.cap) {
        const float m = fmaxf(fabsf(a), fabsf(b));
        PS.kind[PS.n] = kind; PS.outcome[PS.n] = outcome; PS.margin[PS.n] = m > 0.0f ? fabsf(a - b) / m : 0.0f;
        ++PS.n;
    } else if (PS.active && PS.kind) ++PS.n;
    return outcome;
}

#define O_HOOKS 1
#define O_DIV(a, b) p_rel((a) / (b))
#define O_SQRT(x) p_rel(sqrtf(x))
#define O_DET(ax, ay, bx, by) p_sum((ax) * (by), (ay) * (bx), (ax) * (by) - (ay) * (bx))
#define O_DOT(ax, ay, bx, by) p_sum((ax) * (bx), (ay) * (by), (ax) * (bx) + (ay) * (by))
#define O_MAD(p, t, d) p_sum((p), (t) * (d), (p) + (t) * (d))
#define O_GT(kind, a, b) p_decide(kind, (a), (b), (a) > (b))
#define O_LT(kind, a, b) p_decide(kind, (a), (b), (a) < (b))
#define O_GE(kind, a, b) p_decide(kind, (a), (b), (a) >= (b))
#define O_LE(kind, a, b) p_decide(kind, (a), (b), (a) <= (b))
#define O_NOTE(kind, flag) p_decide(kind, (float)(flag), 0.5f, (flag))
#define O_AGENT_BEGIN(a) if (PS.only_agent >= 0 && (a) != PS.only_agent) continue; PS.active = 1

#define orc_orca_new_velocities_pa orcp_orca_new_velocities_pa
#define orc_orca_new_velocities_obst orcp_orca_new_velocities_obst
#define orc_orca_new_velocities orcp_orca_new_velocities
#define orc_orca_step_block_pa orcp_orca_step_block_pa
#define orc_orca_step_block_obst orcp_orca_step_block_obst
#define orc_orca_step_block orcp_orca_step_block
#define orc_orca_step_block_batched_pa orcp_orca_step_block_batched_pa
#define orc_orca_step_block_batched_obst orcp_orca_step_block_batched_obst
#define orc_orca_step_block_batched orcp_orca_step_block_batched
#include "orca_oracle.c"

/*
 * New velocity of ONE agent of a world (the arguments of orc_orca_new_velocities), `probes` times: probe 0 unperturbed (k = 0), probes
 * 1.. with noise amplitude k_ulps and seeds seed + p.  out_vel [probes][2]; the decision traces [probes][cap] (kind / outcome / margin)
 * and their lengths n_trace [probes] (a length above cap: truncated).
 */
void orcp_probe_agent(int na, const float* pos, const float* vel, const float* pref, const float* radius, const float* maxspeed,
                      float neighbor_dist, int max_nb, float time_horizon, float time_step, int agent, uint64_t seed, float k_ulps,
                      int probes, float* out_vel, int* trace_kind, int* trace_outcome, float* trace_margin, int cap, int* n_trace)
{
    float* tmp = (float*)malloc(sizeof(float) * 2 * (size_t)na);
    for (int p = 0; p < probes; ++p) {
        PS.rng = (seed + (uint64_t)p) * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
        if (PS.rng == 0) PS.rng = 1;
        for (int i = 0; i < 8; ++i) (void)p_unit();
        PS.k = p == 0 ? 0.0f : k_ulps;
        PS.only_agent = agent; PS.active = 0;
        PS.kind = trace_kind ? trace_kind + (size_t)p * cap : NULL;
        PS.outcome = trace_outcome ? trace_outcome + (size_t)p * cap : NULL;
        PS.margin = trace_margin ? trace_margin + (size_t)p * cap : NULL;
        PS.cap = cap; PS.n = 0;
        memset(tmp, 0, sizeof(float) * 2 * (size_t)na);
        orcp_orca_new_velocities(na, pos, vel, pref, radius, maxspeed, neighbor_dist, max_nb, time_horizon, time_step, tmp, NULL, NULL);
        PS.active = 0;
        out_vel[2 * p] = tmp[2 * agent]; out_vel[2 * p + 1] = tmp[2 * agent + 1];
        if (n_trace) n_trace[p] = PS.n;
    }
    free(tmp);
}
