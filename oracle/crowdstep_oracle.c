/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's per-substep pedestrian update, used as the parity
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  It is never
 * imported, linked or executed by the shipped package (social_navigation_pyenvs_amd), which must
 * fail loudly when its HIP library is missing.
 *
 * SFM / HSFM / Gym-loop pieces: PINNED against golden vectors captured by running the reference
 * itself in the build container (tests/golden/make_golden.py, fixtures g1..g7).
 * ORCA (orca_oracle.c): PARITY UNPINNED -- the reference delegates to the third-party RVO2 library
 * (Python-RVO2 wrapper, un-vendored, un-pinned, absent here); see that file's header.
 *
 * Build: make -C oracle   (gcc, -ffp-contract=off so the f64 instantiation rounds like numpy).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REAL double
#define ORC(name) orc_##name##_f64
#define SQRT sqrt
#define EXP exp
#define COS cos
#define SIN sin
#define ATAN2 atan2
#define FMOD fmod
#include "sfm_step.inc"
#undef REAL
#undef ORC
#undef SQRT
#undef EXP
#undef COS
#undef SIN
#undef ATAN2
#undef FMOD

#define REAL float
#define ORC(name) orc_##name##_f32
#define SQRT sqrtf
#define EXP expf
#define COS cosf
#define SIN sinf
#define ATAN2 atan2f
#define FMOD fmodf
#include "sfm_step.inc"

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
