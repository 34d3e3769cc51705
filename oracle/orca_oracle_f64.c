/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.   *** PARITY UNPINNED (like orca_oracle.c) ***
 *
 * The ORCA restatement of orca_oracle.c instantiated in DOUBLE: the same statements, every `float` a double.  It is not what
 * RVO2 computes (RVO2 is float32) -- it is what RVO2's algorithm yields without float32 rounding, and serves one purpose:
 * classifying the agent-substeps on which two float32 evaluations of the algorithm disagree (tests/orca_fast_parity.py: is
 * the exact float32 restatement itself within the bar of the real-arithmetic answer there?).
 * Exports orc64_orca_* with double arrays where orca_oracle.c has float arrays.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define float double
#define sqrtf sqrt
#define fabsf fabs
#define fminf fmin
#define fmaxf fmax
#define orc_orca_new_velocities_pa orc64_orca_new_velocities_pa
#define orc_orca_new_velocities_obst orc64_orca_new_velocities_obst
#define orc_orca_new_velocities orc64_orca_new_velocities
#define orc_orca_step_block_pa orc64_orca_step_block_pa
#define orc_orca_step_block_obst orc64_orca_step_block_obst
#define orc_orca_step_block orc64_orca_step_block
#define orc_orca_step_block_batched_pa orc64_orca_step_block_batched_pa
#define orc_orca_step_block_batched_obst orc64_orca_step_block_batched_obst
#define orc_orca_step_block_batched orc64_orca_step_block_batched
#include "orca_oracle.c"
