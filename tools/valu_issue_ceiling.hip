// Diagnostic microbenchmark (round 5): what ONE SIMD of gfx950 sustains in wave64 vector instructions per cycle when 1, 2, 4 or 8
// wavefronts share it -- the denominator of bench.py's `valu_frac`.  MI355X_MICROARCH.md prices v_fma_f32 (wave64) at 2 cycles on
// the SIMD-32 and at 4 for one wavefront alone; tools/valu_microbench.hip only ever ran one layout.  Here every stream is
// written as inline assembly (the compiler can neither fuse, reorder nor drop an instruction), every wavefront runs the same
// count of instructions, and the cost is reported twice: from the wall clock of the launch (at the nominal 2.4 GHz) and from
// s_memtime inside the kernel (shader cycles of the slowest wavefront), so a clock that sags under load shows as a gap.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_issue_ceiling tools/valu_issue_ceiling.hip && tools/valu_issue_ceiling
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

enum { OP_FMA = 0, OP_EXP, OP_RCP, OP_CNDMASK, OP_MUL, OP_FMA_DEP, OP_PAIRMIX, OP_LPMIX, OP_DPP, OP_CMP_CND, OP_MAX3,
       OP_ADD, OP_MAX, OP_CMP, OP_CND_SGPR, OP_CND_INDEP, OP_CMP_CND4, OP_FMA3, OP_MIN64, OP_MOV, OP_AND, OP_CND_SMOV, OP_CND_E64_VCC, OP_CMP_NOP_CND2, OP_CMP_NOP_CND2_E64, OP_CMP_NOP_CND2_SGPR, OP_CND_E32_ALT };

// instructions per unrolled block of each stream (for the per-instruction price)
__host__ __device__ constexpr int block_insts(int op)
{
    return op == OP_PAIRMIX ? 36 : (op == OP_LPMIX ? 24 : (op == OP_CMP_CND ? 16 : (op == OP_CMP_CND4 ? 10 : ((op == OP_CMP_NOP_CND2 || op == OP_CMP_NOP_CND2_E64 || op == OP_CMP_NOP_CND2_SGPR) ? 12 : 8))));
}

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters, float seed)
{
    float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 1.0001f, d = 1e-6f;
    __shared__ float sh[256 * 2];
    sh[threadIdx.x] = a0; sh[256 + threadIdx.x] = a1;
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int ldsaddr = (threadIdx.x & 63) * 16;
    f4 qa = {a0, a1, a2, a3}, qb = qa;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4;
    const unsigned long long smask = 0x5555AAAA3333CCCCull ^ (unsigned long long)iters;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if (OP == OP_FMA) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                              "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (OP == OP_MUL) {
            asm volatile(REP8("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                              "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == OP_EXP) {
            asm volatile(REP8("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                              "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == OP_RCP) {
            asm volatile(REP8("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                              "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == OP_CNDMASK) {   // vcc set once outside the stream: the select alone
            asm volatile("v_cmp_gt_f32 vcc, %8, %0\n\t"
                         REP8("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_cndmask_b32 %6, %6, %7, vcc\n\t"
                              "v_cndmask_b32 %1, %1, %0, vcc\n\tv_cndmask_b32 %3, %3, %2, vcc\n\tv_cndmask_b32 %5, %5, %4, vcc\n\tv_cndmask_b32 %7, %7, %6, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        } else if (OP == OP_CMP_CND) {   // compare + select pairs, as the branch-free LP bodies have them
            asm volatile(REP8("v_cmp_gt_f32 vcc, %8, %0\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_gt_f32 vcc, %8, %2\n\tv_cndmask_b32 %2, %2, %3, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %4\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_cmp_gt_f32 vcc, %8, %6\n\tv_cndmask_b32 %6, %6, %7, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %1\n\tv_cndmask_b32 %1, %1, %0, vcc\n\tv_cmp_gt_f32 vcc, %8, %3\n\tv_cndmask_b32 %3, %3, %2, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %5\n\tv_cndmask_b32 %5, %5, %4, vcc\n\tv_cmp_gt_f32 vcc, %8, %7\n\tv_cndmask_b32 %7, %7, %6, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        } else if (OP == OP_FMA_DEP) {
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                              "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t")
                         : "+v"(a0) : "v"(c), "v"(d));
        } else if (OP == OP_MAX3) {
            asm volatile(REP8("v_max3_f32 %0, %0, %8, %9\n\tv_max3_f32 %1, %1, %8, %9\n\tv_max3_f32 %2, %2, %8, %9\n\tv_max3_f32 %3, %3, %8, %9\n\t"
                              "v_max3_f32 %4, %4, %8, %9\n\tv_max3_f32 %5, %5, %8, %9\n\tv_max3_f32 %6, %6, %8, %9\n\tv_max3_f32 %7, %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (OP == OP_DPP) {       // DPP row rotations fused into v_min / v_max (the LP3 row reductions)
            asm volatile(REP8("v_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                              "v_min_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                              "v_min_f32_dpp %4, %4, %4 row_ror:2 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %5, %5, %5 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                              "v_min_f32_dpp %6, %6, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %7, %7, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == OP_ADD) {
            asm volatile(REP8("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
                              "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == OP_MAX) {
            asm volatile(REP8("v_max_f32 %0, %0, %8\n\tv_min_f32 %1, %1, %8\n\tv_max_f32 %2, %2, %8\n\tv_min_f32 %3, %3, %8\n\t"
                              "v_max_f32 %4, %4, %8\n\tv_min_f32 %5, %5, %8\n\tv_max_f32 %6, %6, %8\n\tv_min_f32 %7, %7, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == OP_MOV) {
            asm volatile(REP8("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4\n\t"
                              "v_mov_b32 %4, %5\n\tv_mov_b32 %5, %6\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %0\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == OP_AND) {
            asm volatile(REP8("v_and_b32 %0, %0, %8\n\tv_or_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\t"
                              "v_xor_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == OP_CMP) {       // compares alone, each into its own SGPR pair
            unsigned long long m0, m1, m2, m3;
            asm volatile(REP8("v_cmp_gt_f32 %0, %4, %5\n\tv_cmp_lt_f32 %1, %6, %7\n\tv_cmp_gt_f32 %2, %8, %9\n\tv_cmp_lt_f32 %3, %10, %11\n\t"
                              "v_cmp_gt_f32 %0, %5, %6\n\tv_cmp_lt_f32 %1, %7, %8\n\tv_cmp_gt_f32 %2, %9, %10\n\tv_cmp_lt_f32 %3, %11, %4\n\t")
                         : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            a0 += (float)(m0 & 1) * 0.0f;
        } else if (OP == OP_CND_SGPR) {  // selects on a lane mask held in an SGPR pair that no VALU instruction wrote
            asm volatile(REP8("v_cndmask_b32 %0, %0, %1, %8\n\tv_cndmask_b32 %2, %2, %3, %8\n\tv_cndmask_b32 %4, %4, %5, %8\n\tv_cndmask_b32 %6, %6, %7, %8\n\t"
                              "v_cndmask_b32 %1, %1, %0, %8\n\tv_cndmask_b32 %3, %3, %2, %8\n\tv_cndmask_b32 %5, %5, %4, %8\n\tv_cndmask_b32 %7, %7, %6, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(smask));
        } else if (OP == OP_CND_SMOV) {  // the same with vcc written by the scalar unit
            asm volatile("s_mov_b64 vcc, %8\n\t"
                         REP8("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_cndmask_b32 %6, %6, %7, vcc\n\t"
                              "v_cndmask_b32 %1, %1, %0, vcc\n\tv_cndmask_b32 %3, %3, %2, vcc\n\tv_cndmask_b32 %5, %5, %4, vcc\n\tv_cndmask_b32 %7, %7, %6, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(smask) : "vcc");
        } else if (OP == OP_CND_INDEP) { // selects whose sources are never a destination (no register dependence at all)
            float t0, t1, t2, t3;
            asm volatile("v_cmp_gt_f32 vcc, %12, %4\n\t"
                         REP8("v_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %6, %7, vcc\n\tv_cndmask_b32 %2, %8, %9, vcc\n\tv_cndmask_b32 %3, %10, %11, vcc\n\t"
                              "v_cndmask_b32 %0, %5, %6, vcc\n\tv_cndmask_b32 %1, %7, %8, vcc\n\tv_cndmask_b32 %2, %9, %10, vcc\n\tv_cndmask_b32 %3, %11, %4, vcc\n\t")
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(c) : "vcc");
            a0 += t0 * 0.0f;
        } else if (OP == OP_CMP_CND4) {  // one compare feeding four selects (x2): 10 instructions
            asm volatile(REP8("v_cmp_gt_f32 vcc, %8, %0\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_cndmask_b32 %6, %6, %7, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %1\n\tv_cndmask_b32 %1, %1, %0, vcc\n\tv_cndmask_b32 %3, %3, %2, vcc\n\tv_cndmask_b32 %5, %5, %4, vcc\n\tv_cndmask_b32 %7, %7, %6, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        } else if (OP == OP_FMA3) {      // fma with three distinct register sources per instruction
            asm volatile(REP8("v_fma_f32 %0, %1, %2, %3\n\tv_fma_f32 %1, %2, %3, %4\n\tv_fma_f32 %2, %3, %4, %5\n\tv_fma_f32 %3, %4, %5, %6\n\t"
                              "v_fma_f32 %4, %5, %6, %7\n\tv_fma_f32 %5, %6, %7, %0\n\tv_fma_f32 %6, %7, %0, %1\n\tv_fma_f32 %7, %0, %1, %2\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == OP_MIN64) {     // the neighbour list's compare-exchange: v_min_f64 / v_max_f64
            asm volatile(REP8("v_min_f64 %0, %0, %4\n\tv_max_f64 %1, %1, %4\n\tv_min_f64 %2, %2, %4\n\tv_max_f64 %3, %3, %4\n\t"
                              "v_min_f64 %0, %0, %4\n\tv_max_f64 %1, %1, %4\n\tv_min_f64 %2, %2, %4\n\tv_max_f64 %3, %3, %4\n\t")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4));
        } else if (OP == OP_CND_E64_VCC) {  // the VOP3 encoding of the select, mask still in vcc (written once by a compare)
            asm volatile("v_cmp_gt_f32 vcc, %8, %0\n\ts_nop 1\n\t"
                         REP8("v_cndmask_b32_e64 %0, %0, %1, vcc\n\tv_cndmask_b32_e64 %2, %2, %3, vcc\n\tv_cndmask_b32_e64 %4, %4, %5, vcc\n\tv_cndmask_b32_e64 %6, %6, %7, vcc\n\t"
                              "v_cndmask_b32_e64 %1, %1, %0, vcc\n\tv_cndmask_b32_e64 %3, %3, %2, vcc\n\tv_cndmask_b32_e64 %5, %5, %4, vcc\n\tv_cndmask_b32_e64 %7, %7, %6, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        } else if (OP == OP_CND_E32_ALT) {  // VOP2 selects on vcc with an unrelated fma between any two of them
            asm volatile("v_cmp_gt_f32 vcc, %8, %0\n\ts_nop 1\n\t"
                         REP8("v_cndmask_b32 %0, %0, %1, vcc\n\tv_fma_f32 %2, %2, %8, %8\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_fma_f32 %6, %6, %8, %8\n\t"
                              "v_cndmask_b32 %1, %1, %0, vcc\n\tv_fma_f32 %3, %3, %8, %8\n\tv_cndmask_b32 %5, %5, %4, vcc\n\tv_fma_f32 %7, %7, %8, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        } else if (OP == OP_CMP_NOP_CND2) { // what hipcc emits for `x = c ? a : x; y = c ? b : y`: v_cmp -> vcc, s_nop 1, two VOP2 selects  (x4: 12 VALU)
            asm volatile(REP8("v_cmp_gt_f32 vcc, %8, %0\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %4\n\ts_nop 1\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_cndmask_b32 %6, %6, %7, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %1\n\ts_nop 1\n\tv_cndmask_b32 %1, %1, %0, vcc\n\tv_cndmask_b32 %3, %3, %2, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %5\n\ts_nop 1\n\tv_cndmask_b32 %5, %5, %4, vcc\n\tv_cndmask_b32 %7, %7, %6, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        } else if (OP == OP_CMP_NOP_CND2_E64) { // the same with the VOP3 encoding of the selects
            asm volatile(REP8("v_cmp_gt_f32 vcc, %8, %0\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, vcc\n\tv_cndmask_b32_e64 %2, %2, %3, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %4, %4, %5, vcc\n\tv_cndmask_b32_e64 %6, %6, %7, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, %1, %0, vcc\n\tv_cndmask_b32_e64 %3, %3, %2, vcc\n\t"
                              "v_cmp_gt_f32 vcc, %8, %5\n\ts_nop 1\n\tv_cndmask_b32_e64 %5, %5, %4, vcc\n\tv_cndmask_b32_e64 %7, %7, %6, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        } else if (OP == OP_CMP_NOP_CND2_SGPR) { // the same with the compare writing an SGPR pair
            unsigned long long m = smask;
            asm volatile(REP8("v_cmp_gt_f32 %9, %8, %0\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, %9\n\tv_cndmask_b32 %2, %2, %3, %9\n\t"
                              "v_cmp_gt_f32 %9, %8, %4\n\ts_nop 1\n\tv_cndmask_b32 %4, %4, %5, %9\n\tv_cndmask_b32 %6, %6, %7, %9\n\t"
                              "v_cmp_gt_f32 %9, %8, %1\n\ts_nop 1\n\tv_cndmask_b32 %1, %1, %0, %9\n\tv_cndmask_b32 %3, %3, %2, %9\n\t"
                              "v_cmp_gt_f32 %9, %8, %5\n\ts_nop 1\n\tv_cndmask_b32 %5, %5, %4, %9\n\tv_cndmask_b32 %7, %7, %6, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(m));
        } else if (OP == OP_PAIRMIX) {
            // the social-force pair body's mix (sfmstep_kernel.h, Helbing / Farina, equal parameters): per partner one LDS row read
            // (ds_read_b128, issued one partner ahead as the real loop does), 2 sub, mul + fma for |d|^2, v_rsq, 3 mul, 1 fma, v_exp,
            // 2 mul, 2 max, compare + select, 2 fma accumulate: 16 plain VALU + 2 quarter-rate + 1 LDS read per partner
#define PAIR_BODY                                                                                                               \
            "v_sub_f32 %2, %4, %0\n\tv_sub_f32 %3, %5, %1\n\t"                                                                \
            "v_mul_f32 %6, %2, %2\n\tv_fma_f32 %6, %3, %3, %6\n\t"                                                            \
            "v_rsq_f32 %7, %6\n\t"                                                                                              \
            "v_mul_f32 %6, %6, %7\n\tv_mul_f32 %2, %2, %7\n\tv_mul_f32 %3, %3, %7\n\t"                                     \
            "v_fma_f32 %6, %6, %8, %9\n\t"                                                                                      \
            "v_exp_f32 %7, %6\n\t"                                                                                              \
            "v_mul_f32 %7, %7, %8\n\tv_mul_f32 %6, %6, %9\n\t"                                                               \
            "v_max_f32 %6, %6, %9\n\tv_max_f32 %7, %7, %9\n\t"                                                               \
            "v_cmp_gt_f32 vcc, %6, %9\n\tv_cndmask_b32 %7, %7, %6, vcc\n\t"                                                  \
            "v_fma_f32 %0, %7, %2, %0\n\tv_fma_f32 %1, %7, %3, %1\n\t"
            asm volatile("ds_read_b128 %0, %1" : "=v"(qb) : "v"(ldsaddr) : "memory");
            asm volatile(PAIR_BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(qa.x), "+v"(qa.y), "+v"(a4), "+v"(a5) : "v"(c), "v"(d) : "vcc");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(qa) : "v"(ldsaddr) : "memory");
            asm volatile(PAIR_BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(qb.x), "+v"(qb.y), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "vcc");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == OP_LPMIX) {
            // one linearProgram1 inner body with v_rcp (the fast ORCA build): 2 det2 (2 mul + 2 fma), 2 sub, v_rcp, mul, abs-compare,
            // compare, min, max, 4 selects, and / or of lane masks (SALU, not counted): 3 x { 7 plain + 1 quarter-rate } = 24
            asm volatile("v_mul_f32 %4, %0, %1\n\tv_fma_f32 %4, %2, %3, -%4\n\tv_sub_f32 %5, %0, %2\n\tv_sub_f32 %6, %1, %3\n\t"
                         "v_rcp_f32 %7, %4\n\tv_mul_f32 %5, %5, %7\n\tv_cmp_gt_f32 vcc, %8, %5\n\tv_cndmask_b32 %0, %0, %5, vcc\n\t"
                         "v_mul_f32 %4, %1, %2\n\tv_fma_f32 %4, %3, %0, -%4\n\tv_sub_f32 %5, %1, %3\n\tv_sub_f32 %6, %2, %0\n\t"
                         "v_rcp_f32 %7, %4\n\tv_min_f32 %5, %5, %7\n\tv_cmp_gt_f32 vcc, %8, %5\n\tv_cndmask_b32 %1, %1, %5, vcc\n\t"
                         "v_mul_f32 %4, %2, %3\n\tv_fma_f32 %4, %0, %1, -%4\n\tv_sub_f32 %5, %2, %0\n\tv_sub_f32 %6, %3, %1\n\t"
                         "v_rcp_f32 %7, %4\n\tv_max_f32 %5, %5, %7\n\tv_cmp_gt_f32 vcc, %8, %5\n\tv_cndmask_b32 %2, %2, %5, vcc\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + qa.x + qb.x + (float)(d0 + d1 + d2 + d3);
    if ((threadIdx.x & 63) == 0) ticks[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// blocks of 256 lanes = four wavefronts = one per SIMD of a CU; `w` blocks per CU give w wavefronts per SIMD
template <int OP>
void run(const char* name, int w, float* d_out, unsigned long long* d_ticks)
{
    const int iters = 2048, blocks = 256 * w;
    const int reps = (OP == OP_PAIRMIX || OP == OP_LPMIX) ? 1 : 8;
    std::vector<unsigned long long> ticks((size_t)blocks * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, iters, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(ticks.data(), d_ticks, ticks.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(ticks.begin(), ticks.end());
    const double n_inst = (double)iters * reps * block_insts(OP);      // vector instructions per wavefront
    const double wall_cyc = ms * 1e-3 * 2.4e9 / (n_inst * w);          // SIMD cycles per wave-instruction, all w wavefronts sharing the SIMD
    const double tick_med = (double)ticks[ticks.size() / 2] / (n_inst * w), tick_max = (double)ticks.back() / (n_inst * w);
    printf("%-34s waves/SIMD=%d  %8.3f ms  wall@2.4GHz %5.2f  s_memtime median %5.2f max %5.2f  cycles per wave64 instruction per SIMD\n", name, w, ms, wall_cyc,
           tick_med, tick_max);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    float* d_out;
    unsigned long long* d_ticks;
    hipMalloc(&d_out, (size_t)256 * 8 * 256 * sizeof(float));
    hipMalloc(&d_ticks, (size_t)256 * 8 * 4 * sizeof(unsigned long long));
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("# %s, %d CUs, clockRate %d kHz; blocks of 4 wavefronts (one per SIMD), w blocks per CU\n", p.name, p.multiProcessorCount, p.clockRate);
    printf("# s_memtime: ticks of the slowest / median wavefront between its first and last instruction, divided by (instructions x w)\n");
    for (int w : {1, 2, 4, 8}) {
        run<OP_FMA>("v_fma_f32 x8 independent", w, d_out, d_ticks);
        run<OP_MUL>("v_mul_f32 x8 independent", w, d_out, d_ticks);
        run<OP_MAX3>("v_max3_f32 x8 independent", w, d_out, d_ticks);
        run<OP_FMA_DEP>("v_fma_f32 dependent chain", w, d_out, d_ticks);
        run<OP_EXP>("v_exp_f32 x8 independent", w, d_out, d_ticks);
        run<OP_RCP>("v_rcp_f32 x8 independent", w, d_out, d_ticks);
        run<OP_CNDMASK>("v_cndmask_b32 x8 (vcc fixed)", w, d_out, d_ticks);
        run<OP_CMP_CND>("v_cmp + v_cndmask x8", w, d_out, d_ticks);
        run<OP_DPP>("v_min/max_f32_dpp row_ror x8", w, d_out, d_ticks);
        run<OP_ADD>("v_add_f32 x8 independent", w, d_out, d_ticks);
        run<OP_MAX>("v_max/min_f32 x8 independent", w, d_out, d_ticks);
        run<OP_MOV>("v_mov_b32 x8", w, d_out, d_ticks);
        run<OP_AND>("v_and/or/xor_b32 x8", w, d_out, d_ticks);
        run<OP_FMA3>("v_fma_f32 x8, 3 distinct VGPR sources", w, d_out, d_ticks);
        run<OP_MIN64>("v_min/max_f64 x8", w, d_out, d_ticks);
        run<OP_CMP>("v_cmp_f32 -> SGPR pair x8", w, d_out, d_ticks);
        run<OP_CND_SGPR>("v_cndmask_b32 x8 (SGPR-pair mask)", w, d_out, d_ticks);
        run<OP_CND_SMOV>("v_cndmask_b32 x8 (vcc from s_mov)", w, d_out, d_ticks);
        run<OP_CND_INDEP>("v_cndmask_b32 x8 (no reg dependence)", w, d_out, d_ticks);
        run<OP_CMP_CND4>("v_cmp + 4 v_cndmask x2", w, d_out, d_ticks);
        run<OP_CND_E64_VCC>("v_cndmask_b32_e64 x8 (vcc fixed)", w, d_out, d_ticks);
        run<OP_CND_E32_ALT>("v_cndmask e32 vcc / v_fma alternating", w, d_out, d_ticks);
        run<OP_CMP_NOP_CND2>("cmp->vcc, s_nop 1, 2 cndmask e32", w, d_out, d_ticks);
        run<OP_CMP_NOP_CND2_E64>("cmp->vcc, s_nop 1, 2 cndmask e64", w, d_out, d_ticks);
        run<OP_CMP_NOP_CND2_SGPR>("cmp->sgpr, s_nop 1, 2 cndmask sgpr", w, d_out, d_ticks);
        run<OP_PAIRMIX>("pair-loop mix (16 VALU+2 trans+LDS)", w, d_out, d_ticks);
        run<OP_LPMIX>("LP1 body mix (21 VALU + 3 v_rcp)", w, d_out, d_ticks);
    }
    return 0;
}
