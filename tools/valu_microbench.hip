// Diagnostic microbenchmark: SIMD issue cost of the VALU ops the pair loop is made of (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed, int flags)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, c = {1.0001f, 0.9999f}, d = {1e-6f, -1e-6f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { // 8 independent v_fma_f32
                a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); a1 = __builtin_fmaf(a1, 1.0001f, 1e-6f); a2 = __builtin_fmaf(a2, 1.0001f, 1e-6f); a3 = __builtin_fmaf(a3, 1.0001f, 1e-6f);
                a4 = __builtin_fmaf(a4, 1.0001f, 1e-6f); a5 = __builtin_fmaf(a5, 1.0001f, 1e-6f); a6 = __builtin_fmaf(a6, 1.0001f, 1e-6f); a7 = __builtin_fmaf(a7, 1.0001f, 1e-6f);
            } else if (OP == 1) { // 4 independent v_pk_fma_f32 (= 8 fma)
                p0 = __builtin_elementwise_fma(p0, c, d); p1 = __builtin_elementwise_fma(p1, c, d); p2 = __builtin_elementwise_fma(p2, c, d); p3 = __builtin_elementwise_fma(p3, c, d);
            } else if (OP == 2) { // 8 independent v_exp_f32
                a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
                a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
            } else if (OP == 3) { // 8 independent v_rsq_f32
                a0 = __builtin_amdgcn_rsqf(a0); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_amdgcn_rsqf(a2); a3 = __builtin_amdgcn_rsqf(a3);
                a4 = __builtin_amdgcn_rsqf(a4); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_amdgcn_rsqf(a6); a7 = __builtin_amdgcn_rsqf(a7);
            } else if (OP == 4) { // 8 independent v_cndmask (select)
                a0 = a0 > 1.0f ? a1 : a0; a2 = a2 > 1.0f ? a3 : a2; a4 = a4 > 1.0f ? a5 : a4; a6 = a6 > 1.0f ? a7 : a6;
                a1 = a1 > 2.0f ? a0 : a1; a3 = a3 > 2.0f ? a2 : a3; a5 = a5 > 2.0f ? a4 : a5; a7 = a7 > 2.0f ? a6 : a7;
            } else if (OP == 5) { // one dependent chain of fma
                a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f);
                a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f);
            } else if (OP == 7) { // 8 independent ds_bpermute_b32 (rotate by one lane) + add
                const int src = ((threadIdx.x + 63) & 63) << 2;
                a0 += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a1)));
                a1 += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a2)));
                a2 += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a3)));
                a3 += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a4)));
                a4 += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a5)));
                a5 += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a6)));
                a6 += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a7)));
                a7 += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a0)));
            } else if (OP == 8) { // the same 8 adds without the permute (baseline for OP 7)
                a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0;
            } else if (OP == 9) { // 8 x { fma ; rare divergent branch that no lane takes (v_cmp, s_and_saveexec, s_cbranch_execz) }
                a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); if (a0 < -1e30f) { a0 = out[threadIdx.x]; }
                a1 = __builtin_fmaf(a1, 1.0001f, 1e-6f); if (a1 < -1e30f) { a1 = out[threadIdx.x + 1]; }
                a2 = __builtin_fmaf(a2, 1.0001f, 1e-6f); if (a2 < -1e30f) { a2 = out[threadIdx.x + 2]; }
                a3 = __builtin_fmaf(a3, 1.0001f, 1e-6f); if (a3 < -1e30f) { a3 = out[threadIdx.x + 3]; }
                a4 = __builtin_fmaf(a4, 1.0001f, 1e-6f); if (a4 < -1e30f) { a4 = out[threadIdx.x + 4]; }
                a5 = __builtin_fmaf(a5, 1.0001f, 1e-6f); if (a5 < -1e30f) { a5 = out[threadIdx.x + 5]; }
                a6 = __builtin_fmaf(a6, 1.0001f, 1e-6f); if (a6 < -1e30f) { a6 = out[threadIdx.x + 6]; }
                a7 = __builtin_fmaf(a7, 1.0001f, 1e-6f); if (a7 < -1e30f) { a7 = out[threadIdx.x + 7]; }
            } else if (OP == 10) { // 8 x { fma ; wave-uniform branch on a ballot that is never set (v_cmp, s_cmp, s_cbranch_scc) }
                a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); if (__builtin_amdgcn_ballot_w64(a0 < -1e30f) != 0) { a0 = out[threadIdx.x]; }
                a1 = __builtin_fmaf(a1, 1.0001f, 1e-6f); if (__builtin_amdgcn_ballot_w64(a1 < -1e30f) != 0) { a1 = out[threadIdx.x + 1]; }
                a2 = __builtin_fmaf(a2, 1.0001f, 1e-6f); if (__builtin_amdgcn_ballot_w64(a2 < -1e30f) != 0) { a2 = out[threadIdx.x + 2]; }
                a3 = __builtin_fmaf(a3, 1.0001f, 1e-6f); if (__builtin_amdgcn_ballot_w64(a3 < -1e30f) != 0) { a3 = out[threadIdx.x + 3]; }
                a4 = __builtin_fmaf(a4, 1.0001f, 1e-6f); if (__builtin_amdgcn_ballot_w64(a4 < -1e30f) != 0) { a4 = out[threadIdx.x + 4]; }
                a5 = __builtin_fmaf(a5, 1.0001f, 1e-6f); if (__builtin_amdgcn_ballot_w64(a5 < -1e30f) != 0) { a5 = out[threadIdx.x + 5]; }
                a6 = __builtin_fmaf(a6, 1.0001f, 1e-6f); if (__builtin_amdgcn_ballot_w64(a6 < -1e30f) != 0) { a6 = out[threadIdx.x + 6]; }
                a7 = __builtin_fmaf(a7, 1.0001f, 1e-6f); if (__builtin_amdgcn_ballot_w64(a7 < -1e30f) != 0) { a7 = out[threadIdx.x + 7]; }
            } else if (OP == 11) { // 8 x { LDS round trip: ds_write_b32 own slot, ds_read_b32 neighbour slot, use }
                __shared__ float sh[64];
                sh[threadIdx.x] = a0; a0 += sh[(threadIdx.x + 1) & 63];
                sh[threadIdx.x] = a0; a0 += sh[(threadIdx.x + 1) & 63];
                sh[threadIdx.x] = a0; a0 += sh[(threadIdx.x + 1) & 63];
                sh[threadIdx.x] = a0; a0 += sh[(threadIdx.x + 1) & 63];
                sh[threadIdx.x] = a0; a0 += sh[(threadIdx.x + 1) & 63];
                sh[threadIdx.x] = a0; a0 += sh[(threadIdx.x + 1) & 63];
                sh[threadIdx.x] = a0; a0 += sh[(threadIdx.x + 1) & 63];
                sh[threadIdx.x] = a0; a0 += sh[(threadIdx.x + 1) & 63];
            } else if (OP == 12) { // dependent chain: rsq -> fma -> fma -> exp -> mul -> fma  (one pair evaluation), x8 serial
                for (int r = 0; r < 8; ++r) {
                    float i = __builtin_amdgcn_rsqf(a0 * a0 + 1.0f);
                    float rd = __builtin_fmaf(-a0, i, 0.6f);
                    float e = __builtin_amdgcn_exp2f(__builtin_fmaf(rd, 18.0f, 10.9f)) * i;
                    a0 = __builtin_fmaf(e, 1e-6f, a0);
                }
            } else if (OP == 13) { // 8 x { fma ; SALU-only branch on a kernel argument bit, body skipped (taken s_cbranch) }
                a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); if (flags & 1) { a0 = out[threadIdx.x]; }
                a1 = __builtin_fmaf(a1, 1.0001f, 1e-6f); if (flags & 2) { a1 = out[threadIdx.x + 1]; }
                a2 = __builtin_fmaf(a2, 1.0001f, 1e-6f); if (flags & 4) { a2 = out[threadIdx.x + 2]; }
                a3 = __builtin_fmaf(a3, 1.0001f, 1e-6f); if (flags & 8) { a3 = out[threadIdx.x + 3]; }
                a4 = __builtin_fmaf(a4, 1.0001f, 1e-6f); if (flags & 16) { a4 = out[threadIdx.x + 4]; }
                a5 = __builtin_fmaf(a5, 1.0001f, 1e-6f); if (flags & 32) { a5 = out[threadIdx.x + 5]; }
                a6 = __builtin_fmaf(a6, 1.0001f, 1e-6f); if (flags & 64) { a6 = out[threadIdx.x + 6]; }
                a7 = __builtin_fmaf(a7, 1.0001f, 1e-6f); if (flags & 128) { a7 = out[threadIdx.x + 7]; }
            } else if (OP == 14) { // 8 x { fma ; loop-invariant exec-mask region (lanes < 50), body = one fma }
                a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); if (threadIdx.x < 50) { a1 = __builtin_fmaf(a1, 1.0001f, 1e-6f); }
                a2 = __builtin_fmaf(a2, 1.0001f, 1e-6f); if (threadIdx.x < 50) { a3 = __builtin_fmaf(a3, 1.0001f, 1e-6f); }
                a4 = __builtin_fmaf(a4, 1.0001f, 1e-6f); if (threadIdx.x < 50) { a5 = __builtin_fmaf(a5, 1.0001f, 1e-6f); }
                a6 = __builtin_fmaf(a6, 1.0001f, 1e-6f); if (threadIdx.x < 50) { a7 = __builtin_fmaf(a7, 1.0001f, 1e-6f); }
                a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); if (threadIdx.x < 50) { a1 = __builtin_fmaf(a1, 1.0001f, 1e-6f); }
                a2 = __builtin_fmaf(a2, 1.0001f, 1e-6f); if (threadIdx.x < 50) { a3 = __builtin_fmaf(a3, 1.0001f, 1e-6f); }
                a4 = __builtin_fmaf(a4, 1.0001f, 1e-6f); if (threadIdx.x < 50) { a5 = __builtin_fmaf(a5, 1.0001f, 1e-6f); }
                a6 = __builtin_fmaf(a6, 1.0001f, 1e-6f); if (threadIdx.x < 50) { a7 = __builtin_fmaf(a7, 1.0001f, 1e-6f); }
            } else if (OP == 6) { // 8 v_mul_f32 (plain, not fma)
                a0 *= 1.0001f; a1 *= 1.0001f; a2 *= 1.0001f; a3 *= 1.0001f; a4 *= 1.0001f; a5 *= 1.0001f; a6 *= 1.0001f; a7 *= 1.0001f;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int OP>
void run(const char* name, int waves_per_simd, float* d_out)
{
    const int iters = 4096;
    const int blocks = 256 * 4 * waves_per_simd; // one wave per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_inst = (double)iters * 8 * (OP == 1 ? 4 : 8);  // wave-instructions per wave
    // SIMD cycles per wave-instruction at 2.4 GHz nominal, all waves of a SIMD sharing it
    const double cyc = ms * 1e-3 * 2.4e9 / (n_inst * waves_per_simd);
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
}

int main()
{
    float* d_out; hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32 x8 indep", w, d_out);
        run<1>("v_pk_fma_f32 x4 indep", w, d_out);
        run<6>("v_mul_f32 x8 indep", w, d_out);
        run<2>("v_exp_f32 x8 indep", w, d_out);
        run<3>("v_rsq_f32 x8 indep", w, d_out);
        run<4>("v_cmp+v_cndmask x8", w, d_out);
        run<5>("v_fma_f32 dependent chain", w, d_out);
        run<7>("ds_bpermute_b32 + v_add x8", w, d_out);
        run<8>("v_add x8 (chain, no permute)", w, d_out);
        run<9>("fma + untaken divergent branch x8", w, d_out);
        run<10>("fma + untaken uniform branch x8", w, d_out);
        run<11>("LDS write->read->use chain x8", w, d_out);
        run<12>("pair chain rsq..exp..fma x8 (7 ops)", w, d_out);
        run<13>("fma + SALU-only skipped branch x8", w, d_out);
        run<14>("fma + invariant exec-mask region x8", w, d_out);
    }
    return 0;
}
