// Diagnostic: where does the dispatcher put G workgroups of T threads (LDS bytes each as given)? Each wave spins on a
// dependent fma chain for about the run time of one k_sfm_step launch and records HW_ID / XCC_ID and its start and end clocks.
// Prints the histogram of waves per SIMD and the spread of start times (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

__global__ void k_spin(unsigned long long* rec, int iters, float seed)
{
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    float a = seed + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a = __builtin_fmaf(a, 1.0001f, 1e-6f);
    }
    if (a == 12345.678f) lds[threadIdx.x] = a;
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)); // HW_REG_XCC_ID
        const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        rec[wave * 6 + 0] = hw; rec[wave * 6 + 1] = xcc; rec[wave * 6 + 2] = t0; rec[wave * 6 + 3] = t1;
        rec[wave * 6 + 4] = r0; rec[wave * 6 + 5] = r1;
    }
}

int main(int argc, char** argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 2048, T = argc > 2 ? atoi(argv[2]) : 64, lds = argc > 3 ? atoi(argv[3]) : 4096;
    const int iters = argc > 4 ? atoi(argv[4]) : 1500;
    const long waves = (long)G * (T / 64);
    unsigned long long* d;
    hipMalloc(&d, waves * 6 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_spin, dim3(G), dim3(T), lds, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves * 6);
    hipMemcpy(h.data(), d, waves * 6 * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu, per_xcc;
    unsigned long long rmin = ~0ull, rmax = 0;
    double lone = 0;
    for (long w = 0; w < waves; ++w) {
        const unsigned hw = (unsigned)h[w * 6], xcc = (unsigned)h[w * 6 + 1] & 0xF;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_simd[(cukey << 2) | simd]++; per_cu[cukey]++; per_xcc[xcc]++;
        rmin = std::min(rmin, h[w * 6 + 4]); rmax = std::max(rmax, h[w * 6 + 5]);
        lone += (double)(h[w * 6 + 5] - h[w * 6 + 4]);
    }
    std::map<int, int> hist, hist_cu;
    for (auto& kv : per_simd) hist[kv.second]++;
    for (auto& kv : per_cu) hist_cu[kv.second]++;
    printf("G=%d T=%d lds=%d iters=%d: kernel %.1f us (events), first start -> last end %.1f us, mean wave %.1f us (100 MHz clock)\n",
           G, T, lds, iters, ms * 1e3, (rmax - rmin) / 100.0, lone / waves / 100.0);
    printf("  SIMDs used %zu, CUs used %zu, XCCs used %zu\n  waves per SIMD histogram:", per_simd.size(), per_cu.size(), per_xcc.size());
    for (auto& kv : hist) printf("  %d waves: %d SIMDs;", kv.first, kv.second);
    printf("\n  waves per CU histogram:");
    for (auto& kv : hist_cu) printf("  %d waves: %d CUs;", kv.first, kv.second);
    printf("\n  waves per XCC:");
    for (auto& kv : per_xcc) printf(" %d", kv.second);
    // start-time spread: how late does the last wave start relative to the first?
    std::vector<unsigned long long> starts;
    for (long w = 0; w < waves; ++w) starts.push_back(h[w * 6 + 4] - rmin);
    std::sort(starts.begin(), starts.end());
    printf("\n  wave start offsets (us): p50 %.2f p90 %.2f p99 %.2f max %.2f\n", starts[waves / 2] / 100.0, starts[waves * 9 / 10] / 100.0,
           starts[waves * 99 / 100] / 100.0, starts[waves - 1] / 100.0);
    return 0;
}
