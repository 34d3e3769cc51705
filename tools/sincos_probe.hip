// Diagnostic: accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against double-precision sin / cos over
// [-pi, pi] -- can they replace the 20-instruction polynomial sincos_fast of the step kernels within the 1e-5 parity bar?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/sincos_probe tools/sincos_probe.hip && /tmp/sincos_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ void k(int n, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double es = 0, ec = 0;
    for (int k = i; k < n; k += gridDim.x * blockDim.x) {
        const float x = -3.14159265f + 6.2831853f * (float)k / (float)n;
        const float rev = x * 0.15915494309189535f;
        const float s = __builtin_amdgcn_sinf(rev), c = __builtin_amdgcn_cosf(rev);
        es = fmax(es, fabs((double)s - sin((double)x)));
        ec = fmax(ec, fabs((double)c - cos((double)x)));
    }
    // max over the block through atomics on the bit pattern (positive doubles order like integers)
    atomicMax((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(es));
    atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(ec));
}

int main()
{
    double* d;
    hipMalloc(&d, 16);
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, 1 << 26, d);
    double h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("max |v_sin - sin| = %.3e   max |v_cos - cos| = %.3e  over 2^26 points of [-pi, pi]\n", h[0], h[1]);
    return 0;
}
