// Calibrates s_memtime (__builtin_readcyclecounter) against a dependent chain of N v_mfma_f32_32x32x2_f32 (64 cycles each at
// the shader clock, one wave per SIMD, nothing else running on it) and against wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void k(long long *out, int iters, float a0)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long)s; }
}
int main()
{
    long long *out; hipMalloc(&out, 1024 * 16);
    for (int grid : {1, 1024}) {
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<grid, 64>>>(out, 100, 1.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<grid, 64>>>(out, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        const double n = (double)iters * 32;
        printf("grid %4d: %lld ticks for %.0f MFMAs = %.2f ticks/MFMA; kernel %.3f ms -> %.3f G ticks/s, %.2f ns per MFMA (64 cycles at 2.4 GHz = 26.67 ns)\n",
               grid, h[0], n, h[0] / n, ms, h[0] / (ms * 1e6), ms * 1e6 / n);
    }
    return 0;
}
