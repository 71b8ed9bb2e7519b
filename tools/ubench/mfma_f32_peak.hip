// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on MI355X (practical ceiling for the igemm kernel).
// build: hipcc --offload-arch=gfx950 -O3 mfma_f32_peak.hip -o mfma_f32_peak ; run: ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters)
{
    int grid = 256 * blocks_per_cu;
    float *out; hipMalloc(&out, grid * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
    printf("nacc %d blocks/CU %d iters %6d: %8.3f ms  %6.1f TFLOP/s\n", NACC, blocks_per_cu, iters, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<4>(1, 200); run<4>(1, 2000); run<4>(1, 20000); run<4>(2, 2000); run<4>(2, 20000);
    run<1>(1, 2000); run<2>(1, 2000); run<2>(2, 2000); run<1>(2, 2000); run<1>(4, 2000);
    return 0;
}
