// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 as a function of the number of independent accumulators the stream
// cycles through (dependency distance), one wave per SIMD.  Motivation: wino44_kernel issues its 8 MFMAs per transform position
// on two alternating accumulators and ran at 41 cycles per MFMA against the 32 of the matrix pipe.
// build: hipcc --offload-arch=gfx950 -O3 mfma_16x16x4_chain.hip -o mfma_16x16x4_chain.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float *out, long long *cyc, int iters, float a0, float b0)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int iters)
{
    int grid = 256;
    float *out; hipMalloc(&out, grid * 256 * 4);
    long long *cyc; hipMalloc(&cyc, grid * 8);
    k<NACC><<<grid, 256>>>(out, cyc, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    k<NACC><<<grid, 256>>>(out, cyc, iters, 1.f, 1.f);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid; ++i) s += (double)h[i];
    printf("independent accumulators %2d: %.2f cycles per v_mfma_f32_16x16x4_f32 (one wave per SIMD, all CUs busy)\n", NACC, s / grid / ((double)iters * 64));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<1>(2000); run<2>(2000); run<4>(2000); run<8>(2000); run<16>(2000);
    return 0;
}
