// Pure v_mfma_f32_32x32x2_f32 stream for ~8 s (clock / power reference for tools/clock_probe.sh).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *out; hipMalloc(&out, 512 * 256 * 4);
    auto t0 = std::chrono::steady_clock::now();
    double ms_last = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 8.0) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<512, 256>>>(out, 20000, 1.f, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms_last = ms;
    }
    printf("mfma: last launch %.3f ms = %.1f TFLOP/s\n", ms_last, 512.0 * 4 * 20000 * 32 * 4096 / ms_last / 1e9);
    return 0;
}
