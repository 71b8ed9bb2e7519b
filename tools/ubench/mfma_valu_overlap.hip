// Microbenchmark: does VALU / LDS / VMEM work of a SECOND wave on the same SIMD overlap with a wave streaming
// v_mfma_f32_32x32x2_f32?  512-thread blocks, 1 per CU: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 (their SIMD
// partners) run `valu_per_mfma` side instructions per MFMA of the partner.  Reports MFMA-wave time vs the idle-partner case.
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: partner idle, 1: v_add_f32, 2: v_pk_add_f32, 3: ds_write_b128, 4: same wave interleaves v_add_f32 with its MFMAs,
// 5: v_fma_f32 (3 source operands), 6: same wave interleaves v_pk_add_f32
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, int side, float a0, float b0)
{
    __shared__ float lds[512 * 4];
    const int wave = threadIdx.x >> 6;
    float s = 0;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = a0 + threadIdx.x * 1e-6f, b = b0;
        float v0 = a, v1 = b, v2 = a + 1, v3 = b + 1;
        f32x2 p0 = {a, b}, p1 = {b, a};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    if (MODE == 4) {
                        for (int q = 0; q < 4; ++q) {    // 4 independent VALU per MFMA
                            asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(a));
                            asm volatile("v_add_f32 %0, %0, %1" : "+v"(v1) : "v"(a));
                        }
                    }
                    if (MODE == 6) {
                        for (int q = 0; q < 4; ++q) {
                            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(p1));
                        }
                    }
                }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        s += v0 + v1 + v2 + v3 + p0[0] + p0[1];
    } else if (MODE != 0 && MODE != 4 && MODE != 6) {
        float v0 = a0, v1 = b0, v2 = a0 + 1, v3 = b0 + 1, c = b0 * 0.5f;
        f32x2 p0 = {a0, b0}, p1 = {b0, a0}, p2 = {a0, a0}, p3 = {b0, b0}, pc = {c, c};
        const int n = iters * 32 * side / 4;          // `side` instructions per partner MFMA
        for (int it = 0; it < n; ++it) {
            if (MODE == 1) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(c));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v1) : "v"(c));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v2) : "v"(c));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v3) : "v"(c));
            } else if (MODE == 2) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(pc));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(pc));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(pc));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(pc));
            } else if (MODE == 3) {
                const unsigned addr = (unsigned)(uintptr_t)&lds[(threadIdx.x - 256) * 4];
                f32x4 w = {v0, v1, v2, v3};
                for (int q = 0; q < 4; ++q) asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(w) : "memory");
            } else if (MODE == 5) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(c), "v"(v1));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(c), "v"(v2));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(c), "v"(v3));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(c), "v"(v0));
            }
        }
        s = v0 + v1 + v2 + v3 + p0[0] + p1[1] + p2[0] + p3[1] + lds[threadIdx.x];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int iters, int side)
{
    int grid = 256;
    float *out; hipMalloc(&out, grid * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 512>>>(out, 10, side, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 512>>>(out, iters, side, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 32 * 2.0 * 32 * 32 * 2;
    printf("%-34s side/MFMA %2d: %8.3f ms  MFMA %6.1f TFLOP/s\n", name, side, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    const int it = 4000;
    run<0>("partner idle", it, 0);
    for (int side : {1, 2, 4, 8, 16}) run<1>("partner v_add_f32", it, side);
    for (int side : {1, 2, 4, 8, 16}) run<2>("partner v_pk_add_f32", it, side);
    for (int side : {1, 2, 4, 8}) run<5>("partner v_fma_f32", it, side);
    for (int side : {1, 2, 4}) run<3>("partner ds_write_b128", it, side);
    run<4>("same wave 8 v_add_f32 per MFMA", it, 8);
    run<6>("same wave 4 v_pk_add_f32 per MFMA", it, 4);
    return 0;
}
