// Effective shader clock while a kernel runs: every wave records s_memtime (1 tick per shader cycle, memtime_calib.hip) and
// s_memrealtime (constant 100 MHz) at its start and end; clock = d(memtime) / d(realtime) * 100 MHz.  Modes: pure MFMA stream,
// MFMA + 16-byte global loads (L2-resident), MFMA + LDS traffic, MFMA + packed VALU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(long long *out, const float *src, int iters, float a0)
{
    __shared__ __attribute__((aligned(16))) float lds[256 * 4];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f;
    f32x4 v = {a, a, a, a};
    f32x2 p = {a, a};
    const float *gp = src + ((blockIdx.x * 256 + threadIdx.x) % (1 << 20)) * 4;
    const long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[i], 0, 0, 0);
            if (MODE == 1) { f32x4 t; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(gp + (u & 3) * 1024) : "memory"); }
            if (MODE == 2) { asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(threadIdx.x * 16)), "v"(v) : "memory"); f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(threadIdx.x * 16)) : "memory"); }
            if (MODE == 3) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p)); }
        }
        if (MODE == 1 || MODE == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = p[0] + lds[threadIdx.x];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if ((threadIdx.x & 63) == 0) {
        long long *o = out + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 3;
        o[0] = t1 - t0; o[1] = r1 - r0; o[2] = (long long)s;
    }
}
template <int MODE>
void run(const char *name, int blocks_per_cu)
{
    const int grid = 256 * blocks_per_cu, iters = 6000;
    long long *out; hipMalloc(&out, (size_t)grid * 4 * 3 * 8);
    float *src; hipMalloc(&src, 16 << 20); hipMemset(src, 0, 16 << 20);
    k<MODE><<<grid, 256>>>(out, src, 50, 1.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, src, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long *h = (long long *)malloc((size_t)grid * 4 * 3 * 8);
    hipMemcpy(h, out, (size_t)grid * 4 * 3 * 8, hipMemcpyDeviceToHost);
    double ct = 0, rt = 0;
    for (int i = 0; i < grid * 4; ++i) { ct += h[i * 3]; rt += h[i * 3 + 1]; }
    const double flops = (double)grid * 4 * iters * 32 * 4096;
    printf("%-28s %d blk/CU: kernel %.3f ms, %6.1f TFLOP/s, shader clock %.3f GHz (memtime/realtime), MFMA-busy %.1f %% of cycles\n",
           name, blocks_per_cu, ms, flops / ms / 1e9, ct / rt * 0.1, 100.0 * iters * 32 * 64 * blocks_per_cu / (ct / (grid * 4)) );
    hipFree(out); hipFree(src); free(h);
}
int main()
{
    run<0>("pure MFMA", 1); run<0>("pure MFMA", 2);
    run<1>("MFMA + global loads (L2)", 1); run<1>("MFMA + global loads (L2)", 2);
    run<2>("MFMA + LDS write/read", 1); run<2>("MFMA + LDS write/read", 2);
    run<3>("MFMA + 2 v_pk_add per MFMA", 1);
    return 0;
}
