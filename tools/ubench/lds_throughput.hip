// Microbenchmark: sustained LDS bandwidth per CU for b32/b64/b128 reads and writes (conflict-free, lane-contiguous),
// 1 block of NW waves per CU, no other work.  Prints bytes/clk/CU using the MFMA-idle clock estimate passed on argv.
// build: hipcc --offload-arch=gfx950 -O3 lds_throughput.hip -o lds_throughput.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { W32, W64, W128, R32, R64, R128, RW128 };
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[16 * 64 * 4 * 2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned a128 = (unsigned)(uintptr_t)&lds[(wave * 64 + lane) * 4];
    const unsigned a64 = (unsigned)(uintptr_t)&lds[(wave * 64 + lane) * 2];
    const unsigned a32 = (unsigned)(uintptr_t)&lds[(wave * 64 + lane)];
    f32x4 w = {1.f, 2.f, 3.f, (float)lane};
    f32x2 w2 = {1.f, (float)lane};
    float w1 = (float)lane;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == W128) asm volatile("ds_write_b128 %0, %1" :: "v"(a128), "v"(w) : "memory");
            if (MODE == W64) asm volatile("ds_write_b64 %0, %1" :: "v"(a64), "v"(w2) : "memory");
            if (MODE == W32) asm volatile("ds_write_b32 %0, %1" :: "v"(a32), "v"(w1) : "memory");
            if (MODE == R128) { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(a128) : "memory"); }
            if (MODE == R64) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a64) : "memory"); }
            if (MODE == R32) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(a32) : "memory"); }
            if (MODE == RW128) {
                if (wave & 1) asm volatile("ds_write_b128 %0, %1" :: "v"(a128), "v"(w) : "memory");
                else { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(a128) : "memory"); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + lds[threadIdx.x];
}
template <int MODE>
void run(const char *name, int nw, int bytes, double ghz)
{
    const int grid = 256, iters = 20000;
    float *out; hipMalloc(&out, grid * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, nw * 64>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, nw * 64>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8 * nw;            // wave-instructions per CU
    const double cyc = ms * 1e-3 * ghz * 1e9;
    printf("%-14s %2d waves/CU: %7.3f ms  %5.1f cycles per wave-instruction  %6.1f B/clk/CU (at %.2f GHz)\n", name, nw, ms,
           cyc / n, n * 64 * bytes / cyc, ghz);
    hipFree(out);
}
int main(int argc, char **argv)
{
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    for (int nw : {4, 8, 16}) {
        run<W128>("ds_write_b128", nw, 16, ghz); run<W64>("ds_write_b64", nw, 8, ghz); run<W32>("ds_write_b32", nw, 4, ghz);
        run<R128>("ds_read_b128", nw, 16, ghz); run<R64>("ds_read_b64", nw, 8, ghz); run<R32>("ds_read_b32", nw, 4, ghz);
        run<RW128>("rd+wr b128", nw, 16, ghz);
    }
    return 0;
}
