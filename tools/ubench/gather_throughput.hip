// Microbenchmark: vector-memory issue cost of buffer_load_dwordx4 for three lane->address maps (data L2-resident):
//   0: fully coalesced  (64 lanes x 16 B contiguous = 8 lines of 128 B)
//   1: MFMA-operand map (lane = (pixel l&31, half l>>5): 32 pixels x 32 B, pixel stride `pstride` bytes = 32 lines)
//   2: line map         (lane = (pixel l>>3, chunk l&7): 8 pixels x 128 B = 8 lines)
// NW waves per CU, each issuing `iters` x 16 loads then waiting.  Reports cycles per wave-instruction per CU.
// build: hipcc --offload-arch=gfx950 -O3 gather_throughput.hip -o gather_throughput.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, const float *src, unsigned bytes, int iters, unsigned pstride)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, bytes, 0x00020000);
    unsigned voff;
    const unsigned base = ((blockIdx.x * 16 + wave) * 37u % 64u) * 32u * pstride % (bytes / 4);
    if (MODE == 0) voff = base + lane * 16u;
    else if (MODE == 1) voff = base + (lane & 31) * pstride + (lane >> 5) * 16u;
    else voff = base + (lane >> 3) * pstride + (lane & 7) * 16u;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned so = (unsigned)(((it * 16 + q) * 32) & 0x1FF);      // walk along the 512-byte pixel rows
            const unsigned so2 = so + (MODE == 2 ? (q & 3) * 8 * pstride : 0) + (unsigned)(it & 63) * 32u * pstride;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v[q]) : "v"(voff), "s"(r), "s"(so2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int MODE>
void run(const char *name, int nw, unsigned pstride, double ghz)
{
    const int grid = 256, iters = 2000;
    const unsigned bytes = 64u << 20;           // 8 MB source: stays in L2/MALL
    float *out, *src; hipMalloc(&out, grid * 1024 * 4); hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, nw * 64>>>(out, src, bytes, 20, pstride);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, nw * 64>>>(out, src, bytes, iters, pstride);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 16 * nw, cyc = ms * 1e-3 * ghz * 1e9;
    printf("%-26s %2d waves/CU: %7.3f ms  %6.1f cycles per wave-load per CU   %6.1f B/clk/CU   %.2f TB/s chip\n", name, nw, ms, cyc / n,
           n * 1024 / cyc, n * 1024 * 256 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(src);
}
int main()
{
    for (int nw : {4, 8}) {
        run<0>("coalesced 1 KB", nw, 512, 2.4);
        run<1>("32 px x 32 B (MFMA map)", nw, 512, 2.4);
        run<2>("8 px x 128 B (line map)", nw, 512, 2.4);
    }
    return 0;
}
