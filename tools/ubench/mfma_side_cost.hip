// Microbenchmark: how many MFMA-pipe cycles does ONE side instruction of the partner wave on the same SIMD cost?
// 512-thread blocks, 1 per CU: waves 0-3 stream v_mfma_f32_32x32x2_f32 (64 cycles each), waves 4-7 issue `num` side
// instructions per `den` partner MFMAs.  cost = (t - t_idle) / (#side instructions) in MFMA-clock cycles, valid while
// the MFMA waves (not the partners) are the critical path.
// build: hipcc --offload-arch=gfx950 -O3 mfma_side_cost.hip -o mfma_side_cost.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { IDLE, V_ADD, V_PK_ADD, DS_W128, DS_W64, DS_W32, DS_R128, GLD128, GLD32, BLD_LDS128, BLD_LDS32 };

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, const float *src, int iters, int n_side, float a0, float b0)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * 4 * 64 * 4 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = a0 + threadIdx.x * 1e-6f, b = b0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else if (MODE != IDLE) {
        float v0 = a0, c = b0 * 0.5f;
        f32x2 p0 = {a0, b0}, pc = {c, c};
        f32x4 w = {a0, b0, a0, b0};
        const unsigned laddr = (unsigned)(uintptr_t)&lds[((wave - 4) * 4 * 64 + lane) * 4];
        const float *gp = src + ((blockIdx.x * 4 + (wave - 4)) * 64 + lane) * 4;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1u << 24, 0x00020000);
        const unsigned voff = ((blockIdx.x * 4 + (wave - 4)) * 64 + lane) * 16;
        f32x4 acc4 = {0, 0, 0, 0};
        __attribute__((address_space(3))) float *lbase = (__attribute__((address_space(3))) float *)&lds[(wave - 4) * 4 * 64 * 4];
        for (int it = 0; it < n_side; it += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE == V_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(c));
                if (MODE == V_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(pc));
                if (MODE == DS_W128) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(laddr), "v"(w), "n"(0) : "memory");
                if (MODE == DS_W64) asm volatile("ds_write_b64 %0, %1" :: "v"(laddr), "v"(p0) : "memory");
                if (MODE == DS_W32) asm volatile("ds_write_b32 %0, %1" :: "v"(laddr), "v"(v0) : "memory");
                if (MODE == DS_R128) { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(laddr) : "memory"); acc4 += t; }
                if (MODE == GLD128) { f32x4 t; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(gp) : "memory"); acc4 += t; }
                if (MODE == GLD32) { float t; asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(gp) : "memory"); acc4[0] += t; }
                if (MODE == BLD_LDS128) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lbase, 16, voff, 0, 0, 0);
                if (MODE == BLD_LDS32) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lbase, 4, voff, 0, 0, 0);
            }
            if (MODE == DS_R128 || MODE == GLD128 || MODE == GLD32) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        s = v0 + p0[0] + p0[1] + acc4[0] + acc4[1] + acc4[2] + acc4[3] + lds[threadIdx.x & 63];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float t_idle = 0;
template <int MODE>
void run(const char *name, int iters, int num, int den)
{
    int grid = 256;
    float *out, *src; hipMalloc(&out, grid * 512 * 4); hipMalloc(&src, 1 << 24); hipMemset(src, 0, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_side = (int)((long long)iters * 32 * num / den);
    k<MODE><<<grid, 512>>>(out, src, 10, 40, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 512>>>(out, src, iters, n_side, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (MODE == IDLE) t_idle = ms;
    const double cyc_per_mfma = 64.0 * ms / t_idle;          // MFMA clock inferred from the idle run
    const double cost = n_side ? (ms - t_idle) / t_idle * 64.0 * iters * 32 / n_side : 0;
    printf("%-28s %d per %d MFMA: %8.3f ms  (%5.1f cyc per MFMA slot)  cost/instr %6.1f MFMA-cycles\n", name, num, den, ms,
           cyc_per_mfma, cost);
    hipFree(out); hipFree(src);
}
int main()
{
    const int it = 4000;
    run<IDLE>("partner idle", it, 0, 1);
    run<V_ADD>("v_add_f32", it, 1, 1); run<V_ADD>("v_add_f32", it, 4, 1);
    run<V_PK_ADD>("v_pk_add_f32", it, 1, 1); run<V_PK_ADD>("v_pk_add_f32", it, 4, 1);
    run<DS_W128>("ds_write_b128", it, 1, 8); run<DS_W128>("ds_write_b128", it, 1, 4); run<DS_W128>("ds_write_b128", it, 1, 2);
    run<DS_W64>("ds_write_b64", it, 1, 4); run<DS_W64>("ds_write_b64", it, 1, 2);
    run<DS_W32>("ds_write_b32", it, 1, 4); run<DS_W32>("ds_write_b32", it, 1, 1);
    run<DS_R128>("ds_read_b128", it, 1, 4); run<DS_R128>("ds_read_b128", it, 1, 2);
    run<GLD128>("global_load_dwordx4 (L2)", it, 1, 8); run<GLD128>("global_load_dwordx4 (L2)", it, 1, 4);
    run<GLD32>("global_load_dword (L2)", it, 1, 4);
    run<BLD_LDS32>("buffer_load_dword lds", it, 1, 4); run<BLD_LDS32>("buffer_load_dword lds", it, 1, 1);
    run<BLD_LDS128>("buffer_load_dwordx4 lds", it, 1, 8); run<BLD_LDS128>("buffer_load_dwordx4 lds", it, 1, 4);
    return 0;
}
