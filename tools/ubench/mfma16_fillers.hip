// Microbenchmark: what does an instruction cost when it sits BETWEEN the v_mfma_f32_16x16x4_f32 of one wave's stream?
// The F(4x4) Winograd kernel (wino44_conv.hip) issues 4 (64-channel form) or 8 (128-channel form) MFMAs per transform position
// and hipcc places the input-transform VALU work between them; its timeline says ~12 cycles per VALU instruction.  This probe
// separates the candidates:
//   * the accumulator pattern of the MFMAs around the filler: 8 independent accumulators, two alternating chains (the
//     128-channel form), one chain (the 64-channel form: back-to-back dependent MFMAs);
//   * the filler: v_fma_f32, v_pk_fma_f32, v_add_f32, v_accvgpr_mov, ds_write_b32, ds_read_b128, s_add, s_nop;
//   * fillers per MFMA gap: 1, 2, 4, 6;
//   * one wave per SIMD and two waves per SIMD running the same stream (512-thread block).
// Output: shader cycles per MFMA per SIMD (32 = the matrix pipe's rate).
// build: hipcc --offload-arch=gfx950 -O3 mfma16_fillers.hip -o mfma16_fillers.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { F_NONE, F_FMA, F_PKFMA, F_ADD, F_ACCMOV, F_DSW32, F_DSR128, F_SADD, F_NOP, F_PKADD, F_BLD };
enum { C_IND8, C_CH2, C_CH1 };

template <int FILL>
__device__ __forceinline__ void filler(float (&f)[8], f32x2 (&p)[4], f32x4 &t, f32x4 &accx, unsigned laddr, int &sx, int i,
                                       __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const int k = i & 7;
    if constexpr (FILL == F_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k]) : "v"(f[(k + 3) & 7]), "v"(f[(k + 5) & 7]));
    if constexpr (FILL == F_ADD) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[k]) : "v"(f[(k + 3) & 7]));
    if constexpr (FILL == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k & 3]) : "v"(p[(k + 1) & 3]), "v"(p[(k + 2) & 3]));
    if constexpr (FILL == F_PKADD) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k & 3]) : "v"(p[(k + 1) & 3]));
    if constexpr (FILL == F_ACCMOV) asm volatile("v_accvgpr_mov_b32 %0, %0" : "+a"(accx[k & 3]));
    if constexpr (FILL == F_DSW32) asm volatile("ds_write_b32 %0, %1" ::"v"(laddr), "v"(f[k]) : "memory");
    if constexpr (FILL == F_DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(laddr) : "memory");
    if constexpr (FILL == F_SADD) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sx));
    if constexpr (FILL == F_NOP) asm volatile("s_nop 0");
    if constexpr (FILL == F_BLD) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(f[k]) : "v"(voff), "s"(r) : "memory");
}

template <int CHAIN, int FILL, int NF, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *out, long long *cyc, const float *src, int iters, float a0, float b0)
{
    __shared__ __attribute__((aligned(16))) float lds[THREADS * 4 + 16];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accx = {0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    float f[8];
    f32x2 p[4];
    for (int i = 0; i < 8; ++i) f[i] = a0 * (i + 1) * 1e-3f;
    for (int i = 0; i < 4; ++i) p[i] = f32x2{a0 * i * 1e-3f, b0 * 1e-3f};
    f32x4 t = {0, 0, 0, 0};
    int sx = 0;
    const unsigned laddr = (unsigned)(uintptr_t)&lds[threadIdx.x * 4];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1u << 20, 0x00020000);
    const unsigned voff = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 256;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            constexpr int dummy = 0;
            (void)dummy;
            const int ai = CHAIN == C_IND8 ? (u & 7) : (CHAIN == C_CH2 ? (u & 1) + 2 * ((u >> 3) & 3) : ((u >> 2) & 7));
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[ai]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < NF; ++q) filler<FILL>(f, p, t, accx, laddr, sx, u * NF + q, r, voff);
        }
        if constexpr (FILL == F_DSR128 || FILL == F_DSW32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (FILL == F_BLD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int q = 0; q < 4; ++q) s += acc[i][q];
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    s += t[0] + t[1] + t[2] + t[3] + accx[0] + accx[1] + accx[2] + accx[3] + (float)sx + lds[(threadIdx.x * 7) % THREADS];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

static const char *chain_name[] = {"8 independent", "2 chains    ", "1 chain     "};
static const char *fill_name[] = {"none", "v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_accvgpr_mov", "ds_write_b32", "ds_read_b128",
                                  "s_add_i32", "s_nop", "v_pk_add_f32", "buffer_load_dword"};

template <int CHAIN, int FILL, int NF, int THREADS>
void run(int iters)
{
    const int grid = 256, waves = THREADS / 64;
    float *out, *src;
    long long *cyc;
    hipMalloc(&out, grid * THREADS * 4); hipMalloc(&cyc, grid * waves * 8); hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    k<CHAIN, FILL, NF, THREADS><<<grid, THREADS>>>(out, cyc, src, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    k<CHAIN, FILL, NF, THREADS><<<grid, THREADS>>>(out, cyc, src, iters, 1.f, 1.f);
    hipDeviceSynchronize();
    static long long h[256 * 8];
    hipMemcpy(h, cyc, grid * waves * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < grid * waves; ++i) s += (double)h[i];
    const double per = s / (grid * waves) / ((double)iters * 32) / (waves / 4);     // cycles per MFMA per SIMD
    printf("%s  %-18s x%d  %d wave(s)/SIMD: %6.2f cycles per MFMA per SIMD  (+%5.2f per filler)\n", chain_name[CHAIN], fill_name[FILL], NF,
           waves / 4, per, NF && FILL != F_NONE ? (per - 32.0) / NF : 0.0);
    hipFree(out); hipFree(cyc); hipFree(src);
}

template <int CHAIN, int THREADS>
void sweep(int iters)
{
    run<CHAIN, F_NONE, 0, THREADS>(iters);
#define FOUR(F) run<CHAIN, F, 1, THREADS>(iters); run<CHAIN, F, 2, THREADS>(iters); run<CHAIN, F, 4, THREADS>(iters); run<CHAIN, F, 6, THREADS>(iters);
    FOUR(F_FMA) FOUR(F_PKFMA) FOUR(F_ADD) FOUR(F_PKADD) FOUR(F_ACCMOV) FOUR(F_DSW32) FOUR(F_DSR128) FOUR(F_SADD) FOUR(F_NOP) FOUR(F_BLD)
#undef FOUR
}

int main()
{
    const int it = 2000;
    sweep<C_IND8, 256>(it);
    sweep<C_CH2, 256>(it);
    sweep<C_CH1, 256>(it);
    sweep<C_IND8, 512>(it);
    sweep<C_CH2, 512>(it);
    sweep<C_CH1, 512>(it);
    return 0;
}
