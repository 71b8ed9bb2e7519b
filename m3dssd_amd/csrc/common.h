// Internal helpers shared by the HIP translation units of libm3dssd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "m3dssd_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- raw buffer access (gfx950) -------------------------------------------------------------------------------
// A buffer resource makes the address of a load  base(SGPR x4) + voffset(VGPR, 32 bit) + soffset(SGPR): no per-load
// 64-bit VALU address arithmetic, and a lane whose voffset is >= num_records reads 0.0f -- out-of-image taps cost no
// v_cndmask.  Every VALU instruction saved matters: on this part a VALU instruction of ANY wave of a SIMD delays that
// SIMD's MFMA stream by ~8 cycles (tools/ubench/mfma_valu_overlap.hip), i.e. VALU work does not hide under MFMAs.
#define M3D_BUF_OOB 0x80000000u          // voffset of a masked lane (views are < 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voffset, unsigned soffset)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0));
}
// f32x4 add / subtract as two packed v_pk_add_f32.  hipcc scalarises a <4 x float> fsub into four v_sub_f32 (and folds
// shuffles / fneg back into that fsub), so the subtraction is spelled in asm; it is not volatile: free to schedule.
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b)
{
    const f32x2 lo = pk_sub2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_sub2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// scalar * f32x4 (+ f32x4) as two packed ops
__device__ __forceinline__ f32x4 pk_mul_s(float w, f32x4 x)
{
    const f32x2 w2 = {w, w};
    const f32x2 lo = w2 * __builtin_shufflevector(x, x, 0, 1), hi = w2 * __builtin_shufflevector(x, x, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 pk_fma_s(float w, f32x4 x, f32x4 c)
{
    const f32x2 w2 = {w, w};
    const f32x2 lo = __builtin_elementwise_fma(w2, __builtin_shufflevector(x, x, 0, 1), __builtin_shufflevector(c, c, 0, 1));
    const f32x2 hi = __builtin_elementwise_fma(w2, __builtin_shufflevector(x, x, 2, 3), __builtin_shufflevector(c, c, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 pk_add(f32x4 a, f32x4 b)
{
    const f32x2 lo = __builtin_shufflevector(a, a, 0, 1) + __builtin_shufflevector(b, b, 0, 1);
    const f32x2 hi = __builtin_shufflevector(a, a, 2, 3) + __builtin_shufflevector(b, b, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

#define M3D_LEAKY_SLOPE 0.01f

void m3d_set_error(const char *fmt, ...);

#define M3D_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            m3d_set_error(__VA_ARGS__); \
            return M3D_E_ARG;         \
        }                             \
    } while (0)

#define M3D_HIP(call)                                                             \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) {                                                  \
            m3d_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                          __FILE__, __LINE__);                                    \
            return M3D_E_HIP;                                                     \
        }                                                                         \
    } while (0)

#define M3D_LAUNCH_CHECK()                                                          \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            m3d_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), \
                          __FILE__, __LINE__);                                      \
            return M3D_E_HIP;                                                       \
        }                                                                           \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int imin(int a, int b) { return a < b ? a : b; }

// LeakyReLU as max(v, slope*v) (identical for 0 < slope < 1, signed zeros included): 2 VALU ops instead of 3
__device__ __forceinline__ float leaky(float v) { return fmaxf(v, v * M3D_LEAKY_SLOPE); }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---- deterministic split-K reduction shared by the LDS-tiled and the wave-granular implicit GEMMs (igemm_conv.hip) -----------
// ws holds `splits` raw partial sums [split][M][Cout_pad]; they are added in split order and the conv epilogue is applied.
struct SplitkReduceArgs {
    const float *ws, *scale, *shift, *res;
    float *out;
    int M, Cout, Cout_pad, splits, out_cs, res_cs, res_mode, act, sigmoid_from;
};
int m3d_launch_splitk_reduce(const SplitkReduceArgs &a, hipStream_t stream);
